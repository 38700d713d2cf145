"""ctypes binding of the CPU oracle (oracle/liboracle.so) and of the shim over the reference's stand-alone
headers (oracle/_ref/libref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the cpu_baseline leg of
bench.py.  The product package never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(_HERE, "liboracle.so")
REF_SO = os.path.join(_HERE, "_ref", "libref.so")


class OrcGrid(C.Structure):
    _fields_ = [("nx", C.c_int), ("ny", C.c_int), ("nz", C.c_int), ("ox", C.c_int), ("oy", C.c_int), ("oz", C.c_int),
                ("dx", C.c_float), ("phi_fluid_bg", C.c_float), ("phi_solid_bg", C.c_float)]


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def build(force: bool = False):
    """Compile liboracle.so (and _ref/libref.so when the reference checkout is present)."""
    if force or not os.path.exists(ORACLE_SO):
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def load_reference_shim():
    """The real headers/util.h + headers/pcgsolver of the reference behind a C ABI, or None when the shim
    has not been built (it can only be built where /root/reference exists)."""
    if not os.path.exists(REF_SO):
        return None
    lib = C.CDLL(REF_SO)
    f32, f64, i32, vp = C.c_float, C.c_double, C.c_int32, C.c_void_p
    lib.ref_trilinear_hat_d.restype = f64; lib.ref_trilinear_hat_d.argtypes = [f64]
    lib.ref_trilinear_hat_f.restype = f32; lib.ref_trilinear_hat_f.argtypes = [f32]
    lib.ref_clamp_f.restype = f32; lib.ref_clamp_f.argtypes = [f32, f32, f32]
    lib.ref_sqr_f.restype = f32; lib.ref_sqr_f.argtypes = [f32]
    lib.ref_fraction_inside2_f.restype = f32; lib.ref_fraction_inside2_f.argtypes = [f32, f32]
    lib.ref_fraction_inside4_f.restype = f32; lib.ref_fraction_inside4_f.argtypes = [f32, f32, f32, f32]
    lib.ref_fraction_inside2_d.restype = f64; lib.ref_fraction_inside2_d.argtypes = [f64, f64]
    lib.ref_pcg_solve.restype = C.c_int
    lib.ref_pcg_solve.argtypes = [i32, vp, vp, vp, vp, f64, i32, vp, C.POINTER(f64), C.POINTER(i32)]
    lib.ref_mic0_invdiag.restype = None; lib.ref_mic0_invdiag.argtypes = [i32, vp, vp, vp, vp]
    lib.ref_spmv.restype = None; lib.ref_spmv.argtypes = [i32, vp, vp, vp, vp, vp]
    return lib


class Oracle:
    """Stage-by-stage CPU restatement on dense [nz,ny,nx] numpy arrays."""

    def __init__(self, desc: dict, threads: int = 1):
        build()
        self.lib = C.CDLL(ORACLE_SO)
        self.g = OrcGrid(desc["nx"], desc["ny"], desc["nz"], desc["ox"], desc["oy"], desc["oz"], desc["dx"],
                         desc["phi_fluid_bg"], desc["phi_solid_bg"])
        self.rho = np.float32(desc["rest_density"])
        self.shape = (desc["nz"], desc["ny"], desc["nx"])
        self.B = desc["nx"] * desc["ny"] * desc["nz"]
        L = self.lib
        f32, f64, i32, i64, vp = C.c_float, C.c_double, C.c_int32, C.c_int64, C.c_void_p
        G = C.POINTER(OrcGrid)
        L.orc_set_num_threads.argtypes = [C.c_int]
        L.orc_max_threads.restype = C.c_int
        L.orc_trilinear_hat.restype = f64; L.orc_trilinear_hat.argtypes = [f64]
        L.orc_clamp_f.restype = f32; L.orc_clamp_f.argtypes = [f32, f32, f32]
        L.orc_fraction_inside2.restype = f32; L.orc_fraction_inside2.argtypes = [f32, f32]
        L.orc_fraction_inside4.restype = f32; L.orc_fraction_inside4.argtypes = [f32, f32, f32, f32]
        L.orc_kernel_staggered.argtypes = [vp, vp]
        L.orc_kernel.restype = f64; L.orc_kernel.argtypes = [vp]
        L.orc_bin_particles.restype = i64; L.orc_bin_particles.argtypes = [G, i64, vp, vp, vp, vp]
        L.orc_particle_mask.argtypes = [G, vp, C.c_int, vp]
        L.orc_p2g.argtypes = [G, i64, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_staggered_sample.argtypes = [G, vp, vp, vp, vp, vp]
        L.orc_g2p.argtypes = [G, i64, vp, vp, vp, vp, vp, vp, vp, vp, f32]
        L.orc_apply_displacements.argtypes = [G, i64, vp, vp, vp, vp]
        L.orc_weight_mask.argtypes = [G, vp, f32, vp]
        L.orc_corner_value.restype = f32; L.orc_corner_value.argtypes = [G, vp, f32, C.c_int, C.c_int, C.c_int]
        L.orc_face_fractions.argtypes = [G, vp, f32, vp, vp, vp, vp]
        L.orc_combine_weights.argtypes = [i64, vp, vp]
        L.orc_index_cells.restype = i32; L.orc_index_cells.argtypes = [G, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_density.restype = i64; L.orc_density.argtypes = [G, i64, vp, vp, vp, vp, vp, vp, f32, vp]
        L.orc_advect_particles.argtypes = [G, i64, vp, vp, vp, vp, vp, vp, f32, f32]
        L.orc_fluid_levelset.argtypes = [G, i64, vp, vp, vp, vp, f32, f32, vp]
        L.orc_redistribute.restype = i64; L.orc_redistribute.argtypes = [G, i64, vp, vp, vp, vp, f32, C.c_uint32, C.POINTER(i64)]
        L.orc_build_system.argtypes = [G, C.c_int, f32, f32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_stencil_to_csr.restype = i64; L.orc_stencil_to_csr.argtypes = [G, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_build_J.argtypes = [G, vp, vp, vp, vp, vp, i32, vp]
        L.orc_sphere_minv.argtypes = [f32, f64, vp]
        L.orc_apply_operator.argtypes = [i32, vp, vp, vp, vp, vp, f64, vp, vp]
        L.orc_cg_jacobi.restype = i32
        L.orc_cg_jacobi.argtypes = [i32, vp, vp, vp, vp, vp, f64, vp, f64, i32, C.c_int, vp, C.POINTER(f64)]
        L.orc_cg_eliminated.restype = i32
        L.orc_cg_eliminated.argtypes = [i32, vp, vp, vp, vp, vp, vp, f64, vp, f64, i32, C.c_int, vp, C.POINTER(f64)]
        L.orc_pcg_mic0.restype = C.c_int
        L.orc_pcg_mic0.argtypes = [i32, vp, vp, vp, vp, f64, i32, vp, C.POINTER(f64), C.POINTER(i32), vp]
        L.orc_project_velocity.argtypes = [G, f32, f32] + [vp] * 15
        L.orc_solid_velocity_update.argtypes = [i32, vp, vp, f32, vp, vp]
        L.orc_project_displacement.argtypes = [G, f32, f32] + [vp] * 9
        L.orc_update_diff.argtypes = [i64, vp, vp, vp]
        L.orc_extrapolate_sweep.argtypes = [G, vp, vp, vp, vp]
        L.orc_add_gravity.argtypes = [i64, vp, vp, f32, f32]
        L.orc_compute_cfl.restype = f32; L.orc_compute_cfl.argtypes = [i64, vp, vp, vp, vp, f32, vp, vp]
        L.orc_constrain_velocity.argtypes = [G, vp, f32] + [vp] * 9 + [C.c_int, vp, vp]
        L.orc_constrain_velocity.restype = None
        L.orc_set_num_threads(threads)

    # ---- helpers ----
    def set_threads(self, n: int):
        self.lib.orc_set_num_threads(n)

    def max_threads(self) -> int:
        return self.lib.orc_max_threads()

    def _f32(self):
        return np.zeros(self.shape, dtype=np.float32)

    def _f64(self):
        return np.zeros(self.shape, dtype=np.float64)

    def _u8(self):
        return np.zeros(self.shape, dtype=np.uint8)

    # ---- stages ----
    def bin_particles(self, pos):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        n = pos.shape[0]
        cell_id = np.empty(n, dtype=np.int32)
        cell_start = np.empty(self.B + 1, dtype=np.int32)
        perm = np.empty(n, dtype=np.int32)
        n_out = self.lib.orc_bin_particles(C.byref(self.g), n, _ptr(pos), _ptr(cell_id), _ptr(cell_start), _ptr(perm))
        return cell_id, cell_start, perm, n_out

    def particle_mask(self, cell_start, radius: int):
        m = self._u8()
        self.lib.orc_particle_mask(C.byref(self.g), _ptr(cell_start), radius, _ptr(m))
        return m

    def p2g(self, pos, vel, cell_start, perm, band2):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        vel = np.ascontiguousarray(vel, dtype=np.float64)
        u, v, w = self._f64(), self._f64(), self._f64()
        self.lib.orc_p2g(C.byref(self.g), pos.shape[0], _ptr(pos), _ptr(vel), _ptr(cell_start), _ptr(perm), _ptr(band2),
                         _ptr(u), _ptr(v), _ptr(w))
        return u, v, w

    def g2p(self, pos, vel, uvw, duvw, ratio):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        out = np.array(vel, dtype=np.float64, order="C", copy=True)
        self.lib.orc_g2p(C.byref(self.g), pos.shape[0], _ptr(pos), _ptr(out), _ptr(uvw[0]), _ptr(uvw[1]), _ptr(uvw[2]),
                         _ptr(duvw[0]), _ptr(duvw[1]), _ptr(duvw[2]), np.float32(ratio))
        return out

    def apply_displacements(self, pos, disp):
        out = np.array(pos, dtype=np.float64, order="C", copy=True)
        self.lib.orc_apply_displacements(C.byref(self.g), out.shape[0], _ptr(out), _ptr(disp[0]), _ptr(disp[1]),
                                         _ptr(disp[2]))
        return out

    def weight_mask(self, phi, bg):
        m = self._u8()
        self.lib.orc_weight_mask(C.byref(self.g), _ptr(phi), np.float32(bg), _ptr(m))
        return m

    def face_fractions(self, phi, bg, mask):
        wu, wv, ww = self._f32(), self._f32(), self._f32()
        self.lib.orc_face_fractions(C.byref(self.g), _ptr(phi), np.float32(bg), _ptr(mask), _ptr(wu), _ptr(wv), _ptr(ww))
        return wu, wv, ww

    def combine_weights(self, w, w_obj):
        for a, b in zip(w, w_obj if w_obj is not None else (None, None, None)):
            self.lib.orc_combine_weights(self.B, _ptr(a), _ptr(b))

    def index_cells(self, pmask1, phi_fluid, phi_solid, w):
        label = self._u8()
        index = np.empty(self.shape, dtype=np.int32)
        n = self.lib.orc_index_cells(C.byref(self.g), _ptr(pmask1), _ptr(phi_fluid), _ptr(phi_solid), _ptr(w[0]), _ptr(w[1]),
                                     _ptr(w[2]), _ptr(label), _ptr(index))
        return label, index, n

    def density(self, pos, mass, cell_start, perm, band2, phi_fluid):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        mass = np.ascontiguousarray(mass, dtype=np.float32)
        d = self._f32()
        n_over = self.lib.orc_density(C.byref(self.g), pos.shape[0], _ptr(pos), _ptr(mass), _ptr(cell_start), _ptr(perm),
                                      _ptr(band2), _ptr(phi_fluid), self.rho, _ptr(d))
        return d, n_over

    def advect_particles(self, pos, uvw, phi_solid, dt, phi_object=None, phi_object_bg=0.0):
        """RK3 advection through the staggered field + push-out of the solids (src/FluidDomain.cpp:63-75)."""
        out = np.ascontiguousarray(pos, dtype=np.float64).copy()
        u, v, w = (np.ascontiguousarray(a, dtype=np.float64) for a in uvw)
        po = None if phi_object is None else np.ascontiguousarray(phi_object, dtype=np.float32)
        self.lib.orc_advect_particles(C.byref(self.g), out.shape[0], _ptr(out), _ptr(u), _ptr(v), _ptr(w),
                                      _ptr(np.ascontiguousarray(phi_solid, dtype=np.float32)), _ptr(po), np.float32(phi_object_bg),
                                      np.float32(dt))
        return out

    def fluid_levelset(self, pos, cell_start, perm, phi_solid, radius_world, half_width_voxels=3.0):
        """Union of the particles' balls merged with the solid (src/FluidDomain.cpp:50-61)."""
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        out = self._f32()
        self.lib.orc_fluid_levelset(C.byref(self.g), pos.shape[0], _ptr(pos), _ptr(cell_start), _ptr(perm),
                                    _ptr(np.ascontiguousarray(phi_solid, dtype=np.float32)), np.float32(radius_world),
                                    np.float32(half_width_voxels), _ptr(out))
        return out

    def redistribute(self, pos, cell_start, perm, density, seed):
        """Re-seeds the particles of every cell above 1.5 rest density (src/Functors.cpp:268-322, hash-driven); returns
        (new positions, cells touched, particles moved)."""
        out = np.ascontiguousarray(pos, dtype=np.float64).copy()
        moved = C.c_int64(0)
        cells = self.lib.orc_redistribute(C.byref(self.g), out.shape[0], _ptr(out), _ptr(cell_start), _ptr(perm),
                                          _ptr(np.ascontiguousarray(density, dtype=np.float32)), self.rho, int(seed) & 0xffffffff,
                                          C.byref(moved))
        return out, int(cells), int(moved.value)

    def build_system(self, kind, dt, label, phi_fluid, phi_solid, w, uvw=None, density=None):
        off = np.zeros((6,) + self.shape, dtype=np.float64)
        diag, rhs = self._f64(), self._f64()
        u, v, ww = uvw if uvw is not None else (None, None, None)
        self.lib.orc_build_system(C.byref(self.g), kind, np.float32(dt), self.rho, _ptr(label), _ptr(phi_fluid),
                                  _ptr(phi_solid), _ptr(w[0]), _ptr(w[1]), _ptr(w[2]), _ptr(u), _ptr(v), _ptr(ww),
                                  _ptr(density), _ptr(off), _ptr(diag), _ptr(rhs))
        return off, diag, rhs

    def to_csr(self, index, off, diag, rhs):
        n = int(index.max()) + 1 if index.size else 0
        rowptr = np.zeros(n + 1, dtype=np.int32)
        col = np.zeros(7 * max(n, 1), dtype=np.int32)
        val = np.zeros(7 * max(n, 1), dtype=np.float64)
        b = np.zeros(max(n, 1), dtype=np.float64)
        nnz = self.lib.orc_stencil_to_csr(C.byref(self.g), _ptr(index), _ptr(off), _ptr(diag), _ptr(rhs), _ptr(rowptr),
                                          _ptr(col), _ptr(val), _ptr(b))
        return rowptr, col[:nnz].copy(), val[:nnz].copy(), b[:n].copy()

    def build_J(self, index, ow, com):
        n = int(index.max()) + 1
        J = np.zeros((6, n), dtype=np.float64)
        com = np.ascontiguousarray(com, dtype=np.float64)
        self.lib.orc_build_J(C.byref(self.g), _ptr(index), _ptr(ow[0]), _ptr(ow[1]), _ptr(ow[2]), _ptr(com), n, _ptr(J))
        return J

    def sphere_minv(self, radius, density):
        m = np.zeros(6, dtype=np.float64)
        self.lib.orc_sphere_minv(np.float32(radius), float(density), _ptr(m))
        return m

    def apply_operator(self, rowptr, col, val, x, J=None, minv=None, dt=0.0):
        n = rowptr.size - 1
        x = np.ascontiguousarray(x, dtype=np.float64)
        y = np.zeros(n, dtype=np.float64)
        self.lib.orc_apply_operator(n, _ptr(rowptr), _ptr(col), _ptr(val), _ptr(J), _ptr(minv), float(dt), _ptr(x), _ptr(y))
        return y

    def cg_jacobi(self, rowptr, col, val, rhs, tol=1e-15, max_iter=0, J=None, minv=None, dt=0.0, drop_null_rows=True):
        n = rowptr.size - 1
        if max_iter <= 0:
            max_iter = 2 * n
        x = np.zeros(n, dtype=np.float64)
        err = C.c_double(0)
        it = self.lib.orc_cg_jacobi(n, _ptr(rowptr), _ptr(col), _ptr(val), _ptr(J), _ptr(minv), float(dt), _ptr(rhs),
                                    float(tol), max_iter, 1 if drop_null_rows else 0, _ptr(x), C.byref(err))
        return x, it, err.value

    def cg_eliminated(self, rowptr, col, val, rhs, dep, tol=1e-15, max_iter=0, J=None, minv=None, dt=0.0,
                      drop_null_rows=True):
        """Block elimination of the dependent rows (dep[i] != 0), CG on the rest; see orc_cg_eliminated."""
        n = rowptr.size - 1
        if max_iter <= 0:
            max_iter = 2 * n
        dep = np.ascontiguousarray(dep, dtype=np.uint8)
        x = np.zeros(n, dtype=np.float64)
        err = C.c_double(0)
        it = self.lib.orc_cg_eliminated(n, _ptr(rowptr), _ptr(col), _ptr(val), _ptr(dep), _ptr(J), _ptr(minv), float(dt),
                                        _ptr(rhs), float(tol), max_iter, 1 if drop_null_rows else 0, _ptr(x), C.byref(err))
        return x, it, err.value

    def solve(self, rowptr, col, val, rhs, dep, tol, max_iter, J, minv, dt, drop_null_rows, eliminate):
        """The solve the GPU library performs: elimination when asked for and possible, else CG on all rows."""
        if eliminate and dep.any() and (J is None or not np.any(J[:, dep.astype(bool)])):
            return self.cg_eliminated(rowptr, col, val, rhs, dep, tol, max_iter, J, minv, dt, drop_null_rows)
        return self.cg_jacobi(rowptr, col, val, rhs, tol, max_iter, J, minv, dt, drop_null_rows)

    def pcg_mic0(self, rowptr, col, val, rhs, tol_factor=1e-12, max_iter=100, want_invdiag=False):
        n = rowptr.size - 1
        x = np.zeros(n, dtype=np.float64)
        res = C.c_double(0)
        it = C.c_int32(0)
        invd = np.zeros(n, dtype=np.float64) if want_invdiag else None
        ok = self.lib.orc_pcg_mic0(n, _ptr(rowptr), _ptr(col), _ptr(val), _ptr(rhs), float(tol_factor), max_iter, _ptr(x),
                                   C.byref(res), C.byref(it), _ptr(invd))
        return x, bool(ok), res.value, it.value, invd

    def project_velocity(self, dt, index, p, phi_fluid, w, uvw):
        out = [self._f64() for _ in range(3)]
        valid = [self._u8() for _ in range(3)]
        p = np.ascontiguousarray(p, dtype=np.float64)
        self.lib.orc_project_velocity(C.byref(self.g), np.float32(dt), self.rho, _ptr(index), _ptr(p), _ptr(phi_fluid),
                                      _ptr(w[0]), _ptr(w[1]), _ptr(w[2]), _ptr(uvw[0]), _ptr(uvw[1]), _ptr(uvw[2]),
                                      _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), _ptr(valid[0]), _ptr(valid[1]),
                                      _ptr(valid[2]))
        return out, valid

    def solid_velocity_update(self, J, minv, dt, p, V):
        V = np.array(V, dtype=np.float64, copy=True)
        self.lib.orc_solid_velocity_update(J.shape[1], _ptr(J), _ptr(minv), np.float32(dt), _ptr(p), _ptr(V))
        return V

    def project_displacement(self, dt, index, p, phi_fluid, w):
        out = [self._f64() for _ in range(3)]
        p = np.ascontiguousarray(p, dtype=np.float64)
        self.lib.orc_project_displacement(C.byref(self.g), np.float32(dt), self.rho, _ptr(index), _ptr(p), _ptr(phi_fluid),
                                          _ptr(w[0]), _ptr(w[1]), _ptr(w[2]), _ptr(out[0]), _ptr(out[1]), _ptr(out[2]))
        return out

    def update_diff(self, front, prev):
        out = []
        for f, p in zip(front, prev):
            d = self._f64()
            self.lib.orc_update_diff(self.B, _ptr(f), _ptr(p), _ptr(d))
            out.append(d)
        return out

    def extrapolate(self, uvw, valid, active, iterations=2):
        uvw = [a.copy() for a in uvw]
        valid = [a.copy() for a in valid]
        active = active.copy()
        for _ in range(iterations):
            new_f, new_v = [], []
            for f, v in zip(uvw, valid):
                fo, vo = self._f64(), self._u8()
                self.lib.orc_extrapolate_sweep(C.byref(self.g), _ptr(f), _ptr(v), _ptr(fo), _ptr(vo))
                new_f.append(fo)
                new_v.append(vo)
            for v_old, v_new in zip(valid, new_v):
                active |= (v_new & ~v_old) & 1      # a written voxel becomes active
            uvw, valid = new_f, new_v
        return uvw, valid, active

    def add_gravity(self, v, active, dt, grav=9.81):
        out = v.copy()
        self.lib.orc_add_gravity(self.B, _ptr(active), _ptr(out), np.float32(grav), np.float32(dt))
        return out

    def compute_cfl(self, uvw, active, solid_translational=None):
        """src/FluidSimulator.cpp:448-458 under the component-wise reading of the Vec3d maximum (see orc_compute_cfl).
        Returns (dt, the three maxima)."""
        m = np.zeros(3)
        st = None if solid_translational is None else np.ascontiguousarray(solid_translational, np.float64)
        dt = self.lib.orc_compute_cfl(self.B, _ptr(np.ascontiguousarray(active, np.uint8)), _ptr(uvw[0]), _ptr(uvw[1]), _ptr(uvw[2]),
                                      np.float32(self.g.dx), _ptr(st), _ptr(m))
        return float(dt), m

    def constrain_velocity(self, phi, bg, w, uvw, obj_vel=None):
        """src/FluidSimulator.cpp:1180-1321: faces with solid fraction 1 lose the normal component (obj_vel: the
        rigid body's 6-vector for the overload that adds the body's normal velocity)."""
        out = [self._f64() for _ in range(3)]
        vt = va = None
        if obj_vel is not None:
            ov = np.ascontiguousarray(obj_vel, dtype=np.float64)
            vt, va = np.ascontiguousarray(ov[:3]), np.ascontiguousarray(ov[3:])
        self.lib.orc_constrain_velocity(C.byref(self.g), _ptr(np.ascontiguousarray(phi, np.float32)), np.float32(bg),
                                        _ptr(w[0]), _ptr(w[1]), _ptr(w[2]), _ptr(uvw[0]), _ptr(uvw[1]), _ptr(uvw[2]),
                                        _ptr(out[0]), _ptr(out[1]), _ptr(out[2]), 0 if obj_vel is None else 1, _ptr(vt), _ptr(va))
        return out

    # ---- one substep, the same stage order as fluidsimulation_amd.substep.FlipSubstep ----
    def substep(self, st: dict, dt: float, ratio: float = 0.95, tol: float = 1e-15, density_projection: bool = True,
                max_iter: int = 0, drop_null_rows: bool = True, eliminate: bool = True, constrain: bool = False) -> dict:
        """st holds: pos, vel, mass, phi_fluid, phi_solid, w (3 stale weights), valid (3), and optionally
        obj = dict(phi, bg, com, minv, vel) and ow (the body's stale weights).  constrain: also the two
        constrain_velocity calls of src/FluidSimulator.cpp:357-361 and :404-408.  Returns a dict with every
        intermediate field."""
        out = {}
        pos, vel, mass = st["pos"], st["vel"], st["mass"]
        phi_f, phi_s = st["phi_fluid"], st["phi_solid"]
        cell_id, cell_start, perm, n_out = self.bin_particles(pos)
        out.update(cell_id=cell_id, cell_start=cell_start, perm=perm)
        pm1 = self.particle_mask(cell_start, 1)
        pm2 = self.particle_mask(cell_start, 2)
        out.update(pmask1=pm1, band2=pm2)
        uvw = self.p2g(pos, vel, cell_start, perm, pm2)
        out["p2g"] = [a.copy() for a in uvw]
        active = pm2.copy()
        valid = [np.maximum(v, active) for v in st["valid"]]
        uvw, valid, active = self.extrapolate(uvw, valid, active, 2)
        out["extrap1"] = [a.copy() for a in uvw]
        if constrain:   # with the face fractions of the previous substep, as the reference's call order has it
            uvw = self.constrain_velocity(phi_s, self.g.phi_solid_bg, st["w"], uvw)
            if st.get("obj") is not None and st.get("ow") is not None:
                uvw = self.constrain_velocity(st["obj"]["phi"], st["obj"]["bg"], st["ow"], uvw, st["obj"]["vel"])
            out["constrained1"] = [a.copy() for a in uvw]
        prev = [a.copy() for a in uvw]
        uvw = [uvw[0], self.add_gravity(uvw[1], active, dt), uvw[2]]
        out["pre_project"] = [a.copy() for a in uvw]
        label, index, n_rows = self.index_cells(pm1, phi_f, phi_s, st["w"])
        out.update(label=label, index=index, n_rows=n_rows)
        obj = st.get("obj")
        ow = None
        if obj is not None:
            om = self.weight_mask(obj["phi"], obj["bg"])
            ow = self.face_fractions(obj["phi"], obj["bg"], om)
        wm = self.weight_mask(phi_f, self.g.phi_fluid_bg)
        fw = self.face_fractions(phi_f, self.g.phi_fluid_bg, wm)
        w = self.face_fractions(phi_s, self.g.phi_solid_bg, wm)
        self.combine_weights(w, ow)
        out.update(w=w, fw=fw, ow=ow)
        dens, n_over = self.density(pos, mass, cell_start, perm, pm2, phi_f)
        out.update(density=dens, n_over=n_over)
        off, diag, rhs = self.build_system(0, dt, label, phi_f, phi_s, w, uvw=uvw)
        rowptr, col, val, b = self.to_csr(index, off, diag, rhs)
        J = minv = None
        V = None
        if obj is not None:
            J = self.build_J(index, ow, obj["com"])
            minv = np.ascontiguousarray(obj["minv"], dtype=np.float64)
            V = np.ascontiguousarray(obj["vel"], dtype=np.float64)
            b = b - J.T @ V
        out["sys_div"] = (off, diag, rhs, b)
        dep = (phi_f[label == 1] >= 0).astype(np.uint8)      # rows of labelled cells that are not liquid
        out["dep"] = dep
        p, it, err = self.solve(rowptr, col, val, b, dep, tol, max_iter, J, minv, float(np.float32(dt)), drop_null_rows,
                                eliminate)
        out.update(p_div=p, it_div=it, err_div=err)
        uvw, valid = self.project_velocity(dt, index, p, phi_f, w, uvw)
        active = np.ones_like(active)
        out["projected"] = [a.copy() for a in uvw]
        if obj is not None:
            out["obj_vel"] = self.solid_velocity_update(J, minv, dt, p, V)
        disp = [self._f64() for _ in range(3)]
        if density_projection:
            off2, diag2, rhs2 = self.build_system(1, dt, label, phi_f, phi_s, w, density=dens)
            rowptr2, col2, val2, b2 = self.to_csr(index, off2, diag2, rhs2)
            out["sys_den"] = (off2, diag2, rhs2, b2)
            p2, it2, err2 = self.solve(rowptr2, col2, val2, b2, dep, tol, max_iter, None, None, 0.0, drop_null_rows, eliminate)
            out.update(p_den=p2, it_den=it2, err_den=err2)
            disp = self.project_displacement(dt, index, p2, phi_f, w)
            out["disp"] = disp
        uvw, valid, active = self.extrapolate(uvw, valid, active, 2)
        if constrain:
            uvw = self.constrain_velocity(phi_s, self.g.phi_solid_bg, w, uvw)
            if obj is not None:
                uvw = self.constrain_velocity(obj["phi"], obj["bg"], ow, uvw, out["obj_vel"])
        out["final_vel"] = uvw
        diff = self.update_diff(uvw, prev)
        out["diff"] = diff
        new_vel = self.g2p(pos, vel, uvw, diff, ratio)
        new_pos = self.apply_displacements(pos, disp) if density_projection else np.array(pos, copy=True)
        out.update(new_vel=new_vel, new_pos=new_pos, valid=valid, active=active)
        return out
