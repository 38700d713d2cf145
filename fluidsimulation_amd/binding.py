"""ctypes binding of libflip_hip.so (include/flip_hip.h).

This is plumbing for the tests and the benchmark: it moves numpy arrays across the C ABI and nothing
else.  There is no fallback: if the shared library is missing, or no HIP device is present, the calls
raise.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libflip_hip.so")


class FlipError(RuntimeError):
    pass


class GridDesc(C.Structure):
    _fields_ = [("nx", C.c_int32), ("ny", C.c_int32), ("nz", C.c_int32), ("ox", C.c_int32), ("oy", C.c_int32),
                ("oz", C.c_int32), ("dx", C.c_float), ("phi_fluid_bg", C.c_float), ("phi_solid_bg", C.c_float),
                ("rest_density", C.c_float)]


class PcgInfo(C.Structure):
    _fields_ = [("converged", C.c_int32), ("iterations", C.c_int32), ("rows", C.c_int32), ("cg_rows", C.c_int32),
                ("active_chunks", C.c_int32), ("rel_residual_2", C.c_double), ("rhs_norm_2", C.c_double),
                ("seconds", C.c_double), ("spmv_seconds", C.c_double), ("spmv_launches", C.c_int32),
                ("dependent_rows", C.c_int32), ("iterations_inf", C.c_int32), ("rel_residual_inf", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# enum flip_field: name -> (id, dtype)
FIELDS = {
    "phi_fluid": (0, np.float32), "phi_solid": (1, np.float32),
    "vel_u": (2, np.float64), "vel_v": (3, np.float64), "vel_w": (4, np.float64),
    "prev_u": (5, np.float64), "prev_v": (6, np.float64), "prev_w": (7, np.float64),
    "diff_u": (8, np.float64), "diff_v": (9, np.float64), "diff_w": (10, np.float64),
    "disp_x": (11, np.float64), "disp_y": (12, np.float64), "disp_z": (13, np.float64),
    "weight_u": (14, np.float32), "weight_v": (15, np.float32), "weight_w": (16, np.float32),
    "fluid_weight_u": (17, np.float32), "fluid_weight_v": (18, np.float32), "fluid_weight_w": (19, np.float32),
    "density": (20, np.float32), "pressure": (21, np.float64),
    "valid_u": (22, np.uint8), "valid_v": (23, np.uint8), "valid_w": (24, np.uint8),
    "active": (25, np.uint8), "label": (26, np.uint8), "row_index": (27, np.int32), "band": (28, np.uint8),
    "obj_weight_u": (29, np.float32), "obj_weight_v": (30, np.float32), "obj_weight_w": (31, np.float32),
    "phi_object": (32, np.float32),
    "vel_back_u": (33, np.float64), "vel_back_v": (34, np.float64), "vel_back_w": (35, np.float64),
    "valid_back_u": (36, np.uint8), "valid_back_v": (37, np.uint8), "valid_back_w": (38, np.uint8),
}

SOLVE_DIVERGENCE, SOLVE_DENSITY = 0, 1
PRECOND_JACOBI, PRECOND_MULTIGRID = 0, 1

# every symbol include/flip_hip.h declares
EXPORTS = [
    "flip_ctx_create", "flip_ctx_destroy", "flip_get_stream", "flip_system_dependent_coupled", "flip_last_error", "flip_synchronize", "flip_get_desc",
    "flip_particles_upload", "flip_particles_download", "flip_particles_download_mass", "flip_particles_count", "flip_bin_particles",
    "flip_binning_download", "flip_field_upload", "flip_field_download", "flip_field_bytes", "flip_p2g", "flip_gather_stats", "flip_set_gather_mode", "flip_g2p",
    "flip_apply_displacements", "flip_advect_particles", "flip_constrain_velocity", "flip_fluid_levelset_from_particles", "flip_mark_valid_from_active", "flip_extrapolate", "flip_add_forces",
    "flip_update_prev", "flip_update_diff", "flip_swap_velocity_buffers", "flip_swap_valid_masks", "flip_velocity_max", "flip_velocity_max_static", "flip_index_cells", "flip_face_fractions",
    "flip_density", "flip_redistribute_overdense", "flip_solid_object_set", "flip_solid_object_velocity", "flip_build_system",
    "flip_update_particles", "flip_set_system_options", "flip_set_warm_start", "flip_system_download", "flip_pcg_solve", "flip_apply_operator", "flip_scan_i32", "flip_set_profiling", "flip_solver_kernel_times", "flip_solve_pair", "flip_select_system", "flip_pair_begin", "flip_pair_select", "flip_pair_end", "flip_set_side_stream", "flip_side_fork", "flip_side_select", "flip_side_join", "flip_project_velocity",
    "flip_project_displacement", "flip_set_slab", "flip_solid_coupling_partial", "flip_solid_velocity_apply", "flip_pcg_dist_buffer", "flip_pcg_dist_begin", "flip_pcg_dist_direction",
    "flip_pcg_dist_update", "flip_pcg_dist_status", "flip_pcg_dist_finish", "flip_set_slab_overlap",
    "flip_pcg_dist_precondition", "flip_pcg_dist_combine", "flip_particles_reserve", "flip_particles_pack",
    "flip_particles_exchange_apply",
]

_lib = None


def _one_hip_runtime():
    """A process can drive the device through one HIP/HSA runtime only.  PyTorch-ROCm wheels carry their own copy and load
    it by path: imported after libflip_hip.so has pulled in the system's copy, torch brings a second runtime that finds no
    device ("No HIP GPUs are available"), whereas a library loaded after torch resolves its libamdhip64 dependency to the
    copy already in the process.  The distributed drivers of this package need torch (device tensors, torch.distributed),
    so where torch is installed it is imported first, whatever the order the caller's own imports come in.
    FLIP_NO_TORCH_PRELOAD=1 skips this for a process that will never import torch."""
    if "torch" in sys.modules or os.environ.get("FLIP_NO_TORCH_PRELOAD") == "1":
        return
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def load_library(path: str = LIB_PATH):
    """Load libflip_hip.so.  Raises FlipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise FlipError(f"{path} not found: build it with `make -C fluidsimulation_amd/csrc` "
                        "(there is no CPU fallback for the HIP path)")
    _one_hip_runtime()
    lib = C.CDLL(path)
    vp, i32, i64, f32, f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double
    P = C.POINTER
    sig = {
        "flip_ctx_create": (C.c_int, [P(GridDesc), C.c_int, vp, P(vp)]),
        "flip_ctx_destroy": (None, [vp]),
        "flip_last_error": (C.c_char_p, [vp]),
        "flip_synchronize": (C.c_int, [vp]),
        "flip_get_desc": (C.c_int, [vp, P(GridDesc)]),
        "flip_get_stream": (C.c_int, [vp, P(vp), P(C.c_int)]),
        "flip_system_dependent_coupled": (C.c_int, [vp, P(i32)]),
        "flip_particles_upload": (C.c_int, [vp, i64, vp, vp, vp]),
        "flip_particles_download": (C.c_int, [vp, vp, vp]),
        "flip_particles_download_mass": (C.c_int, [vp, vp]),
        "flip_particles_count": (i64, [vp]),
        "flip_bin_particles": (C.c_int, [vp, P(i64)]),
        "flip_binning_download": (C.c_int, [vp, vp, vp, vp]),
        "flip_field_upload": (C.c_int, [vp, C.c_int, vp]),
        "flip_field_download": (C.c_int, [vp, C.c_int, vp]),
        "flip_field_bytes": (i64, [vp, C.c_int]),
        "flip_p2g": (C.c_int, [vp]),
        "flip_gather_stats": (C.c_int, [vp, P(i64)]),
        "flip_set_gather_mode": (C.c_int, [vp, C.c_int]),
        "flip_g2p": (C.c_int, [vp, f32]),
        "flip_apply_displacements": (C.c_int, [vp]),
        "flip_advect_particles": (C.c_int, [vp, C.c_float]),
        "flip_constrain_velocity": (C.c_int, [vp, C.c_int]),
        "flip_fluid_levelset_from_particles": (C.c_int, [vp, C.c_float, C.c_float]),
        "flip_mark_valid_from_active": (C.c_int, [vp]),
        "flip_extrapolate": (C.c_int, [vp, C.c_int]),
        "flip_add_forces": (C.c_int, [vp, f32]),
        "flip_update_prev": (C.c_int, [vp]),
        "flip_update_diff": (C.c_int, [vp]),
        "flip_swap_velocity_buffers": (C.c_int, [vp]),
        "flip_swap_valid_masks": (C.c_int, [vp]),
        "flip_velocity_max": (C.c_int, [vp, P(f64)]),
        "flip_velocity_max_static": (C.c_int, [vp, P(f64)]),
        "flip_index_cells": (C.c_int, [vp, P(i32)]),
        "flip_face_fractions": (C.c_int, [vp]),
        "flip_density": (C.c_int, [vp, P(i64)]),
        "flip_redistribute_overdense": (C.c_int, [vp, C.c_uint32, P(i64), P(i64)]),
        "flip_solid_object_set": (C.c_int, [vp, vp, f32, vp, vp, vp]),
        "flip_solid_object_velocity": (C.c_int, [vp, P(f64)]),
        "flip_build_system": (C.c_int, [vp, C.c_int, f32]),
        "flip_set_system_options": (C.c_int, [vp, C.c_int, C.c_int]),
        "flip_set_warm_start": (C.c_int, [vp, C.c_int]),
        "flip_pair_begin": (C.c_int, [vp]),
        "flip_pair_select": (C.c_int, [vp, C.c_int]),
        "flip_pair_end": (C.c_int, [vp]),
        "flip_set_side_stream": (C.c_int, [vp, C.c_int]),
        "flip_side_fork": (C.c_int, [vp]),
        "flip_side_select": (C.c_int, [vp, C.c_int]),
        "flip_side_join": (C.c_int, [vp]),
        "flip_update_particles": (C.c_int, [vp, f32, f32, C.c_int, C.c_int]),
        "flip_system_download": (C.c_int, [vp, vp, vp, vp]),
        "flip_pcg_solve": (C.c_int, [vp, f64, i32, C.c_int, P(PcgInfo)]),
        "flip_apply_operator": (C.c_int, [vp, vp, vp]),
        "flip_scan_i32": (C.c_int, [vp, vp, C.c_int64, C.c_int32]),
        "flip_set_profiling": (C.c_int, [vp, C.c_int]),
        "flip_solver_kernel_times": (C.c_int, [vp, P(f64)]),
        "flip_solve_pair": (C.c_int, [vp, f32, f64, i32, C.c_int, P(PcgInfo), P(PcgInfo)]),
        "flip_select_system": (C.c_int, [vp, C.c_int]),
        "flip_project_velocity": (C.c_int, [vp, f32]),
        "flip_project_displacement": (C.c_int, [vp, f32]),
        "flip_set_slab": (C.c_int, [vp, i32, i32]),
        "flip_solid_coupling_partial": (C.c_int, [vp, P(f64)]),
        "flip_solid_velocity_apply": (C.c_int, [vp, f32, P(f64)]),
        "flip_pcg_dist_buffer": (C.c_int, [vp, C.c_int, P(vp), P(i64)]),
        "flip_pcg_dist_begin": (C.c_int, [vp, f64, i32, C.c_int]),
        "flip_pcg_dist_direction": (C.c_int, [vp]),
        "flip_pcg_dist_update": (C.c_int, [vp]),
        "flip_pcg_dist_status": (C.c_int, [vp, P(i32), P(i32)]),
        "flip_pcg_dist_finish": (C.c_int, [vp, P(PcgInfo)]),
        "flip_set_slab_overlap": (C.c_int, [vp, i32]),
        "flip_pcg_dist_precondition": (C.c_int, [vp]),
        "flip_pcg_dist_combine": (C.c_int, [vp]),
        "flip_particles_reserve": (C.c_int, [vp, i64]),
        "flip_particles_pack": (C.c_int, [vp, i32, i32, vp, i64, P(i64)]),
        "flip_particles_exchange_apply": (C.c_int, [vp, vp, i64, vp, i64]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class FlipContext:
    """One device context: the dense MAC grid of a FluidDomain plus its ParticleSet, resident in HBM."""

    def __init__(self, desc: dict, device: int = 0, stream: int | None = None):
        self.lib = load_library()
        self.desc = GridDesc(**desc)
        self.dims = (desc["nx"], desc["ny"], desc["nz"])
        self.shape = (desc["nz"], desc["ny"], desc["nx"])
        self.n_cells = desc["nx"] * desc["ny"] * desc["nz"]
        h = C.c_void_p()
        rc = self.lib.flip_ctx_create(C.byref(self.desc), device, C.c_void_p(stream or 0), C.byref(h))
        if rc != 0:
            raise FlipError(self.lib.flip_last_error(None).decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.flip_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc):
        if rc != 0:
            raise FlipError(self.lib.flip_last_error(self.h).decode())

    # ---- particles ----
    def upload_particles(self, pos, vel, mass):
        pos = np.ascontiguousarray(pos, dtype=np.float64)
        vel = np.ascontiguousarray(vel, dtype=np.float64)
        mass = np.ascontiguousarray(mass, dtype=np.float32)
        n = pos.shape[0]
        assert pos.shape == (n, 3) and vel.shape == (n, 3) and mass.shape == (n,)
        self._check(self.lib.flip_particles_upload(self.h, n, _ptr(pos), _ptr(vel), _ptr(mass)))

    def download_particles(self):
        n = self.lib.flip_particles_count(self.h)
        pos = np.empty((n, 3), dtype=np.float64)
        vel = np.empty((n, 3), dtype=np.float64)
        self._check(self.lib.flip_particles_download(self.h, _ptr(pos), _ptr(vel)))
        return pos, vel

    def download_mass(self):
        m = np.zeros(self.particle_count(), np.float32)
        self._check(self.lib.flip_particles_download_mass(self.h, _ptr(m)))
        return m

    def bin_particles(self, want_outside: bool = True) -> int:
        out = C.c_int64(0)
        self._check(self.lib.flip_bin_particles(self.h, C.byref(out) if want_outside else None))
        return out.value

    def download_binning(self):
        n = self.lib.flip_particles_count(self.h)
        cell_id = np.empty(n, dtype=np.int32)
        cell_start = np.empty(self.n_cells + 1, dtype=np.int32)
        perm = np.empty(n, dtype=np.int32)
        self._check(self.lib.flip_binning_download(self.h, _ptr(cell_id), _ptr(cell_start), _ptr(perm)))
        return cell_id, cell_start, perm

    # ---- fields ----
    def upload(self, name: str, arr):
        fid, dt = FIELDS[name]
        a = np.ascontiguousarray(arr, dtype=dt)
        assert a.size == self.n_cells, (name, a.shape, self.shape)
        self._check(self.lib.flip_field_upload(self.h, fid, _ptr(a)))

    def download(self, name: str):
        fid, dt = FIELDS[name]
        a = np.empty(self.shape, dtype=dt)
        self._check(self.lib.flip_field_download(self.h, fid, _ptr(a)))
        return a

    # ---- stages ----
    def p2g(self):
        self._check(self.lib.flip_p2g(self.h))

    def gather_stats(self):
        out = (C.c_int64 * 2)()
        self._check(self.lib.flip_gather_stats(self.h, out))
        return {"staged_tiles": int(out[0]), "unstaged_tiles": int(out[1])}

    def system_dependent_coupled(self) -> int:
        n = C.c_int32(0)
        self._check(self.lib.flip_system_dependent_coupled(self.h, C.byref(n)))
        return int(n.value)

    def set_gather_mode(self, mode: int):
        self._check(self.lib.flip_set_gather_mode(self.h, int(mode)))

    def g2p(self, ratio: float):
        self._check(self.lib.flip_g2p(self.h, ratio))

    def apply_displacements(self):
        self._check(self.lib.flip_apply_displacements(self.h))

    def update_particles(self, ratio: float, dt: float, advect: bool, displace: bool):
        """g2p, advect_particles and apply_displacements in one pass (see flip_update_particles)."""
        self._check(self.lib.flip_update_particles(self.h, ratio, dt, 1 if advect else 0, 1 if displace else 0))

    def advect_particles(self, dt):
        self._check(self.lib.flip_advect_particles(self.h, float(dt)))

    def fluid_levelset_from_particles(self, radius_world: float, half_width_voxels: float = 3.0):
        self._check(self.lib.flip_fluid_levelset_from_particles(self.h, float(radius_world), float(half_width_voxels)))

    def constrain_velocity(self, rigid_body: bool = False):
        self._check(self.lib.flip_constrain_velocity(self.h, 1 if rigid_body else 0))

    def mark_valid_from_active(self):
        self._check(self.lib.flip_mark_valid_from_active(self.h))

    def extrapolate(self, iterations: int = 2):
        self._check(self.lib.flip_extrapolate(self.h, iterations))

    def add_forces(self, dt: float):
        self._check(self.lib.flip_add_forces(self.h, dt))

    def update_prev(self):
        self._check(self.lib.flip_update_prev(self.h))

    def update_diff(self):
        self._check(self.lib.flip_update_diff(self.h))

    def swap_velocity_buffers(self):
        self._check(self.lib.flip_swap_velocity_buffers(self.h))

    def swap_valid_masks(self):
        self._check(self.lib.flip_swap_valid_masks(self.h))

    def velocity_max(self):
        out = (C.c_double * 3)()
        self._check(self.lib.flip_velocity_max(self.h, out))
        return np.array(out[:], dtype=np.float64)

    def velocity_max_static(self):
        """flip_velocity_max through the entry point that allocates nothing (static device workspace)."""
        out = (C.c_double * 3)()
        self._check(self.lib.flip_velocity_max_static(self.h, out))
        return np.array(out[:], dtype=np.float64)

    def index_cells(self, want_count: bool = True) -> int:
        """Number of labelled cells; with want_count=False nothing is read back (no host wait) and -1 is returned."""
        n = C.c_int32(-1)
        self._check(self.lib.flip_index_cells(self.h, C.byref(n) if want_count else None))
        return n.value

    def face_fractions(self):
        self._check(self.lib.flip_face_fractions(self.h))

    def density(self, want_count: bool = True) -> int:
        n = C.c_int64(0)
        self._check(self.lib.flip_density(self.h, C.byref(n) if want_count else None))
        return n.value

    def redistribute_overdense(self, seed: int):
        """(cells touched, particles moved) by the re-seeding of over-dense cells (src/Functors.cpp:268-322)."""
        a, b = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.flip_redistribute_overdense(self.h, int(seed) & 0xffffffff, C.byref(a), C.byref(b)))
        return a.value, b.value

    def set_solid_object(self, phi_object, bg, com, minv, vel):
        if phi_object is None:
            self._check(self.lib.flip_solid_object_set(self.h, None, 0.0, None, None, None))
            return
        phi = np.ascontiguousarray(phi_object, dtype=np.float32)
        com = np.ascontiguousarray(com, dtype=np.float64)
        minv = np.ascontiguousarray(minv, dtype=np.float64)
        vel = np.ascontiguousarray(vel, dtype=np.float64)
        assert phi.size == self.n_cells and com.size == 3 and minv.size == 6 and vel.size == 6
        self._check(self.lib.flip_solid_object_set(self.h, _ptr(phi), bg, _ptr(com), _ptr(minv), _ptr(vel)))

    def solid_object_velocity(self):
        out = (C.c_double * 6)()
        self._check(self.lib.flip_solid_object_velocity(self.h, out))
        return np.array(out[:], dtype=np.float64)

    def build_system(self, kind: int, dt: float):
        self._check(self.lib.flip_build_system(self.h, kind, dt))

    def set_system_options(self, drop_null_rows: bool = True, eliminate_dependent_rows: bool = True):
        self._check(self.lib.flip_set_system_options(self.h, 1 if drop_null_rows else 0,
                                                     1 if eliminate_dependent_rows else 0))

    def set_warm_start(self, on: bool = True):
        """Start each solve from the last converged solution of the same kind (see flip_set_warm_start)."""
        self._check(self.lib.flip_set_warm_start(self.h, 1 if on else 0))

    def download_system(self):
        off = np.empty((6,) + self.shape, dtype=np.float64)
        diag = np.empty(self.shape, dtype=np.float64)
        rhs = np.empty(self.shape, dtype=np.float64)
        self._check(self.lib.flip_system_download(self.h, _ptr(off), _ptr(diag), _ptr(rhs)))
        return off, diag, rhs

    def pcg_solve(self, tol: float = 1e-15, max_iter: int = 0, precond: int = PRECOND_JACOBI) -> dict:
        info = PcgInfo()
        self._check(self.lib.flip_pcg_solve(self.h, tol, max_iter, precond, C.byref(info)))
        return info.as_dict()

    def apply_operator(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        assert x.size == self.n_cells
        y = np.empty(self.shape, dtype=np.float64)
        self._check(self.lib.flip_apply_operator(self.h, _ptr(x), _ptr(y)))
        return y

    def scan_i32(self, values, offset: int = 0):
        """Exclusive prefix sum of int32 values by the device scan (parity hook); `offset` shifts the data off a 16-byte boundary."""
        v = np.ascontiguousarray(values, dtype=np.int32).copy()
        self._check(self.lib.flip_scan_i32(self.h, _ptr(v), v.size, offset))
        return v

    def solve_pair(self, dt: float, tol: float = 1e-15, max_iter: int = 0, precond: int = PRECOND_JACOBI):
        """Assembles and solves the divergence and the density system side by side; returns their two info dicts."""
        a, b = PcgInfo(), PcgInfo()
        self._check(self.lib.flip_solve_pair(self.h, dt, tol, max_iter, precond, C.byref(a), C.byref(b)))
        return a.as_dict(), b.as_dict()

    def select_system(self, kind: int):
        self._check(self.lib.flip_select_system(self.h, kind))

    # the context's second stream for stages of a substep that do not depend on each other (include/flip_hip.h)
    def set_side_stream(self, on: bool):
        self._check(self.lib.flip_set_side_stream(self.h, 1 if on else 0))

    def side_fork(self):
        self._check(self.lib.flip_side_fork(self.h))

    def side_select(self, side: bool):
        self._check(self.lib.flip_side_select(self.h, 1 if side else 0))

    def side_join(self):
        self._check(self.lib.flip_side_join(self.h))

    def set_profiling(self, on: bool):
        self._check(self.lib.flip_set_profiling(self.h, 1 if on else 0))

    def solver_kernel_times(self):
        """Microseconds of the level-0 kernels of one profiled solver iteration (see flip_solver_kernel_times)."""
        out = (C.c_double * 6)()
        self._check(self.lib.flip_solver_kernel_times(self.h, out))
        return dict(zip(("k_pcg_direction", "k_pcg_update_pre", "k_mg_restrict", "coarse_levels", "k_mg_post", "k_mg_smooth"), out[:]))

    def project_velocity(self, dt: float):
        self._check(self.lib.flip_project_velocity(self.h, dt))

    def project_displacement(self, dt: float):
        self._check(self.lib.flip_project_displacement(self.h, dt))

    def synchronize(self):
        self._check(self.lib.flip_synchronize(self.h))

    # ---- particle exchange of the slab decomposition (device buffers given as integer addresses) ----
    def reserve_particles(self, capacity: int):
        self._check(self.lib.flip_particles_reserve(self.h, int(capacity)))

    def pack_particles(self, plane_lo: int, plane_hi: int, device_ptr: int, capacity: int) -> int:
        n = C.c_int64(0)
        self._check(self.lib.flip_particles_pack(self.h, int(plane_lo), int(plane_hi), C.c_void_p(device_ptr), int(capacity),
                                                 C.byref(n)))
        return n.value

    def exchange_apply(self, recv_lo_ptr: int, n_lo: int, recv_hi_ptr: int, n_hi: int):
        self._check(self.lib.flip_particles_exchange_apply(self.h, C.c_void_p(recv_lo_ptr), int(n_lo), C.c_void_p(recv_hi_ptr),
                                                           int(n_hi)))

    def particle_count(self) -> int:
        return int(self.lib.flip_particles_count(self.h))

    # ---- z-slab decomposition of the solve ----
    def set_slab(self, k0: int, k1: int):
        self._check(self.lib.flip_set_slab(self.h, k0, k1))

    def solid_coupling_partial(self):
        out = (C.c_double * 6)()
        self._check(self.lib.flip_solid_coupling_partial(self.h, out))
        return np.array(out[:], dtype=np.float64)

    def solid_velocity_apply(self, dt: float, jp_total):
        arr = (C.c_double * 6)(*[float(v) for v in jp_total])
        self._check(self.lib.flip_solid_velocity_apply(self.h, dt, arr))

    def set_slab_overlap(self, planes: int):
        self._check(self.lib.flip_set_slab_overlap(self.h, planes))

    def dist_precondition(self):
        self._check(self.lib.flip_pcg_dist_precondition(self.h))

    def dist_combine(self):
        self._check(self.lib.flip_pcg_dist_combine(self.h))

    def dist_buffer(self, which: int):
        """(device pointer, element count) of buffer 0 (sums), 1 (z), 2 (x), 3 (r), 4 / 5 (corrections received from
        the lower / upper neighbour, overlapping variant only); elements are doubles."""
        ptr, n = C.c_void_p(), C.c_int64(0)
        self._check(self.lib.flip_pcg_dist_buffer(self.h, which, C.byref(ptr), C.byref(n)))
        return ptr.value, n.value

    def dist_begin(self, tol: float, max_iter: int, precond: int):
        self._check(self.lib.flip_pcg_dist_begin(self.h, tol, max_iter, precond))

    def dist_direction(self):
        self._check(self.lib.flip_pcg_dist_direction(self.h))

    def dist_update(self):
        self._check(self.lib.flip_pcg_dist_update(self.h))

    def dist_status(self):
        done, it = C.c_int32(0), C.c_int32(0)
        self._check(self.lib.flip_pcg_dist_status(self.h, C.byref(done), C.byref(it)))
        return done.value, it.value

    def dist_finish(self) -> dict:
        info = PcgInfo()
        self._check(self.lib.flip_pcg_dist_finish(self.h, C.byref(info)))
        return info.as_dict()

    # ---- convenience ----
    def upload_scene(self, sc):
        self.upload("phi_solid", sc.phi_solid)
        self.upload("phi_fluid", sc.phi_fluid)
        self.upload_particles(sc.pos, sc.vel, sc.mass)
        if sc.phi_object is not None:
            self.set_solid_object(sc.phi_object, sc.phi_object_bg, sc.object_com, sc.object_minv, sc.object_vel)
