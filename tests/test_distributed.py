"""z-slab decomposition (fluidsimulation_amd/distributed.py), world size 2.

CPU part (gloo, no GPU): each rank assembles the system of its extended slab with the CPU oracle, checks that
its owned planes carry exactly the global system's coefficients (the ghost planes absorb every effect of the
artificial slab boundary), and runs the production communication loop `slab_pcg_solve` with a numpy stand-in
for the device kernels.  The joined solution must equal the single-domain solve.

GPU part: the same two ranks share the one GPU of the box (gloo with host staging, because RCCL refuses two
ranks on one device) and run the real split-phase kernels; particle state after a whole substep must match the
single-context run to the task's tolerances.
"""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, fresh_state

WORLD = 2
GHOST = 6


def _assemble(o, sc, dt, kind=0):
    """First-substep assembly with the oracle's stages on whatever box `o` describes."""
    z32 = lambda: np.zeros(o.shape, np.float32)  # noqa: E731
    _, cell_start, perm, _ = o.bin_particles(sc.pos)
    pm1, pm2 = o.particle_mask(cell_start, 1), o.particle_mask(cell_start, 2)
    uvw = o.p2g(sc.pos, sc.vel, cell_start, perm, pm2)
    valid = [pm2.copy() for _ in range(3)]
    uvw, valid, active = o.extrapolate(uvw, valid, pm2.copy(), 2)
    uvw = [uvw[0], o.add_gravity(uvw[1], active, dt), uvw[2]]
    label, index, n_rows = o.index_cells(pm1, sc.phi_fluid, sc.phi_solid, [z32(), z32(), z32()])
    wm = o.weight_mask(sc.phi_fluid, sc.phi_fluid_bg)
    w = o.face_fractions(sc.phi_solid, sc.phi_solid_bg, wm)
    o.combine_weights(w, None)
    dens, _ = o.density(sc.pos, sc.mass, cell_start, perm, pm2, sc.phi_fluid)
    off, diag, rhs = o.build_system(kind, dt, label, sc.phi_fluid, sc.phi_solid, w, uvw=uvw, density=dens)
    return label, off, diag, rhs


def _scene():
    from fluidsimulation_amd import scenes
    sc = scenes.cube_drop(16, ppc=4, seed=3)
    rng = np.random.default_rng(11)
    sc.vel = rng.normal(0.0, 0.25, sc.pos.shape)
    return sc


class NumpySlabSolver:
    """What the split-phase kernels compute, in numpy, for the communication-loop test: preconditioned CG on the
    rows of the owned planes; ghost planes hold the neighbour's values.  overlap = 0: point Jacobi.  overlap = m:
    additive Schwarz, every rank applying the symmetric polynomial (2 I - D^-1 A) D^-1 of its own matrix restricted
    to planes [k0 - m, k1 + m)."""

    def __init__(self, layout, rank, label, off, diag, rhs, overlap=0):
        import torch
        from fluidsimulation_amd.distributed import SlabExchange
        self._ex = SlabExchange()
        self._ex._init_slab(layout, rank)
        self.k0, self.k1 = self._ex.k0, self._ex.k1
        nz, ny, nx = label.shape
        self.shape = label.shape
        self.own = np.zeros(label.shape, bool)
        self.own[self.k0:self.k1] = label[self.k0:self.k1] > 0
        self.off, self.diag, self.rhs = off, diag, rhs
        self.xa = np.zeros(label.shape)
        self.za = np.zeros(label.shape)
        self.ra = np.zeros(label.shape)
        self._ex.x = torch.from_numpy(self.xa).view(nz, ny * nx)
        self._ex.z = torch.from_numpy(self.za).view(nz, ny * nx)
        self._ex.r = torch.from_numpy(self.ra).view(nz, ny * nx)
        self.sums = torch.zeros(8, dtype=torch.float64)
        self.overlap = self._ex.overlap = overlap if layout.world > 1 else 0
        if self.overlap:
            m = self.overlap
            self.region = np.zeros(label.shape, bool)
            lo, hi = max(self.k0 - m, 0), min(self.k1 + m, nz)
            self.region[lo:hi] = label[lo:hi] > 0
            self.recv_lo, self.recv_hi = np.zeros((m + 1, ny, nx)), np.zeros((m + 1, ny, nx))
            self._ex.recv_lo = torch.from_numpy(self.recv_lo).view(m + 1, ny * nx)
            self._ex.recv_hi = torch.from_numpy(self.recv_hi).view(m + 1, ny * nx)

    def exchange_z(self, comm):
        self._ex.exchange_z(comm)

    def exchange_x(self, comm):
        self._ex.exchange_x(comm)

    def exchange_r(self, comm):
        self._ex.exchange_r(comm)

    def exchange_corrections(self, comm):
        self._ex.exchange_corrections(comm)

    def _apply(self, p, rows=None):
        rows = self.own if rows is None else rows
        pp = np.pad(p, 1)
        nb = [pp[1:-1, 1:-1, :-2], pp[1:-1, 1:-1, 2:], pp[1:-1, :-2, 1:-1], pp[1:-1, 2:, 1:-1], pp[:-2, 1:-1, 1:-1], pp[2:, 1:-1, 1:-1]]
        y = self.diag * p
        for f in range(6):
            y = y + self.off[f] * nb[f]
        return np.where(rows, y, 0.0)

    def begin(self, tol, max_iter, precond):
        self.tol2, self.max_iter = tol * tol, max_iter
        self.it, self.done = 0, 0
        self.ra[...] = np.where(self.own, self.rhs, 0.0)
        self.inv = np.where(self.own & (self.diag != 0), 1.0 / np.where(self.diag != 0, self.diag, 1.0), 0.0)
        self.za[...] = self.inv * self.ra
        self.xa[...] = 0.0
        self.p = np.zeros(self.shape)
        self.sums[:] = 0
        self.sums[0], self.sums[1] = float((self.ra * self.ra).sum()), float((self.ra * self.za).sum())

    def precondition(self):
        if self.done:
            return
        inv = np.where(self.region & (self.diag != 0), 1.0 / np.where(self.diag != 0, self.diag, 1.0), 0.0)
        t = inv * np.where(self.region, self.ra, 0.0)
        self.za[...] = 2.0 * t - inv * self._apply(t, self.region)

    def combine(self):
        if self.done:
            return
        m = self.overlap
        if self._ex.g_lo:
            sl = slice(self.k0 - 1, self.k0 + m)
            self.za[sl] += np.where(self.region[sl], self.recv_lo, 0.0)
        if self._ex.g_hi:
            sl = slice(self.k1 - m, self.k1 + 1)
            self.za[sl] += np.where(self.region[sl], self.recv_hi, 0.0)
        r = np.where(self.own, self.ra, 0.0)
        self.sums[:] = 0
        self.sums[0], self.sums[1] = float((r * r).sum()), float((r * self.za).sum())

    def direction(self):
        if self.done:
            return
        rr, rz = float(self.sums[0]), float(self.sums[1])
        if self.it == 0:
            self.thr = max(self.tol2 * rr, np.finfo(np.float64).tiny)
            if rr == 0.0 or rr < self.thr:
                self.done = 1
                return
            beta = 0.0
        else:
            if rr < self.thr or self.it >= self.max_iter:
                self.done = 1 if rr < self.thr else 2
                return
            beta = rz / self.rz_prev
        self.rz_prev = rz
        ghost = np.zeros(self.shape, bool)
        if self._ex.g_lo:
            ghost[self.k0 - 1] = True
        if self._ex.g_hi:
            ghost[self.k1] = True
        self.p = np.where(self.own | ghost, self.za + beta * self.p, 0.0)
        self.za[ghost] = 0.0
        self.ap = self._apply(self.p)
        self.sums[:] = 0
        self.sums[2] = float((np.where(self.own, self.p, 0.0) * self.ap).sum())

    def update(self):
        if self.done:
            return
        alpha = self.rz_prev / float(self.sums[2])
        self.xa += alpha * np.where(self.own, self.p, 0.0)
        self.ra[...] = np.where(self.own, self.ra - alpha * self.ap, 0.0)
        self.za[...] = self.inv * self.ra
        self.sums[:] = 0
        self.sums[0], self.sums[1] = float((self.ra * self.ra).sum()), float((self.ra * self.za).sum())
        self.it += 1

    def status(self):
        return self.done, self.it

    def finish(self):
        return {"converged": int(self.done == 1), "iterations": self.it}


def _cpu_layout(nz, world):
    from fluidsimulation_amd.distributed import SlabLayout
    if world == WORLD:
        return SlabLayout(nz, world, GHOST)
    # more ranks: uneven slabs cut by load, thinner ghost zone
    w = np.full(nz, 0.2)
    w[5:-5] = 1.0
    w[: nz // 2] *= 1.7
    return SlabLayout.balanced(w, world, 2)


def _cpu_worker(rank, port, out_dir, overlap, world=WORLD):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from fluidsimulation_amd.distributed import Comm, SlabLayout, partition_scene, slab_pcg_solve
    from oracle import Oracle
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        dt = 0.004
        sc = _scene()
        layout = _cpu_layout(sc.dims[2], world)
        loc = partition_scene(sc, layout, rank)
        label, off, diag, rhs = _assemble(Oracle(loc.desc(), threads=2), loc, dt)
        # the owned planes of the local assembly are the global assembly, bit for bit
        g_label, g_off, g_diag, g_rhs = _assemble(Oracle(sc.desc(), threads=2), sc, dt)
        o0, o1 = layout.owned(rank)
        k0, k1 = layout.local_owned(rank)
        assert np.array_equal(label[k0:k1], g_label[o0:o1])
        assert np.array_equal(diag[k0:k1], g_diag[o0:o1]) and np.array_equal(rhs[k0:k1], g_rhs[o0:o1])
        assert np.array_equal(off[:, k0:k1], g_off[:, o0:o1])
        solver = NumpySlabSolver(layout, rank, label, off, diag, rhs, overlap=overlap)
        info = slab_pcg_solve(solver, Comm(), 1e-13, 5000, 0, check_every=4)
        assert info["converged"] == 1
        # after exchange_x the ghost planes hold the neighbour's solution
        np.save(os.path.join(out_dir, f"x_{rank}.npy"), solver.xa)
        np.save(os.path.join(out_dir, f"it_{rank}.npy"), np.array([info["iterations"]]))
    finally:
        dist.destroy_process_group()


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_slab_layout_and_partition():
    from fluidsimulation_amd.distributed import SlabLayout, partition_scene
    lay = SlabLayout(40, 4, 6)
    assert [lay.owned(r) for r in range(4)] == [(0, 10), (10, 20), (20, 30), (30, 40)]
    assert lay.extended(0) == (0, 16) and lay.extended(2) == (14, 36) and lay.local_owned(2) == (6, 16)
    with pytest.raises(AssertionError):
        SlabLayout(40, 8, 6)     # slabs thinner than the ghost zone
    # cut by load: the planes inside the walls weigh little, every slab stays at least as thick as the ghost zone
    w = np.full(138, 0.1)
    w[5:-5] = 1.0
    bal = SlabLayout.balanced(w, 8, 8)
    thick = [bal.owned(r)[1] - bal.owned(r)[0] for r in range(8)]
    assert sum(thick) == 138 and min(thick) >= 8 and thick[0] > thick[3] and thick[-1] > thick[3]
    loads = [w[bal.owned(r)[0]:bal.owned(r)[1]].sum() for r in range(8)]
    assert max(loads) - min(loads) <= 2.0
    assert [SlabLayout.balanced(np.ones(16), 2, 8).owned(r) for r in range(2)] == [(0, 8), (8, 16)]
    sc = _scene()
    lay = SlabLayout(sc.dims[2], 2, GHOST)
    parts = [partition_scene(sc, lay, r) for r in range(2)]
    owned = np.concatenate([p.global_index[p.owned] for p in parts])
    assert np.array_equal(np.sort(owned), np.arange(sc.pos.shape[0]))     # every particle owned exactly once
    for r, p in enumerate(parts):
        e0, e1 = lay.extended(r)
        assert p.dims[2] == e1 - e0 and p.origin[2] == sc.origin[2] + e0
        assert np.array_equal(p.phi_solid, sc.phi_solid[e0:e1])
        assert (~p.owned).sum() > 0                                       # ghost copies are present


@pytest.mark.parametrize("overlap,world", [(0, WORLD), (3, WORLD), (0, 4), (1, 4)])
def test_slab_solve_over_ranks_matches_the_global_solve(tmp_path, overlap, world):
    """The production communication loop (slab_pcg_solve) over gloo with 2 even or 4 uneven slabs."""
    import torch.multiprocessing as mp
    from oracle import Oracle
    port = _free_port()
    mp.spawn(_cpu_worker, args=(port, str(tmp_path), overlap, world), nprocs=world, join=True)
    sc = _scene()
    o = Oracle(sc.desc(), threads=2)
    label, off, diag, rhs = _assemble(o, sc, 0.004)
    from fluidsimulation_amd.distributed import SlabLayout
    lay = _cpu_layout(sc.dims[2], world)
    if world == 4:
        thick = [lay.owned(r)[1] - lay.owned(r)[0] for r in range(world)]
        assert len(set(thick)) > 1, thick       # the slabs really are uneven
    single = NumpySlabSolver(SlabLayout(sc.dims[2], 1, 0), 0, label, off, diag, rhs)

    class _NoComm:
        world, rank = 1, 0
        def all_reduce_sum(self, t): pass
        def exchange(self, *a): pass
    from fluidsimulation_amd.distributed import slab_pcg_solve
    info = slab_pcg_solve(single, _NoComm(), 1e-13, 5000, 0, check_every=4)
    assert info["converged"] == 1
    joined = np.zeros_like(single.xa)
    its = []
    for r in range(world):
        x = np.load(os.path.join(str(tmp_path), f"x_{r}.npy"))
        its.append(int(np.load(os.path.join(str(tmp_path), f"it_{r}.npy"))[0]))
        o0, o1 = lay.owned(r)
        k0, k1 = lay.local_owned(r)
        e0, e1 = lay.extended(r)
        joined[o0:o1] = x[k0:k1]
        # ghost planes: the neighbour's values arrived
        assert np.allclose(x, single.xa[e0:e1], rtol=0, atol=1e-9 * np.abs(single.xa).max())
    assert len(set(its)) == 1
    if overlap == 0:
        # with point Jacobi the slab iteration is the global iteration up to summation order
        assert abs(its[0] - info["iterations"]) <= 2
    else:
        # the overlapping preconditioner is stronger than point Jacobi
        assert its[0] < info["iterations"]
    assert np.abs(joined - single.xa).max() <= 1e-9 * np.abs(single.xa).max()
    # and it solves the system: residual of the joined vector under the global operator
    res = np.where(label > 0, rhs - single._apply(joined), 0.0)
    assert np.linalg.norm(res) <= 1e-11 * np.linalg.norm(np.where(label > 0, rhs, 0.0))


# ------------------------------------------------------------------------------------------------
# GPU: two ranks on the one device of the box, real kernels, gloo with host staging

def _gpu_worker(rank, port, out_dir, precond, kind, overlap, full=False, driver="python"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from fluidsimulation_amd import FlipContext
    from fluidsimulation_amd.distributed import Comm, DistributedSubstep, SlabLayout, partition_scene
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=WORLD)
    try:
        _gpu_rank_body(rank, out_dir, precond, kind, overlap, full, driver)
    except BaseException:
        import traceback
        with open(os.path.join(out_dir, f"error_{rank}.txt"), "w") as f:
            f.write(traceback.format_exc())
        raise
    finally:
        dist.destroy_process_group()


def _gpu_rank_body(rank, out_dir, precond, kind, overlap, full=False, driver="python"):
    import torch
    from fluidsimulation_amd import FlipContext
    from fluidsimulation_amd.distributed import Comm, DistributedSubstep, SlabLayout, partition_scene
    if True:
        torch.cuda.set_device(0)
        sc = _scene_gpu(kind)
        layout = _gpu_layout(sc, kind)
        loc = partition_scene(sc, layout, rank)
        with FlipContext(loc.desc(), device=0) as ctx:
            ctx.reserve_particles(2 * loc.pos.shape[0])
            ctx.upload_scene(loc)
            if driver == "native":
                # the C driver (libflip_dist.so) with its transport callbacks over gloo: RCCL refuses two ranks on one device
                from fluidsimulation_amd.dist_native import NativeDistributedSubstep, TorchTransport
                comm = Comm()
                step = NativeDistributedSubstep(ctx, layout, comm, device=0, precond=precond, max_iter=20000, overlap=overlap,
                                                transport=TorchTransport(comm, 0), **_extras(sc, full))
            else:
                step = DistributedSubstep(ctx, layout, Comm(), device=0, precond=precond, max_iter=20000, overlap=overlap, **_extras(sc, full))
            info = step.step(0.004)
            if driver == "native":
                # the C driver solves the two systems side by side and leaves the divergence system active; the reference
                # run below solves them one after the other on one set of arrays, which then holds the density solve's result
                from fluidsimulation_amd import SOLVE_DENSITY, SOLVE_DIVERGENCE
                ctx.select_system(SOLVE_DENSITY)
                p = ctx.download("pressure")
                ctx.select_system(SOLVE_DIVERGENCE)
            else:
                p = ctx.download("pressure")
            # after the exchange the set holds this rank's owned particles and fresh ghost copies
            pos, vel = ctx.download_particles()
            o0, o1 = layout.owned(rank)
            plane = np.floor(pos[:, 2] + 0.5).astype(np.int64) - sc.origin[2]
            owned1 = (plane >= o0) & (plane < o1)
            info2 = step.step(0.004)
            pos2, vel2 = ctx.download_particles()
            plane2 = np.floor(pos2[:, 2] + 0.5).astype(np.int64) - sc.origin[2]
            owned2 = (plane2 >= o0) & (plane2 < o1)
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), pos=pos[owned1], vel=vel[owned1], pos2=pos2[owned2], vel2=vel2[owned2],
                 n_ghost=int((~owned1).sum()), sent=np.array(info["exchange"]["sent"]),
                 conv=info["divergence"]["converged"] * info["density"]["converged"]
                 * info2["divergence"]["converged"] * info2["density"]["converged"], p=p,
                 iterations=info["divergence"]["iterations"])


def _extras(sc, full):
    """The optional stages of the whole substep: level set from the particles, constrain_velocity, RK3 advection."""
    if not full:
        return {}
    return dict(constrain=True, advect=True, levelset_radius=1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0)


def _gpu_layout(sc, kind):
    from fluidsimulation_amd.distributed import SlabLayout
    if kind == "full_domain":
        # uneven slabs, cut by load as the strong-scaling benchmark does
        w = np.full(sc.dims[2], 0.1)
        w[5:-5] = 1.0
        w[: sc.dims[2] // 3] *= 2.0
        return SlabLayout.balanced(w, WORLD, 8)
    return SlabLayout(sc.dims[2], WORLD, 8)


def _scene_gpu(kind="dam"):
    from fluidsimulation_amd import scenes
    if kind == "full_domain":
        # the 512^3 configuration's scene (container over the whole nominal grid) at reduced size
        sc = scenes.dam_break_slab(24, ppc=8, seed=12345, full_domain=True)
    elif kind == "long":
        # the benchmark's multi-GPU scene: slab faces at odd planes, so coarse multigrid cells straddle them
        sc = scenes.dam_break_slab(32, ppc=8, seed=12345, z_repeat=2)
    else:
        sc = scenes.dam_break(32, ppc=4, seed=9)
    rng = np.random.default_rng(2)
    sc.vel = rng.normal(0.0, 0.2, sc.pos.shape)
    return sc


@pytest.mark.gpu
@pytest.mark.parametrize("precond,kind,overlap,full,driver",
                         [(0, "dam", 0, False, "python"), (1, "dam", 0, False, "python"), (1, "long", 0, False, "python"),
                          (1, "dam", 8, False, "python"), (1, "long", 5, False, "python"), (0, "dam", 2, False, "python"),
                          (1, "dam", 4, True, "python"), (1, "full_domain", 4, True, "python"),
                          (1, "dam", 0, False, "native"), (1, "dam", 4, True, "native"), (1, "full_domain", 4, True, "native")])
def test_two_ranks_on_one_gpu_match_the_single_context_substep(tmp_path, precond, kind, overlap, full, driver):
    import torch.multiprocessing as mp
    from fluidsimulation_amd import FlipContext, FlipSubstep
    from fluidsimulation_amd.distributed import SlabLayout
    port = _free_port()
    try:
        mp.spawn(_gpu_worker, args=(port, str(tmp_path), precond, kind, overlap, full, driver), nprocs=WORLD, join=True)
    except Exception:
        for r in range(WORLD):
            f = os.path.join(str(tmp_path), f"error_{r}.txt")
            if os.path.exists(f):
                print(f"---- rank {r} ----\n" + open(f).read())
        raise
    sc = _scene_gpu(kind)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        ref_info = FlipSubstep(ctx, precond=precond, max_iter=20000, **_extras(sc, full)).step(0.004)
        ref_pos, ref_vel = ctx.download_particles()
        ref_p = ctx.download("pressure")
    assert ref_info["divergence"]["converged"] == 1
    lay = _gpu_layout(sc, kind)
    from scipy.spatial import cKDTree
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        stepper = FlipSubstep(ctx, precond=precond, max_iter=20000, **_extras(sc, full))
        stepper.step(0.004)
        stepper.step(0.004)
        ref_pos2, ref_vel2 = ctx.download_particles()
    got = {k: [] for k in ("pos", "vel", "pos2", "vel2")}
    for r in range(WORLD):
        d = np.load(os.path.join(str(tmp_path), f"r{r}.npz"))
        assert int(d["conv"]) == 1
        assert int(d["n_ghost"]) > 0 and d["sent"].sum() > 0
        for k in got:
            got[k].append(d[k])
        # pressure on the owned planes, and on the ghost planes after the exchange
        e0, e1 = lay.extended(r)
        scale = np.abs(ref_p).max()
        assert np.abs(d["p"] - ref_p[e0:e1])[1:-1].max() <= 1e-7 * scale
    # the owned particles of the two ranks are the whole set, each exactly once; identity by position
    for tag, rp, rv in (("", ref_pos, ref_vel), ("2", ref_pos2, ref_vel2)):
        pos, vel = np.concatenate(got["pos" + tag]), np.concatenate(got["vel" + tag])
        assert pos.shape == rp.shape
        dist, idx = cKDTree(rp).query(pos, k=1)
        assert dist.max() < 1e-5
        assert np.array_equal(np.sort(idx), np.arange(rp.shape[0]))
        assert np.abs(vel - rv[idx]).max() / np.abs(rv).max() < 1e-4


# ------------------------------------------------------------------------------------------------
# GPU: the strongly coupled rigid body under slabs (src/FluidSimulator.cpp:810-816, 1095-1100)

def _rigid_scene():
    from fluidsimulation_amd import scenes
    sc = scenes.add_sphere(scenes.cube_drop(32, ppc=4, seed=6), radius_world=0.12, centre_world=(0.5, 0.3, 0.5), density=500.0,
                           vel=(0.0, -0.2, 0.0, 0.0, 0.0, 0.0))
    rng = np.random.default_rng(4)
    sc.vel = rng.normal(0.0, 0.1, sc.pos.shape)
    return sc


def _rigid_worker(rank, port, out_dir, driver="python"):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist
    from fluidsimulation_amd import FlipContext
    from fluidsimulation_amd.distributed import Comm, DistributedSubstep, SlabLayout, partition_scene
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=WORLD)
    try:
        torch.cuda.set_device(0)
        sc = _rigid_scene()
        layout = SlabLayout(sc.dims[2], WORLD, 8)
        loc = partition_scene(sc, layout, rank)
        with FlipContext(loc.desc(), device=0) as ctx:
            ctx.reserve_particles(2 * loc.pos.shape[0])
            ctx.upload_scene(loc)
            if driver == "native":
                from fluidsimulation_amd.dist_native import NativeDistributedSubstep, TorchTransport
                comm = Comm()
                step = NativeDistributedSubstep(ctx, layout, comm, device=0, precond=1, max_iter=20000, overlap=4, constrain=True,
                                                rigid_body=True, transport=TorchTransport(comm, 0))
            else:
                step = DistributedSubstep(ctx, layout, Comm(), device=0, precond=1, max_iter=20000, overlap=4, constrain=True,
                                          rigid_body=True)
            info = step.step(0.004)
            pos, vel = ctx.download_particles()
            o0, o1 = layout.owned(rank)
            plane = np.floor(pos[:, 2] + 0.5).astype(np.int64) - sc.origin[2]
            own = (plane >= o0) & (plane < o1)
            np.savez(os.path.join(out_dir, f"rigid{rank}.npz"), pos=pos[own], vel=vel[own], v=ctx.solid_object_velocity(),
                     conv=info["divergence"]["converged"] * info["density"]["converged"], it=info["divergence"]["iterations"])
    except BaseException:
        import traceback
        with open(os.path.join(out_dir, f"error_{rank}.txt"), "w") as f:
            f.write(traceback.format_exc())
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("driver", ["python", "native"])
def test_rigid_body_coupling_under_two_slabs_matches_the_single_context(driver):
    import tempfile
    import torch.multiprocessing as mp
    from fluidsimulation_amd import FlipContext, FlipSubstep
    from fluidsimulation_amd.distributed import SlabLayout
    tmp = tempfile.mkdtemp()
    port = _free_port()
    try:
        mp.spawn(_rigid_worker, args=(port, tmp, driver), nprocs=WORLD, join=True)
    except Exception:
        for r in range(WORLD):
            f = os.path.join(tmp, f"error_{r}.txt")
            if os.path.exists(f):
                print(f"---- rank {r} ----\n" + open(f).read())
        raise
    sc = _rigid_scene()
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        info = FlipSubstep(ctx, precond=1, max_iter=20000, constrain=True, rigid_body=True).step(0.004)
        ref_v = ctx.solid_object_velocity()
        ref_pos, ref_vel = ctx.download_particles()
    assert info["divergence"]["converged"] == 1
    assert np.abs(ref_v - np.asarray(sc.object_vel)).max() > 1e-6       # the pressure really moved the body
    # Pressures themselves are not compared: a pocket of liquid cells walled in by the body has a pure Neumann block (and
    # a cell inside it an all-zero fluid row), so both systems are singular there and each solver returns its own member of
    # the solution set; what the pressures DO -- the body's velocity, the particles' velocities and displacements -- is
    # determined and must agree.
    from scipy.spatial import cKDTree
    vs, pos, vel = [], [], []
    for r in range(WORLD):
        d = np.load(os.path.join(tmp, f"rigid{r}.npz"))
        assert int(d["conv"]) == 1
        vs.append(d["v"]); pos.append(d["pos"]); vel.append(d["vel"])
    assert np.array_equal(vs[0], vs[1])                                  # every rank holds the same body
    assert np.abs(vs[0] - ref_v).max() <= 1e-8 * max(np.abs(ref_v).max(), 1e-30)
    pos, vel = np.concatenate(pos), np.concatenate(vel)
    assert pos.shape == ref_pos.shape
    dist_, idx = cKDTree(ref_pos).query(pos, k=1)
    assert dist_.max() < 1e-5 and np.array_equal(np.sort(idx), np.arange(ref_pos.shape[0]))
    assert np.abs(vel - ref_vel[idx]).max() / np.abs(ref_vel).max() < 1e-4


# ------------------------------------------------------------------------------------------------
# GPU: the C driver over RCCL itself, one rank (all a one-GPU box can hold): a communicator of one rank, the slab path forced

@pytest.mark.gpu
def test_native_driver_over_rccl_with_one_rank():
    """libflip_dist.so with its own RCCL communicator (ncclCommInitRank over one rank, ncclAllReduce after every CG kernel on
    the context's stream).  Forced onto the slab path it reproduces the Python-driven split-phase substep bit for bit (the
    same kernels in the same order), whether its two solves run side by side (two streams, two communicators) or one
    after the other; left alone it falls through to the single-context substep, also bit for bit."""
    from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_MULTIGRID
    from fluidsimulation_amd.distributed import Comm, DistributedSubstep, SlabLayout
    from fluidsimulation_amd.dist_native import NativeDistributedSubstep
    sc = _scene_gpu("dam")
    extras = _extras(sc, True)
    layout = SlabLayout(sc.dims[2], 1, 0)

    def run(make, paired=False):
        with FlipContext(sc.desc()) as ctx:
            ctx.upload_scene(sc)
            step = make(ctx)
            infos = [step.step(0.004) for _ in range(2)]
            pos, vel = ctx.download_particles()
            if paired:   # two system sets, the divergence one active: compare what the one-after-the-other runs leave, the density result
                from fluidsimulation_amd import SOLVE_DENSITY
                ctx.select_system(SOLVE_DENSITY)
            p = ctx.download("pressure")
            if hasattr(step, "close"):
                step.close()
        return infos, pos, vel, p

    ref_i, ref_pos, ref_vel, ref_p = run(lambda c: FlipSubstep(c, precond=PRECOND_MULTIGRID, max_iter=20000, concurrent_solves=True, **extras))
    py_i, py_pos, py_vel, py_p = run(lambda c: DistributedSubstep(c, layout, Comm(), device=0, precond=PRECOND_MULTIGRID, max_iter=20000, **extras))
    nat_i, nat_pos, nat_vel, nat_p = run(lambda c: NativeDistributedSubstep(c, layout, Comm(), device=0, precond=PRECOND_MULTIGRID,
                                                                            max_iter=20000, force_slab=True, **extras), paired=True)
    seq_i, seq_pos, seq_vel, seq_p = run(lambda c: NativeDistributedSubstep(c, layout, Comm(), device=0, precond=PRECOND_MULTIGRID,
                                                                            max_iter=20000, force_slab=True, sequential_solves=True, **extras))
    thru_i, thru_pos, thru_vel, thru_p = run(lambda c: NativeDistributedSubstep(c, layout, Comm(), device=0, precond=PRECOND_MULTIGRID,
                                                                                max_iter=20000, **extras))
    # Eigen's default iteration limit (twice the row count) makes the solver fetch the count flip_index_cells left on the device
    lim_i, lim_pos, lim_vel, lim_p = run(lambda c: NativeDistributedSubstep(c, layout, Comm(), device=0, precond=PRECOND_MULTIGRID,
                                                                            max_iter=0, **extras))
    assert np.array_equal(lim_pos, thru_pos) and np.array_equal(lim_vel, thru_vel) and np.array_equal(lim_p, thru_p)
    for i in nat_i + thru_i + lim_i:
        assert i["divergence"]["converged"] == 1 and i["density"]["converged"] == 1
    # forced slab path over RCCL == the Python-driven split-phase loop
    assert np.array_equal(nat_pos, py_pos) and np.array_equal(nat_vel, py_vel) and np.array_equal(nat_p, py_p)
    assert [i["divergence"]["iterations"] for i in nat_i] == [i["divergence"]["iterations"] for i in py_i]
    # the two slab solves side by side on two streams and two communicators (the default) == one after the other
    assert np.array_equal(nat_pos, seq_pos) and np.array_equal(nat_vel, seq_vel) and np.array_equal(nat_p, seq_p)
    assert [(i["divergence"]["iterations"], i["density"]["iterations"]) for i in nat_i] == \
        [(i["divergence"]["iterations"], i["density"]["iterations"]) for i in seq_i]
    # fall-through == flip_solve_pair substep
    assert np.array_equal(thru_pos, ref_pos) and np.array_equal(thru_vel, ref_vel) and np.array_equal(thru_p, ref_p)
    # and the two paths agree to the solver tolerance
    assert np.abs(nat_pos - ref_pos).max() < 1e-5
    assert np.abs(nat_vel - ref_vel).max() / np.abs(ref_vel).max() < 1e-4
