"""z-slab decomposition of the FLIP substep over several GPUs: one process and one FlipContext per GPU.

The reference is a single-process CPU program, so nothing here has a counterpart in it.  The dense box is cut
into slabs along z, the slowest-varying index, so a plane of a field is one contiguous block of nx*ny elements.
Rank r owns the cells, rows and particles of planes [start[r], start[r+1]) and additionally computes on `ghost`
planes of its neighbours' slabs on either side:

  * every stage up to the assembly of the linear system, and every stage after the solve, reaches at most a
    few cells sideways, so with the neighbours' particles and level-set values present in the ghost planes a
    rank computes its own slab exactly with no communication at all (redundant work in the ghost planes);
  * the solve is global: conjugate gradients over all ranks' rows with one all-reduce of <= 8 doubles after
    each of its two kernels and one exchange of a single plane per slab face and iteration; the
    preconditioner (the V-cycle, or point Jacobi) acts within each slab only, which keeps it symmetric
    positive definite without communication (block Jacobi over the ranks);
  * after the solve the ghost planes of the pressure are filled from the neighbours, after the particle update
    the particles near a slab face are exchanged (migration and ghost copies in one message per face).

The communication layer is torch.distributed: backend "nccl" (RCCL over xGMI) on GPU tensors that alias the
library's device buffers, or "gloo" with host staging (CPU tests, or several ranks sharing one GPU).
"""
from __future__ import annotations

import numpy as np

from .binding import PRECOND_JACOBI, PRECOND_MULTIGRID, SOLVE_DENSITY, SOLVE_DIVERGENCE


class SlabLayout:
    """Even split of nz planes over `world` ranks with `ghost` extra planes on interior faces."""

    def __init__(self, nz: int, world: int, ghost: int, starts=None):
        if starts is None:
            starts = [(nz * r) // world for r in range(world + 1)]
        self.nz, self.world, self.ghost = int(nz), int(world), int(ghost)
        self.starts = [int(s) for s in starts]
        assert self.starts[0] == 0 and self.starts[-1] == nz and len(self.starts) == world + 1
        for r in range(world):
            assert self.starts[r + 1] - self.starts[r] >= max(ghost, 1) or world == 1, \
                "a slab must be at least as thick as the ghost zone, so that ghosts reach only the next rank"

    @classmethod
    def balanced(cls, weights, world: int, ghost: int):
        """Split so that every rank owns about the same total weight (e.g. particles per plane): a dam break does not
        fill the planes inside the container walls, and an even split would leave the end ranks nearly idle.  Every
        slab stays at least as thick as the ghost zone."""
        w = np.asarray(weights, dtype=np.float64)
        nz = int(w.size)
        if world == 1:
            return cls(nz, 1, ghost)
        cum = np.concatenate([[0.0], np.cumsum(w)])
        total = cum[-1] if cum[-1] > 0 else 1.0
        min_thick = max(int(ghost), 1)
        assert nz >= world * min_thick, "more ranks than slabs of ghost thickness fit into the box"
        starts = [0]
        for r in range(1, world):
            k = int(np.searchsorted(cum, total * r / world))
            k = max(k, starts[-1] + min_thick)              # thick enough
            k = min(k, nz - (world - r) * min_thick)         # and room for the ranks that follow
            starts.append(k)
        starts.append(nz)
        return cls(nz, world, ghost, starts=starts)

    def owned(self, rank: int):
        return self.starts[rank], self.starts[rank + 1]

    def extended(self, rank: int):
        lo, hi = self.owned(rank)
        return max(0, lo - self.ghost), min(self.nz, hi + self.ghost)

    def local_owned(self, rank: int):
        """Owned planes in the coordinates of the rank's extended box."""
        lo, hi = self.owned(rank)
        e0, _ = self.extended(rank)
        return lo - e0, hi - e0


class _DeviceArray:
    """Exposes raw device memory through __cuda_array_interface__ so that torch can alias it."""

    def __init__(self, ptr: int, n: int):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def device_tensor(ptr: int, n: int, device: int):
    import torch
    return torch.as_tensor(_DeviceArray(ptr, n), device=torch.device("cuda", device))


class Comm:
    """Sum all-reduce and neighbour exchange over torch.distributed; stages through the host when the backend
    cannot take the tensors as they are (gloo with device tensors)."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else "none"

    def _native(self, t) -> bool:
        return self.backend == "nccl" or not t.is_cuda

    def all_reduce_sum(self, t):
        if self.world == 1:
            return
        if self._native(t):
            self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)

    def all_reduce_max(self, t):
        if self.world == 1:
            return
        if self._native(t):
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=self.dist.ReduceOp.MAX, group=self.group)
            t.copy_(h)

    def exchange(self, send_lo, recv_lo, send_hi, recv_hi):
        """send_lo -> rank-1 (arrives in its recv_hi), send_hi -> rank+1 (arrives in its recv_lo).  Tensors
        toward a side without a neighbour are ignored."""
        if self.world == 1:
            return
        dist = self.dist
        ops, staged = [], []

        def add(kind, t, peer):
            if self._native(t) and t.is_contiguous():
                wire = t
            elif kind == "send":
                wire = t.detach().cpu().contiguous()
            else:
                wire = t.new_empty(t.shape, device="cpu")
                staged.append((t, wire))
            ops.append(dist.P2POp(dist.isend if kind == "send" else dist.irecv, wire, peer, self.group))

        if self.rank > 0:
            add("send", send_lo, self.rank - 1)
            add("recv", recv_lo, self.rank - 1)
        if self.rank < self.world - 1:
            add("send", send_hi, self.rank + 1)
            add("recv", recv_hi, self.rank + 1)
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        for t, wire in staged:
            t.copy_(wire)


class SlabExchange:
    """Plane exchanges shared by the device solver and its numpy stand-in: `self.z` and `self.x` are tensors
    of shape (nz_local, nx*ny) over the rank's extended box."""

    def _init_slab(self, layout: SlabLayout, rank: int):
        assert layout.world == 1 or layout.ghost >= 1
        self.layout, self.rank = layout, rank
        self.k0, self.k1 = layout.local_owned(rank)
        e0, e1 = layout.extended(rank)
        o0, o1 = layout.owned(rank)
        # ghost thickness on either side of this rank's box, and of the neighbours' boxes toward this rank
        self.g_lo, self.g_hi = o0 - e0, e1 - o1
        self.g_of_lower = (layout.extended(rank - 1)[1] - layout.owned(rank - 1)[1]) if rank > 0 else 0
        self.g_of_upper = (layout.owned(rank + 1)[0] - layout.extended(rank + 1)[0]) if rank < layout.world - 1 else 0

    overlap = 0   # planes of either neighbour's slab the local preconditioner also covers (0: disjoint blocks)

    def exchange_z(self, comm: Comm):
        k0, k1 = self.k0, self.k1
        comm.exchange(self.z[k0], self.z[k0 - 1] if self.g_lo else None, self.z[k1 - 1], self.z[k1] if self.g_hi else None)

    def exchange_x(self, comm: Comm):
        """Fill every ghost plane of the solution from the neighbour that owns it."""
        k0, k1 = self.k0, self.k1
        comm.exchange(self.x[k0:k0 + self.g_of_lower], self.x[k0 - self.g_lo:k0] if self.g_lo else None,
                      self.x[k1 - self.g_of_upper:k1], self.x[k1:k1 + self.g_hi] if self.g_hi else None)

    def exchange_r(self, comm: Comm):
        """Overlapping variant: the residual on the overlap planes, from the neighbour that owns them."""
        k0, k1, m = self.k0, self.k1, self.overlap
        comm.exchange(self.r[k0:k0 + m], self.r[k0 - m:k0] if self.g_lo else None,
                      self.r[k1 - m:k1], self.r[k1:k1 + m] if self.g_hi else None)

    def exchange_corrections(self, comm: Comm):
        """Overlapping variant: what this rank's preconditioner computed on the neighbours' planes goes to them,
        together with its value on the owned plane next to the face (the neighbours' first ghost plane)."""
        k0, k1, m = self.k0, self.k1, self.overlap
        comm.exchange(self.z[k0 - m:k0 + 1] if self.g_lo else None, self.recv_lo,
                      self.z[k1 - 1:k1 + m] if self.g_hi else None, self.recv_hi)


class ContextSlabSolver(SlabExchange):
    """The split-phase solver of one FlipContext (flip_pcg_dist_*) with its buffers aliased as torch tensors."""

    def __init__(self, ctx, layout: SlabLayout, rank: int, device: int, overlap: int = 0):
        self._init_slab(layout, rank)
        self.ctx = ctx
        ctx.set_slab(self.k0, self.k1)
        nx, ny, nz = ctx.dims
        self.overlap = int(overlap) if layout.world > 1 else 0
        assert self.overlap <= layout.ghost, "the overlap lives in the ghost planes"
        if self.overlap:
            ctx.set_slab_overlap(self.overlap)
            p, n = ctx.dist_buffer(3)
            self.r = device_tensor(p, n, device).view(nz, nx * ny)
            p, n = ctx.dist_buffer(4)
            self.recv_lo = device_tensor(p, n, device).view(self.overlap + 1, nx * ny)
            p, n = ctx.dist_buffer(5)
            self.recv_hi = device_tensor(p, n, device).view(self.overlap + 1, nx * ny)
        p, n = ctx.dist_buffer(0)
        self.sums = device_tensor(p, n, device)
        p, n = ctx.dist_buffer(1)
        self.z = device_tensor(p, n, device).view(nz, nx * ny)
        p, n = ctx.dist_buffer(2)
        self.x = device_tensor(p, n, device).view(nz, nx * ny)

    def begin(self, tol, max_iter, precond):
        self.ctx.dist_begin(tol, max_iter, precond)

    def direction(self):
        self.ctx.dist_direction()

    def update(self):
        self.ctx.dist_update()

    def precondition(self):
        self.ctx.dist_precondition()

    def combine(self):
        self.ctx.dist_combine()

    def status(self):
        return self.ctx.dist_status()

    def finish(self):
        return self.ctx.dist_finish()


def slab_pcg_solve(solver, comm: Comm, tol: float, max_iter: int, precond: int, check_every: int = 8) -> dict:
    """Conjugate gradients over all ranks' slabs.  `solver` is a ContextSlabSolver or anything with the same
    methods (the CPU tests drive a numpy stand-in through this very loop).  Every rank executes the same number
    of iterations: the stop flag is a function of all-reduced values only."""
    overlapping = getattr(solver, "overlap", 0) > 0

    def precondition_and_sums():
        # z = M^-1 r, then the global r.r and r.z
        if overlapping:
            solver.exchange_r(comm)
            solver.precondition()
            solver.exchange_corrections(comm)
            solver.combine()
        comm.all_reduce_sum(solver.sums)
        if not overlapping:
            solver.exchange_z(comm)

    solver.begin(tol, max_iter, precond)
    precondition_and_sums()
    done = 0
    while not done:
        for _ in range(check_every):
            solver.direction()
            comm.all_reduce_sum(solver.sums)
            solver.update()
            precondition_and_sums()
        done, _ = solver.status()
    solver.exchange_x(comm)
    return solver.finish()


def partition_scene(sc, layout: SlabLayout, rank: int):
    """The part of a global scene one rank holds: level sets over its extended box and the particles whose cell
    lies in it (ghost copies included).  Positions stay in global index space; only the box origin moves."""
    import copy
    e0, e1 = layout.extended(rank)
    loc = copy.copy(sc)
    nx, ny, _ = sc.dims
    loc.dims = (nx, ny, e1 - e0)
    loc.origin = (sc.origin[0], sc.origin[1], sc.origin[2] + e0)
    loc.phi_solid = np.ascontiguousarray(sc.phi_solid[e0:e1])
    loc.phi_fluid = np.ascontiguousarray(sc.phi_fluid[e0:e1])
    if sc.phi_object is not None:
        loc.phi_object = np.ascontiguousarray(sc.phi_object[e0:e1])
    cell_k = np.floor(sc.pos[:, 2] + 0.5).astype(np.int64) - sc.origin[2]
    keep = (cell_k >= e0) & (cell_k < e1)
    loc.pos = np.ascontiguousarray(sc.pos[keep])
    loc.vel = np.ascontiguousarray(sc.vel[keep])
    loc.mass = np.ascontiguousarray(sc.mass[keep])
    o0, o1 = layout.owned(rank)
    loc.owned = ((cell_k[keep] >= o0) & (cell_k[keep] < o1))
    loc.global_index = np.nonzero(keep)[0]
    return loc


class ParticleExchange:
    """Migration and ghost refresh after the particle update: one count message and one payload message per
    slab face, device to device (flip_particles_pack / flip_particles_exchange_apply)."""

    RECORD_DOUBLES = 8   # 64-byte records

    def __init__(self, ctx, geometry: SlabExchange, comm: Comm, device: int, capacity: int):
        import torch
        self.ctx, self.geo, self.comm = ctx, geometry, comm
        self.capacity = max(int(capacity), 1)
        dev = torch.device("cuda", device)
        mk = lambda: torch.zeros((self.capacity, self.RECORD_DOUBLES), dtype=torch.float64, device=dev)  # noqa: E731
        self.send_lo, self.send_hi, self.recv_lo, self.recv_hi = mk(), mk(), mk(), mk()
        self.counts = torch.zeros(4, dtype=torch.int64, device=dev)  # send_lo, send_hi, recv_lo, recv_hi

    def run(self) -> dict:
        g, c = self.geo, self.ctx
        big = 1 << 30
        has_lo, has_hi = self.comm.rank > 0, self.comm.rank < self.comm.world - 1
        # everything authoritative that now lies inside the neighbour's extended slab: migrants and its ghosts
        n_lo = c.pack_particles(-big, g.k0 + g.g_of_lower, self.send_lo.data_ptr(), self.capacity) if has_lo else 0
        n_hi = c.pack_particles(g.k1 - g.g_of_upper, big, self.send_hi.data_ptr(), self.capacity) if has_hi else 0
        self.counts[0], self.counts[1] = n_lo, n_hi
        self.comm.exchange(self.counts[0:1], self.counts[2:3], self.counts[1:2], self.counts[3:4])
        got = self.counts.tolist()
        m_lo, m_hi = (got[2] if has_lo else 0), (got[3] if has_hi else 0)
        assert m_lo <= self.capacity and m_hi <= self.capacity, "particle exchange buffer too small"
        # at least one record travels either way so that no message is empty
        self.comm.exchange(self.send_lo[:max(n_lo, 1)], self.recv_lo[:max(m_lo, 1)], self.send_hi[:max(n_hi, 1)],
                           self.recv_hi[:max(m_hi, 1)])
        c.exchange_apply(self.recv_lo.data_ptr(), m_lo, self.recv_hi.data_ptr(), m_hi)
        return {"sent": (n_lo, n_hi), "received": (m_lo, m_hi)}


class DistributedSubstep:
    """FlipSubstep over a slab decomposition: the same call order (src/FluidSimulator.cpp:307-446), with the two
    solves replaced by slab_pcg_solve and a particle exchange at the end of the substep.  `constrain`, `advect` and
    `levelset_radius` switch on the same optional stages as in FlipSubstep; each reaches only a few cells sideways, so
    the ghost planes cover them without further communication."""

    def __init__(self, ctx, layout: SlabLayout, comm: Comm, device: int = 0, flip_pic_ratio: float = 0.95,
                 tol: float = 1e-15, max_iter: int = 20000, precond: int = PRECOND_MULTIGRID,
                 density_projection: bool = True, overlap: int = 0, constrain: bool = False, advect: bool = False,
                 levelset_radius: float | None = None, rigid_body: bool = False):
        self.ctx, self.layout, self.comm = ctx, layout, comm
        self.rigid_body = rigid_body   # every rank holds the same body (flip_solid_object_set) and the same velocity of it
        self.solver = ContextSlabSolver(ctx, layout, comm.rank, device, overlap=overlap)
        self.ratio, self.tol, self.max_iter, self.precond = flip_pic_ratio, tol, max_iter, precond
        self.density_projection = density_projection
        self.constrain, self.advect, self.levelset_radius = constrain, advect, levelset_radius
        # a message holds at most the sender's authoritative particles; slabs hold similar numbers of them
        self.exchange = ParticleExchange(ctx, self.solver, comm, device, capacity=ctx.particle_count() * 5 // 4 + 4096) \
            if comm.world > 1 else None

    def solve(self, kind: int, dt: float) -> dict:
        self.ctx.build_system(kind, dt)
        if self.rigid_body and self.comm.world > 1:
            # The slab solver works on the eliminated system.  A rigid-body row that is also a dependent row (a free-surface
            # cell next to the body) rules that out, and whether one exists is a property of all slabs together: agree before
            # the first collective of the solve, so that every rank takes the same way out.
            import torch
            flag = torch.tensor([float(self.ctx.system_dependent_coupled() > 0)], dtype=torch.float64)
            if self.comm.backend == "nccl":
                flag = flag.cuda()
            self.comm.all_reduce_max(flag)
            if float(flag.cpu()[0]) > 0:
                raise RuntimeError("the rigid body touches dependent rows (cells with phi_fluid >= 0 that carry a pressure row) on some "
                                   "rank; the slab solver needs the eliminated system, which such rows rule out")
        return slab_pcg_solve(self.solver, self.comm, self.tol, self.max_iter, self.precond)

    def step(self, dt: float) -> dict:
        c = self.ctx
        c.bin_particles(want_outside=False)
        if self.levelset_radius is not None:
            c.fluid_levelset_from_particles(self.levelset_radius)   # :324
        c.p2g()
        c.mark_valid_from_active()
        c.extrapolate(2)
        if self.constrain:
            c.constrain_velocity(False)                             # :357
            if self.rigid_body:
                c.constrain_velocity(True)                          # :358-361
        c.update_prev()
        c.add_forces(dt)
        c.index_cells()
        c.face_fractions()
        c.density(want_count=False)
        out = {"divergence": self.solve(SOLVE_DIVERGENCE, dt), "density": None}
        c.project_velocity(dt)
        if self.rigid_body:
            # V += dt Minv (J p), J p summed over every slab's rows (src/FluidSimulator.cpp:1095-1100)
            import torch
            jp = torch.from_numpy(c.solid_coupling_partial())
            if self.comm.backend == "nccl":
                jp = jp.cuda()
            self.comm.all_reduce_sum(jp)
            c.solid_velocity_apply(dt, jp.cpu().numpy())
        if self.density_projection:
            out["density"] = self.solve(SOLVE_DENSITY, dt)
            c.project_displacement(dt)
        c.extrapolate(2)
        if self.constrain:
            c.constrain_velocity(False)                             # :404
            if self.rigid_body:
                c.constrain_velocity(True)                          # :405-408
        c.update_diff()
        c.update_particles(self.ratio, dt, self.advect, self.density_projection)   # :420, :426, :429 in one pass
        if self.exchange is not None:
            out["exchange"] = self.exchange.run()
        return out


__all__ = ["SlabLayout", "Comm", "SlabExchange", "ContextSlabSolver", "slab_pcg_solve", "partition_scene", "ParticleExchange", "DistributedSubstep",
           "device_tensor", "PRECOND_JACOBI", "PRECOND_MULTIGRID"]
