"""The hot-path part of one FLIP substep as a sequence of C-ABI calls.

Mirrors the order of FluidSimulator::advance_flip_pic (src/FluidSimulator.cpp:307-446) for the stages
that are on the device.  Fluid sources and the level set from particles are not part of it: the level sets
are inputs.  `constrain=True` adds the two constrain_velocity calls (:357-361, :404-408) and `advect=True` the
RK3 particle advection (:426), `levelset_radius` rebuilds the fluid level set from the particles (:324); all default to off, which is the configuration the benchmark and the parity tests'
reference substep use (the particles then move only by the density-projection displacements).
"""
from __future__ import annotations

from .binding import FlipContext, PRECOND_JACOBI, SOLVE_DENSITY, SOLVE_DIVERGENCE


class FlipSubstep:
    def __init__(self, ctx: FlipContext, flip_pic_ratio: float = 0.95, tol: float = 1e-15, max_iter: int = 0,
                 precond: int = PRECOND_JACOBI, density_projection: bool = True, constrain: bool = False,
                 advect: bool = False, rigid_body: bool = False, levelset_radius: float | None = None,
                 concurrent_solves: bool = False, redistribute_seed: int | None = None, fused_particle_update: bool = True):
        self.ctx = ctx
        # the three particle updates that close the substep as one kernel (same bits) or as the three calls
        self.fused_particle_update = fused_particle_update
        # the density system does not depend on the divergence solve: both can be assembled first and solved side by side
        self.concurrent_solves = concurrent_solves and density_projection
        self.ratio = flip_pic_ratio      # main.cpp:167
        self.tol = tol                   # src/FluidSimulator.cpp:660
        self.max_iter = max_iter
        self.precond = precond
        self.density_projection = density_projection   # DENSITY_CONSTRAINT_PPE, src/FluidSimulator.cpp:31
        self.constrain, self.advect, self.rigid_body = constrain, advect, rigid_body
        self.levelset_radius = levelset_radius   # world units; None: the fluid level set is an input
        # re-seeding of over-dense cells (src/Functors.cpp:268-322) with draws keyed on (seed + substep number); None: off,
        # as in every fixture (the reference's own draws are not reproducible)
        self.redistribute_seed = redistribute_seed
        self.substeps = 0
        self.info = {}

    def _constrain(self):
        if self.constrain:
            self.ctx.constrain_velocity(False)    # :357 / :404
            if self.rigid_body:
                self.ctx.constrain_velocity(True) # :358-361 / :405-408

    def step(self, dt: float) -> dict:
        try:
            return self._step(dt)
        finally:
            self.ctx.side_join()   # a failed call must not leave the context on its side stream

    def _step(self, dt: float) -> dict:
        c = self.ctx
        c.bin_particles(want_outside=False)
        if self.levelset_radius is not None:
            c.fluid_levelset_from_particles(self.levelset_radius)   # :324
        # the labels and the face fractions do not depend on the velocity chain: they go to the context's second stream
        c.side_fork()
        c.side_select(True)
        c.index_cells(want_count=False)           # :486  (weights of the previous substep)
        c.side_select(False)
        c.p2g()                                   # :337
        c.mark_valid_from_active()                # :343-345
        c.extrapolate(2)                          # :349
        self._constrain()
        c.side_select(True)
        c.face_fractions()                        # :488  (writes the other buffer: the calls above keep the old values)
        c.side_select(False)
        c.update_prev()                           # :367
        c.add_forces(dt)                          # :373
        c.density(want_count=False)               # :491
        c.side_join()
        if self.redistribute_seed is not None:
            c.redistribute_overdense(self.redistribute_seed + self.substeps)   # src/Functors.cpp:268-322, inside :491
        self.substeps += 1
        info_den = None
        if self.concurrent_solves:
            info_div, info_den = c.solve_pair(dt, self.tol, self.max_iter, self.precond)   # :505 and :521
            c.side_fork()
            c.side_select(True)
            c.select_system(SOLVE_DENSITY)
            c.project_displacement(dt)            # :523
            c.select_system(SOLVE_DIVERGENCE)
            c.side_select(False)
            c.project_velocity(dt)                # :507
        else:
            c.build_system(SOLVE_DIVERGENCE, dt)  # :505
            info_div = c.pcg_solve(self.tol, self.max_iter, self.precond)
            c.project_velocity(dt)                # :507
        if self.density_projection and not self.concurrent_solves:
            c.build_system(SOLVE_DENSITY, dt)     # :521
            info_den = c.pcg_solve(self.tol, self.max_iter, self.precond)
            c.project_displacement(dt)            # :523
        c.extrapolate(2)                          # :398
        self._constrain()
        c.update_diff()                           # :416
        c.side_join()
        if self.fused_particle_update:
            c.update_particles(self.ratio, dt, self.advect, self.density_projection)   # :420, :426, :429 in one pass
        else:
            c.g2p(self.ratio)                         # :420
            if self.advect:
                c.advect_particles(dt)                # :426
            if self.density_projection:
                c.apply_displacements()               # :429
        self.info = {"divergence": info_div, "density": info_den}
        return self.info
