"""Wall-clock time of the phases of a substep on the bench scene, the device drained between them (measurement aid):
shows what the solves' own timers do not cover -- assembly and set-up inside flip_solve_pair -- and the phases' sizes."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_MULTIGRID, SOLVE_DENSITY, SOLVE_DIVERGENCE, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = 0.001
sc = scenes.dam_break_slab(n, ppc=8, seed=12345)
radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
with FlipContext(sc.desc()) as c:
    c.upload_scene(sc)
    step = FlipSubstep(c, precond=PRECOND_MULTIGRID, max_iter=20000, concurrent_solves=True, constrain=True, advect=True, levelset_radius=radius)
    for _ in range(8):
        step.step(dt)
    acc = {}
    def phase(name, fn):
        c.synchronize()
        t0 = time.perf_counter()
        r = fn()
        c.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    K = 12
    solver = 0.0
    for _ in range(K):
        phase("bin", lambda: c.bin_particles(want_outside=False))
        phase("levelset", lambda: c.fluid_levelset_from_particles(radius))
        def pre():
            c.side_fork(); c.side_select(True); c.index_cells(want_count=False); c.side_select(False)
            c.p2g(); c.mark_valid_from_active(); c.extrapolate(2); c.constrain_velocity(False)
            c.side_select(True); c.face_fractions(); c.side_select(False)
            c.update_prev(); c.add_forces(dt); c.density(want_count=False); c.side_join()
        phase("p2g..forces | labels, fractions", pre)
        d, e = phase("solve_pair (whole call)", lambda: c.solve_pair(dt, 1e-15, 20000, PRECOND_MULTIGRID))
        solver += max(d["seconds"], e["seconds"])
        def post():
            c.side_fork(); c.side_select(True); c.select_system(SOLVE_DENSITY); c.project_displacement(dt); c.select_system(SOLVE_DIVERGENCE)
            c.side_select(False); c.project_velocity(dt); c.extrapolate(2); c.constrain_velocity(False); c.update_diff(); c.side_join()
        phase("projections..diff", post)
        phase("update_particles", lambda: c.update_particles(0.95, dt, True, True))
    tot = 0.0
    for k, v in acc.items():
        print(f"{v / K * 1e3:8.3f} ms  {k}")
        tot += v / K
    print(f"{tot * 1e3:8.3f} ms  sum;  the longer solve's own timer (first iteration to tail): {solver / K * 1e3:.3f} ms")
