"""Wall time of every stage of the substep at the bench configuration (device-synchronised around each call)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_MULTIGRID, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = scenes.dam_break_slab(n, ppc=8, seed=12345)
with FlipContext(sc.desc()) as c:
    c.reserve_particles(sc.pos.shape[0] * 5 // 4)
    c.upload_scene(sc)
    st = FlipSubstep(c, precond=PRECOND_MULTIGRID, max_iter=20000, concurrent_solves=True)
    for _ in range(2):
        st.step(0.005)
    acc = {}

    def t(name, fn):
        c.synchronize()
        t0 = time.perf_counter()
        r = fn()
        c.synchronize()
        acc[name] = acc.get(name, 0.0) + time.perf_counter() - t0
        return r
    reps = 4
    dt = 0.005
    for _ in range(reps):
        t("bin_particles", lambda: c.bin_particles(want_outside=False))
        t("p2g(+density)", c.p2g)
        t("mark_valid", c.mark_valid_from_active)
        t("extrapolate#1", lambda: c.extrapolate(2))
        t("update_prev", c.update_prev)
        t("add_forces", lambda: c.add_forces(dt))
        t("index_cells", c.index_cells)
        t("face_fractions", c.face_fractions)
        t("density(no-op)", lambda: c.density(want_count=False))
        t("solve_pair", lambda: c.solve_pair(dt, 1e-15, 20000, PRECOND_MULTIGRID))
        t("project_velocity", lambda: c.project_velocity(dt))
        c.select_system(1)
        t("project_displacement", lambda: c.project_displacement(dt))
        c.select_system(0)
        t("extrapolate#2", lambda: c.extrapolate(2))
        t("update_diff", c.update_diff)
        t("g2p", lambda: c.g2p(0.95))
        t("apply_displacements", c.apply_displacements)
    tot = sum(acc.values())
    for k, v in acc.items():
        print(f"{k:22s} {v / reps * 1e3:8.3f} ms  {100 * v / tot:5.1f} %")
    print(f"{'sum':22s} {tot / reps * 1e3:8.3f} ms")
