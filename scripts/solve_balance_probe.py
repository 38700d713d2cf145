"""Per-solve iteration counts and device times of the two solves of a substep on the bench scene (measurement aid)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_MULTIGRID, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = scenes.dam_break_slab(n, ppc=8, seed=12345)
radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
with FlipContext(sc.desc()) as ctx:
    ctx.upload_scene(sc)
    for conc in (True, False):
        step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=20000, concurrent_solves=conc, constrain=True, advect=True, levelset_radius=radius)
        for _ in range(5):
            step.step(0.001)
        ctx.synchronize()
        t0 = time.perf_counter()
        infos = [step.step(0.001) for _ in range(10)]
        ctx.synchronize()
        el = (time.perf_counter() - t0) / 10
        print("concurrent" if conc else "sequential", f"{el * 1e3:.2f} ms/step")
        for i in infos:
            d, e = i["divergence"], i["density"]
            print(f"  div {d['iterations']:3d} it {d['seconds'] * 1e3:6.2f} ms ({d['iterations_inf']:3d} to the inf rule) | den {e['iterations']:3d} it {e['seconds'] * 1e3:6.2f} ms ({e['iterations_inf']:3d})")
