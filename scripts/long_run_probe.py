"""The bench scene stepped K times; prints the iteration counts, residuals and time of every tenth substep (measurement
aid: how the solves behave as the column collapses)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_MULTIGRID, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.001
sc = scenes.dam_break_slab(n, ppc=8, seed=12345)
radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
with FlipContext(sc.desc()) as ctx:
    ctx.upload_scene(sc)
    step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=2000, concurrent_solves=True, constrain=True, advect=True, levelset_radius=radius)
    t0 = time.perf_counter()
    for k in range(K):
        info = step.step(dt)
        d, e = info["divergence"], info["density"]
        bad = not (d["converged"] == 1 and e["converged"] == 1)
        if (k + 1) % 10 == 0 or bad:
            ctx.synchronize()
            vmax = ctx.velocity_max()
            print(f"substep {k + 1:4d}  {1e3 * (time.perf_counter() - t0) / (k + 1):7.2f} ms/substep  div {d['iterations']:4d} it res {d['rel_residual_2']:.1e} conv {d['converged']}  "
                  f"den {e['iterations']:4d} it res {e['rel_residual_2']:.1e} conv {e['converged']}  rows {d['cg_rows']}  max|u| {np.abs(vmax).max():.2f}  cfl dt {float(sc.dx) / max(np.abs(vmax).max(), 1e-9):.4f}", flush=True)
        if bad:
            print("NOT CONVERGED", flush=True)
            break
