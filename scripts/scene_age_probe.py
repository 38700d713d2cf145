"""Per-cell particle count statistics of the bench scene as it ages (diagnostic).
    python scripts/scene_age_probe.py [n] [steps...]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_MULTIGRID, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
dt = float(os.environ.get("PROBE_DT", "0.005"))
marks = [int(a) for a in sys.argv[2:]] or [5, 12, 25, 40]
sc = scenes.dam_break_slab(n, ppc=8, seed=12345)
ctx = FlipContext(sc.desc(), device=0)
ctx.reserve_particles(sc.pos.shape[0] * 5 // 4 + 4096)
ctx.upload_scene(sc)
radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
step = FlipSubstep(ctx, tol=1e-15, max_iter=20000, precond=PRECOND_MULTIGRID, constrain=True, advect=True, levelset_radius=radius)
done = 0
for m in marks:
    while done < m:
        info = step.step(dt)
        done += 1
    n_out = ctx.bin_particles()
    _, cell_start, _ = ctx.download_binning()
    cnt = np.diff(cell_start.astype(np.int64))
    occ = cnt[cnt > 0]
    vmax = ctx.velocity_max()
    pos, vel = ctx.download_particles()
    sp = np.abs(vel).max(axis=1)
    print(f"step {done}: outside {n_out}, occupied cells {occ.size}, mean {occ.mean():.2f}, p99 {np.percentile(occ, 99):.0f}, max {occ.max()}, "
          f">16: {(occ > 16).sum()}, >32: {(occ > 32).sum()}, >64: {(occ > 64).sum()}; vmax {np.round(vmax, 2)}; particle speed p50 {np.percentile(sp, 50):.2f} p99 {np.percentile(sp, 99):.2f} p99.99 {np.percentile(sp, 99.99):.2f} max {sp.max():.1f}; iters {info['divergence']['iterations']}/{info['density']['iterations']}", flush=True)
ctx.close()
