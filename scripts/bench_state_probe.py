"""What the scene looks like after the bench's timed region and its hot-path-only leg (measurement aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_MULTIGRID, scenes
sc = scenes.dam_break_slab(256, ppc=8, seed=12345)
radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
def report(ctx, tag):
    n_out = ctx.bin_particles()
    _, cs, _ = ctx.download_binning()
    cnt = np.diff(cs[: sc.n_cells + 1])
    pos, vel = ctx.download_particles()
    print(f"{tag}: n {ctx.particle_count()} outside {n_out} max/cell {cnt.max()} occupied {int((cnt > 0).sum())} |v|max {np.abs(vel).max():.3f} finite {np.isfinite(pos).all()}", flush=True)
def tbin(ctx, step):
    step.step(0.001); ctx.synchronize(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ctx.bin_particles(want_outside=False); e1.record(); e1.synchronize()
    return e0.elapsed_time(e1)
with FlipContext(sc.desc()) as ctx:
    ctx.upload_scene(sc)
    step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=20000, concurrent_solves=True, constrain=True, advect=True, levelset_radius=radius)
    for _ in range(25):
        step.step(0.001)
    report(ctx, "after 25 full steps")
    print("bin after a full step", tbin(ctx, step))
    hstep = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=20000, concurrent_solves=True, fused_particle_update=os.environ.get("PROBE_FUSED", "1") == "1")
    for q in range(20):
        info = hstep.step(0.001)
        if q % 4 == 3:
            _, vel = ctx.download_particles()
            print(f"  hot step {q}: |v|max {np.abs(vel).max():.3e} div it {info['divergence']['iterations']} den it {info['density']['iterations']}", flush=True)
    report(ctx, "after 20 hot-path steps")
    for _ in range(3):
        print("bin after a full step", tbin(ctx, step))
    report(ctx, "after 3 more full steps")
