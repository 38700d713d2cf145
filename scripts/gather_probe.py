"""Times flip_p2g alone on the bench scene after a few substeps (tuning aid for the gather kernels).
    python scripts/gather_probe.py [n] [steps]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_MULTIGRID, scenes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sc = scenes.dam_break_slab(n, ppc=8, seed=12345)
ctx = FlipContext(sc.desc(), device=0)
ctx.reserve_particles(sc.pos.shape[0] * 5 // 4 + 4096)
ctx.upload_scene(sc)
radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
step = FlipSubstep(ctx, tol=1e-15, max_iter=20000, precond=PRECOND_MULTIGRID, constrain=True, advect=True, levelset_radius=radius)
for _ in range(steps):
    step.step(float(os.environ.get('PROBE_DT', '0.001')))
ctx.bin_particles(want_outside=False)
s0 = ctx.gather_stats()
def timeit(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); ctx.synchronize()
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
ctx.gather_stats()
t = timeit(ctx.p2g)
s1 = ctx.gather_stats()
print(f"n={n} steps={steps} particles={ctx.particle_count()} p2g {t*1e3:.1f} us; tiles per launch staged {(s1['staged_tiles']-s0['staged_tiles'])/6:.0f} "
      f"unstaged {(s1['unstaged_tiles']-s0['unstaged_tiles'])/6:.0f}")
def rebin():
    ctx.apply_displacements(); ctx.bin_particles(want_outside=False)
print(f"disp+bin {timeit(rebin)*1e3:.1f} us, disp {timeit(ctx.apply_displacements)*1e3:.1f} us")
ctx.bin_particles(want_outside=False)
print(f"levelset {timeit(lambda: ctx.fluid_levelset_from_particles(radius))*1e3:.1f} us")
ctx.update_diff()
print(f"g2p {timeit(lambda: ctx.g2p(0.95))*1e3:.1f} us")
# binning right after a real substep
ctx2 = ctx
for _ in range(3):
    step.step(0.001)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ctx.synchronize(); torch.cuda.synchronize()
    e0.record(); ctx.bin_particles(want_outside=False); e1.record(); e1.synchronize()
    print(f"bin after a real substep {e0.elapsed_time(e1)*1e3:.1f} us")
ctx.close()
