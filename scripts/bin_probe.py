"""Device and host time of flip_bin_particles right after a whole substep (measurement aid)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_MULTIGRID, scenes
sc = scenes.dam_break_slab(256, ppc=8, seed=12345)
radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
with FlipContext(sc.desc()) as ctx:
    ctx.upload_scene(sc)
    step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=20000, concurrent_solves=True, constrain=True, advect=True, levelset_radius=radius)
    for _ in range(5):
        step.step(0.001)
    for rep in range(4):
        step.step(0.001)
        ctx.synchronize(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        ctx.bin_particles(want_outside=False)
        t1 = time.perf_counter()
        e1.record(); e1.synchronize()
        t2 = time.perf_counter()
        print(f"bin: events {e0.elapsed_time(e1):.3f} ms, host enqueue {1e3 * (t1 - t0):.3f} ms, host until done {1e3 * (t2 - t0):.3f} ms", flush=True)
