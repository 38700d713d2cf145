"""Device time of the P2G gather (and the stand-alone density gather) at the bench configuration."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fluidsimulation_amd import FlipContext, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sc = scenes.dam_break_slab(n, ppc=8, seed=12345)
with FlipContext(sc.desc()) as c:
    c.upload_scene(sc)
    c.bin_particles(want_outside=False)
    def t(fn, reps=5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn(); tot = 0.0
        for _ in range(reps):
            e0.record(); fn(); e1.record(); e1.synchronize(); tot += e0.elapsed_time(e1)
        return tot / reps
    print(f"column {os.environ.get('FLIP_GATHER_COLUMN', 'default')} planes {os.environ.get('FLIP_GATHER_PLANES', 'default')}: p2g {t(c.p2g) * 1e3:.0f} us", flush=True)
    def dens():
        c.upload("phi_fluid", sc.phi_fluid); c.density(want_count=False)
    def up():
        c.upload("phi_fluid", sc.phi_fluid)
    print(f"   density alone {(t(dens) - t(up)) * 1e3:.0f} us")
