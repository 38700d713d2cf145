"""Prints a digest of the particle state after three substeps (run with FLIP_FUSED_UPDATE=0 and =1 and compare)."""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import FlipContext, FlipSubstep, scenes
sc = scenes.dam_break(48, ppc=6, seed=21)
sc.vel = np.random.default_rng(5).normal(0.0, 0.3, sc.pos.shape)
for paired in (False, True):
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        st = FlipSubstep(ctx, precond=1, max_iter=20000, concurrent_solves=paired)
        its = []
        for _ in range(3):
            i = st.step(0.004)
            its.append((i["divergence"]["iterations"], i["density"]["iterations"]))
        pos, vel = ctx.download_particles()
        print(paired, its, hashlib.sha1(pos.tobytes() + vel.tobytes()).hexdigest())
