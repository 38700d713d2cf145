"""CG iterations per substep of ONE context on the multi-GPU benchmark's long domain (z_repeat slabs in a row): the
count the slab solver's iterations should be compared with."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_MULTIGRID, scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for rep in (1, 2, 4, 8):
    sc = scenes.dam_break_slab(n, ppc=8, seed=12345, z_repeat=rep)
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        st = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=20000, concurrent_solves=True)
        st.step(0.005)
        its = []
        for _ in range(2):
            i = st.step(0.005)
            its.append(i["divergence"]["iterations"] + i["density"]["iterations"])
        print(f"n {n} z_repeat {rep}: {np.mean(its):.1f} iterations per substep, rows {i['divergence']['cg_rows']}")
