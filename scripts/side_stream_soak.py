"""The bench scene stepped K times with and without the second stream inside a substep; the particle states must be equal
bit for bit after every tenth substep (a race between the two streams would show as a difference that comes and goes),
every solve must converge and the iteration counts must agree (measurement aid / soak test at full size)."""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_MULTIGRID, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
K = int(sys.argv[2]) if len(sys.argv) > 2 else 60
sc = scenes.dam_break_slab(n, ppc=8, seed=12345)
radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0
runs = []
for side in (True, False):
    log = []
    with FlipContext(sc.desc()) as ctx:
        ctx.upload_scene(sc)
        ctx.set_side_stream(side)
        step = FlipSubstep(ctx, precond=PRECOND_MULTIGRID, max_iter=20000, concurrent_solves=True, constrain=True, advect=True,
                           levelset_radius=radius)
        for k in range(K):
            info = step.step(0.001)
            assert info["divergence"]["converged"] == 1 and info["density"]["converged"] == 1, (k, info)
            entry = (info["divergence"]["iterations"], info["density"]["iterations"])
            if (k + 1) % 10 == 0:
                pos, vel = ctx.download_particles()
                h = hashlib.sha1(pos.tobytes() + vel.tobytes() + ctx.download("label").tobytes() + ctx.download("weight_u").tobytes())
                entry += (h.hexdigest()[:16],)
            log.append(entry)
    runs.append(log)
    print("side stream" if side else "one stream ", [e[2] for e in log if len(e) > 2], flush=True)
same = runs[0] == runs[1]
print("identical" if same else "DIFFERENT", "after", K, "substeps at", n)
sys.exit(0 if same else 1)
