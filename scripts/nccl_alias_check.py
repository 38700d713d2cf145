"""One-rank RCCL check of what the slab solver asks of torch.distributed: collectives on tensors that alias the
library's own device buffers (not torch allocations), on the stream the library launches on."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fluidsimulation_amd import FlipContext, scenes  # noqa: E402
from fluidsimulation_amd.distributed import device_tensor  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sc = scenes.dam_break(32, ppc=4, seed=1)
with FlipContext(sc.desc(), device=0) as ctx:
    ctx.upload_scene(sc)
    ctx.set_slab(0, sc.dims[2])
    nx, ny, nz = ctx.dims
    p, n = ctx.dist_buffer(0)
    sums = device_tensor(p, n, 0)
    sums.copy_(torch.arange(n, dtype=torch.float64, device="cuda"))
    dist.all_reduce(sums)
    assert sums.cpu().tolist() == list(range(n)), sums
    p, n = ctx.dist_buffer(1)
    z = device_tensor(p, n, 0).view(nz, nx * ny)
    z[3] = 7.0
    ops = [dist.P2POp(dist.isend, z[3], 0), dist.P2POp(dist.irecv, z[5], 0)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    torch.cuda.synchronize()
    assert float(z[5].min()) == 7.0 and float(z[5].max()) == 7.0
    # the library reads what RCCL wrote without an explicit synchronisation in between (same null stream)
    print("nccl on aliased library buffers: ok")
dist.destroy_process_group()
