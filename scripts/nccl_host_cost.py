"""Host and device cost of the collectives the slab solver issues per CG iteration, one rank talking to itself."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
sums = torch.zeros(8, dtype=torch.float64, device="cuda")
plane = 256 * 256
a = torch.zeros(5 * plane, dtype=torch.float64, device="cuda")
b = torch.zeros(5 * plane, dtype=torch.float64, device="cuda")
c = torch.zeros(5 * plane, dtype=torch.float64, device="cuda")
d = torch.zeros(5 * plane, dtype=torch.float64, device="cuda")


def exchange():
    ops = [dist.P2POp(dist.isend, a, 0), dist.P2POp(dist.irecv, b, 0), dist.P2POp(dist.isend, c, 0), dist.P2POp(dist.irecv, d, 0)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()


for name, fn in (("all_reduce(8 doubles)", lambda: dist.all_reduce(sums)), ("exchange(2 x 5 planes of 256^2)", exchange)):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    n = 500
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{name}: host {t_host / n * 1e6:.1f} us per call, with device {t_all / n * 1e6:.1f} us per call")
dist.destroy_process_group()
