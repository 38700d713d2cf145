"""Achievable HBM bandwidth on this GPU for plain streaming kernels of several sizes (calibration for the rooflines)."""
import torch
torch.cuda.set_device(0)
for mb in (16, 64, 256, 1024, 4096):
    n = mb * 1024 * 1024 // 8
    a = torch.zeros(n, dtype=torch.float64, device="cuda")
    b = torch.ones(n, dtype=torch.float64, device="cuda")
    c = torch.ones(n, dtype=torch.float64, device="cuda")
    for name, fn, nbytes in (("copy", lambda: a.copy_(b), 16 * n), ("add", lambda: torch.add(b, c, out=a), 24 * n),
                             ("sum", lambda: b.sum(), 8 * n)):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) * 1e-3 / reps
        print(f"{mb:5d} MB arrays  {name:5s} {t * 1e6:9.1f} us  {nbytes / t / 1e12:6.2f} TB/s")
