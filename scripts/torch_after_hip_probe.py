"""Does torch.cuda initialise when torch is IMPORTED only after libflip_hip.so has used the device? (measurement aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
mode = sys.argv[1] if len(sys.argv) > 1 else "late"
if mode == "early":
    import torch
from fluidsimulation_amd import FlipContext, scenes
sc = scenes.dam_break(24, ppc=4, seed=1)
with FlipContext(sc.desc()) as ctx:
    ctx.upload_scene(sc)
    ctx.bin_particles()
    print("context ok", flush=True)
    import torch
    print("torch imported; hip libs:", [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l][::8], flush=True)
    print("device_count", torch.cuda.device_count(), flush=True)
    print("torch after:", torch.zeros(1, device="cuda"), flush=True)
