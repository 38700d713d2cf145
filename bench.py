"""Benchmark of the FLIP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" is one pass of the hot path (fluidsimulation_amd.substep.FlipSubstep.step: binning, P2G,
extrapolation, forces, labels, face fractions, density, divergence solve + projection, density solve +
projection, G2P, displacements) over the synthetic side-drop dam break at 256^3 with 8 particles per cell
(BASELINE.json configs[3]; --n/--scene select another size for quick runs).  Inputs are generated on the host
from a fixed seed and uploaded before the timed region.  Rank 0 prints ONE JSON line.

For N > 1 every rank runs the whole scene on its own GPU (independent replicas, no data-path collective):
the z-slab decomposition of the pressure solve is not built yet, so N > 1 measures replica throughput only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
DIRECTION_BYTES_PER_ROW = 53   # fused direction+SpMV kernel: read z 8, p 8, diag 8, 3 float terms 12, flags 1; write p 8, Ap 8
ITERATION_BYTES_PER_ROW = 109  # + update kernel: read p 8, Ap 8, x 8, r 8, 1/diag 8; write x 8, r 8, z 8


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", type=int, default=int(os.environ.get("FLIP_BENCH_N", "256")))
    ap.add_argument("--ppc", type=int, default=8)
    ap.add_argument("--scene", default="dam_break", choices=["dam_break", "cube_drop"])
    ap.add_argument("--dt", type=float, default=0.005)
    ap.add_argument("--tol", type=float, default=1e-15)
    ap.add_argument("--max-iter", type=int, default=20000)
    ap.add_argument("--precond", default=os.environ.get("FLIP_BENCH_PRECOND", "jacobi"), choices=["jacobi", "mg"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=40, help="CG iterations per solve timed on the CPU")
    return ap.parse_args()


def make_scene(args):
    from fluidsimulation_amd import scenes
    exact = args.n <= 64
    if args.scene == "dam_break":
        return scenes.dam_break(args.n, ppc=args.ppc, seed=12345, exact_levelset=exact)
    return scenes.cube_drop(args.n, ppc=args.ppc, seed=12345, exact_levelset=exact)


def cpu_baseline(sc, args, gpu_iters):
    """Times the CPU oracle (OpenMP port of the same stages) on this host: every non-solver stage once at
    full size plus `cpu_iters` CG iterations per solve, extrapolated to the iteration counts the GPU run
    needed.  Reported beside the GPU number; it is a baseline, not a target."""
    from oracle import Oracle
    threads = min(os.cpu_count() or 1, 16)
    o = Oracle(sc.desc(), threads=threads)
    dt = args.dt
    z32 = lambda: np.zeros(o.shape, np.float32)  # noqa: E731
    t = {}
    t0 = time.perf_counter()
    cell_id, cell_start, perm, _ = o.bin_particles(sc.pos)
    pm1, pm2 = o.particle_mask(cell_start, 1), o.particle_mask(cell_start, 2)
    t["bin"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    uvw = o.p2g(sc.pos, sc.vel, cell_start, perm, pm2)
    t["p2g"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    valid = [pm2.copy() for _ in range(3)]
    uvw, valid, active = o.extrapolate(uvw, valid, pm2.copy(), 2)
    prev = [a.copy() for a in uvw]
    uvw = [uvw[0], o.add_gravity(uvw[1], active, dt), uvw[2]]
    label, index, n_rows = o.index_cells(pm1, sc.phi_fluid, sc.phi_solid, [z32(), z32(), z32()])
    wm = o.weight_mask(sc.phi_fluid, sc.phi_fluid_bg)
    w = o.face_fractions(sc.phi_solid, sc.phi_solid_bg, wm)
    o.combine_weights(w, None)
    t["grid"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    dens, _ = o.density(sc.pos, sc.mass, cell_start, perm, pm2, sc.phi_fluid)
    t["density"] = time.perf_counter() - t0
    t_iter = []
    p = None
    for kind in (0, 1):
        t0 = time.perf_counter()
        off, diag, rhs = o.build_system(kind, dt, label, sc.phi_fluid, sc.phi_solid, w, uvw=uvw, density=dens)
        rowptr, col, val, b = o.to_csr(index, off, diag, rhs)
        t[f"build{kind}"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        p, it, _ = o.cg_jacobi(rowptr, col, val, b, tol=args.tol, max_iter=args.cpu_iters)
        t_iter.append((time.perf_counter() - t0) / max(it, 1))
        t0 = time.perf_counter()
        if kind == 0:
            uvw, valid = o.project_velocity(dt, index, p, sc.phi_fluid, w, uvw)
        else:
            disp = o.project_displacement(dt, index, p, sc.phi_fluid, w)
        t[f"project{kind}"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    diff = o.update_diff(uvw, prev)
    new_vel = o.g2p(sc.pos, sc.vel, uvw, diff, 0.95)
    o.apply_displacements(sc.pos, disp)
    t["g2p"] = time.perf_counter() - t0
    fixed = sum(t.values())
    total = fixed + t_iter[0] * gpu_iters[0] + t_iter[1] * gpu_iters[1]
    return {"value": 1.0 / total, "unit": "substeps/s", "cores": threads, "kind": "port",
            "sample": f"all non-solver stages once at full size ({fixed:.2f} s) + {args.cpu_iters} CG iterations per solve "
                      f"({t_iter[0] * 1e3:.1f}/{t_iter[1] * 1e3:.1f} ms per iteration), extrapolated to the GPU run's "
                      f"{gpu_iters[0]}+{gpu_iters[1]} iterations",
            "stage_seconds": {k: round(v, 4) for k, v in t.items()}}


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from fluidsimulation_amd import FlipContext, FlipSubstep

    sc = make_scene(args)
    ctx = FlipContext(sc.desc(), device=local)
    ctx.upload_scene(sc)
    from fluidsimulation_amd import PRECOND_JACOBI, PRECOND_MULTIGRID
    step = FlipSubstep(ctx, tol=args.tol, max_iter=args.max_iter,
                       precond=PRECOND_MULTIGRID if args.precond == "mg" else PRECOND_JACOBI)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step.step(args.dt)
    ctx.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    infos = []
    for _ in range(args.steps):
        infos.append(step.step(args.dt))
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        solves = [i[k] for i in infos for k in ("divergence", "density") if i[k] is not None]
        rows = solves[0]["rows"]
        spmv_s = sum(s["spmv_seconds"] for s in solves)
        spmv_n = sum(s["spmv_launches"] for s in solves)
        avg_launch = spmv_s / max(spmv_n, 1)
        achieved = rows * DIRECTION_BYTES_PER_ROW / avg_launch / 1e9 if spmv_n else 0.0
        iters = sum(s["iterations"] for s in solves)
        pcg_s = sum(s["seconds"] for s in solves)
        pcg_gbs = iters * rows * ITERATION_BYTES_PER_ROW / pcg_s / 1e9 if pcg_s > 0 else 0.0
        out = {
            "metric": "FLIP substeps/s (hot path, one MI355X per replica) + PCG GB/s",
            "value": world * args.steps / elapsed,
            "unit": "substeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.scene} {args.n}^3, {args.ppc} particles/cell, density projection on, "
                                   f"{args.precond} PCG tol {args.tol:g}",
                       "cells": sc.n_cells, "particles": int(sc.pos.shape[0]), "rows": rows,
                       "parallelism": "single GPU" if world == 1 else f"{world} independent replicas"},
            "pcg": {"iterations_per_step": iters / args.steps, "seconds_per_step": pcg_s / args.steps,
                    "effective_GBps": pcg_gbs, "pct_hbm_peak": 100.0 * pcg_gbs / HBM_PEAK_GBS,
                    "bytes_per_row_per_iteration": ITERATION_BYTES_PER_ROW,
                    "converged": all(s["converged"] == 1 for s in solves)},
            "roofline": {"bound": "hbm", "kernel": "k_pcg_direction", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "bytes_per_row": DIRECTION_BYTES_PER_ROW, "rows_per_launch": rows,
                         "avg_launch_us": avg_launch * 1e6, "launches_timed": spmv_n},
        }
        if not args.no_cpu_baseline:
            first = infos[0]
            gi = (first["divergence"]["iterations"], first["density"]["iterations"] if first["density"] else 0)
            out["cpu_baseline"] = cpu_baseline(sc, args, gi)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
