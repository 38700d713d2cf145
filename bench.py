"""Benchmark of the FLIP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" is one pass of the hot path (binning, P2G + density, extrapolation, forces, labels, face fractions,
divergence solve + projection, density solve + projection, G2P, displacements; the call order of
src/FluidSimulator.cpp:307-446) over the synthetic side-drop dam break at 256^3 with 8 particles per cell
(--n/--ppc select another size for quick runs).  Inputs are generated on the host from a fixed seed and uploaded
before the timed region.  Rank 0 prints ONE JSON line.

N = 1: one context, the fused solver loop.
N > 1: z-slab decomposition, one process per GPU (fluidsimulation_amd/distributed.py): the container is N
times as long along z and every rank owns one 256^3-sized slab of it plus ghost planes (weak scaling); the CG
scalars are all-reduced and a few planes per slab face are exchanged every iteration over RCCL (overlapping
additive-Schwarz preconditioner), particles near a face are exchanged once per step.  `value` counts slab-substeps: N x (substeps of the long domain) per second.
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
# Algorithmic bytes (minimum traffic if every array were touched once per kernel):
DIRECTION_BYTES_PER_ROW = 53   # fused direction + SpMV: the matrix-free SpMV (read p 8, diagonal 8, 3 float couplings 12,
                               # cell type 1; write Ap 8 = 37) plus the direction update folded into it (read z 8, write p 8)
UPDATE_BYTES_PER_ROW = 56      # read p 8, Ap 8, x 8, r 8, 1/diag 8; write x 8, r 8 (z is rewritten by the V-cycle)
VCYCLE_BYTES_PER_ROW = 226     # level 0: pre 44 + restrict 44 + correct/post 52 + post 52 = 192; coarse levels ~ +1/7 of a
                               # level with double couplings (240): 34
JACOBI_ITERATION_BYTES = DIRECTION_BYTES_PER_ROW + UPDATE_BYTES_PER_ROW + 8          # + write z
MG_ITERATION_BYTES = DIRECTION_BYTES_PER_ROW + UPDATE_BYTES_PER_ROW + VCYCLE_BYTES_PER_ROW
P2G_BYTES_PER_PARTICLE, P2G_BYTES_PER_BAND_CELL = 48, 24   # read pos + vel; write 3 velocity components
G2P_BYTES_PER_PARTICLE, G2P_BYTES_PER_BAND_CELL = 72, 48   # read pos + vel, write vel; read velocity + difference grids


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=int(os.environ.get("FLIP_BENCH_N", "256")))
    ap.add_argument("--ppc", type=int, default=8)
    ap.add_argument("--dt", type=float, default=0.005)
    ap.add_argument("--tol", type=float, default=1e-15)
    ap.add_argument("--max-iter", type=int, default=20000)
    ap.add_argument("--precond", default=os.environ.get("FLIP_BENCH_PRECOND", "mg"), choices=["jacobi", "mg"])
    ap.add_argument("--parallel", default="auto", choices=["auto", "slabs", "replicas"],
                    help="N > 1: z-slab decomposition of one long domain (auto) or independent replicas")
    ap.add_argument("--ghost", type=int, default=8, help="ghost planes on either side of a slab")
    ap.add_argument("--overlap", type=int, default=int(os.environ.get("FLIP_BENCH_OVERLAP", "-1")),
                    help="N > 1: planes of either neighbour's slab the local preconditioner also covers (additive Schwarz); "
                         "0 = disjoint blocks, default 4")
    ap.add_argument("--split-phase", action="store_true", help="N = 1: drive the solver through the split-phase entry points")
    ap.add_argument("--full-substep", action="store_true",
                    help="N = 1: also rebuild the level set from the particles, constrain_velocity and RK3 advection (the whole "
                         "loop body of advance_flip_pic except the fluid source); informational, not the headline configuration")
    ap.add_argument("--sequential-solves", action="store_true",
                    help="N = 1: divergence solve, projection, density solve one after the other (the reference's order) instead "
                         "of both solves side by side on two streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-iters", type=int, default=40, help="CG iterations per solve timed on the CPU")
    return ap.parse_args()


def cpu_baseline(sc, args, jacobi_iters):
    """Times the CPU oracle (OpenMP port of the same stages, Jacobi CG as the reference runs it) on this host:
    every non-solver stage once at full size plus `cpu_iters` CG iterations per solve, extrapolated to the
    iteration counts Jacobi CG needs on this scene (measured on the GPU).  A baseline, not a target."""
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
    reference_solver = None
    for kind in (0, 1):
        t0 = time.perf_counter()
        off, diag, rhs = o.build_system(kind, dt, label, sc.phi_fluid, sc.phi_solid, w, uvw=uvw, density=dens)
        rowptr, col, val, b = o.to_csr(index, off, diag, rhs)
        t[f"build{kind}"] = time.perf_counter() - t0
        if kind == 0:
            reference_solver = reference_pcgsolver(rowptr, col, val, b, sc.phi_fluid[label == 1] >= 0)
        o.cg_jacobi(rowptr, col, val, b, tol=args.tol, max_iter=2)   # thread-pool and page warm-up, not timed
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
    o.g2p(sc.pos, sc.vel, uvw, diff, 0.95)
    o.apply_displacements(sc.pos, disp)
    t["g2p"] = time.perf_counter() - t0
    fixed = sum(t.values())
    total = fixed + t_iter[0] * jacobi_iters[0] + t_iter[1] * jacobi_iters[1]
    return {"value": 1.0 / total, "unit": "substeps/s", "cores": threads, "kind": "port", "reference_solver": reference_solver,
            "sample": f"all non-solver stages once at full size ({fixed:.2f} s) + {args.cpu_iters} Jacobi-CG iterations per "
                      f"solve ({t_iter[0] * 1e3:.1f}/{t_iter[1] * 1e3:.1f} ms per iteration), extrapolated to the "
                      f"{jacobi_iters[0]}+{jacobi_iters[1]} iterations Jacobi CG needs on this scene (counted on the GPU); particle state "
                      f"as after the timed steps",
            "stage_seconds": {k: round(v, 4) for k, v in t.items()}}


def reference_pcgsolver(rowptr, col, val, b, dependent):
    """The reference's own headers/pcgsolver (PCGSolver<double>::solve: MIC(0)-preconditioned CG, one thread),
    compiled from the reference's header into oracle/_ref/libref.so, on the symmetric liquid block of the divergence
    system.  The as-run reference calls Eigen's CG instead; this is the one solver of the reference that can be
    built here, timed for scale.  None when the shim is absent."""
    import ctypes as C
    import scipy.sparse as sp
    from oracle.binding import load_reference_shim
    ref = load_reference_shim()
    if ref is None:
        return None
    n = rowptr.size - 1
    keep = ~np.asarray(dependent, bool)
    A = sp.csr_matrix((val, col, rowptr), shape=(n, n))[keep][:, keep].tocsr()
    A.sort_indices()
    m = A.shape[0]
    rp, ci, va = A.indptr.astype(np.int32), A.indices.astype(np.int32), np.ascontiguousarray(A.data, np.float64)
    rhs = np.ascontiguousarray(b[keep], np.float64)
    x = np.zeros(m)
    res, it = C.c_double(0), C.c_int32(0)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731
    t0 = time.perf_counter()
    ok = ref.ref_pcg_solve(m, ptr(rp), ptr(ci), ptr(va), ptr(rhs), 1e-12, 2000, ptr(x), C.byref(res), C.byref(it))
    return {"kind": "reference", "what": "headers/pcgsolver PCGSolver<double>::solve (MIC(0) PCG, tolerance factor 1e-12, 1 thread) on "
                                         "the liquid block of the divergence system, matrix set-up included",
            "rows": int(m), "seconds": time.perf_counter() - t0, "iterations": int(it.value), "converged": bool(ok)}


def stored_traffic(kernel_prefix):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC passes of this very command (FETCH_SIZE and
    WRITE_SIZE are collected in separate runs, so they cannot be read live here): 2 x FETCH_SIZE (the gfx950 correction
    for streaming reads) + WRITE_SIZE, both reported in KB.  None when the files are absent."""
    import csv
    vals = {}
    for key, name in (("fetch", "r01_d_bench256_mg_pmc_fetch_size.csv"), ("write", "r01_d_bench256_mg_pmc_write_size.csv")):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            return None
        with open(path, newline="") as f:
            for row in csv.reader(f):
                if row and row[0].startswith(kernel_prefix):
                    vals[key] = float(row[2])
    if len(vals) != 2:
        return None
    return (2.0 * vals["fetch"] + vals["write"]) * 1024.0


def time_stage(torch, fn, reps=3):
    """Average device time of `fn` (kernels on torch's current stream, which is the context's stream)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    total = 0.0
    for _ in range(reps):
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        total += e0.elapsed_time(e1)
    return total / reps * 1e-3


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    # rehearsal on a one-GPU box: FLIP_BENCH_REHEARSE=1 puts every rank on device 0 and uses gloo (RCCL refuses two
    # ranks on one device); the communication layer then stages through the host.  Not a measurement.
    rehearse = os.environ.get("FLIP_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from fluidsimulation_amd import FlipContext, FlipSubstep, PRECOND_JACOBI, PRECOND_MULTIGRID, scenes
    from fluidsimulation_amd.distributed import Comm, DistributedSubstep, SlabLayout

    precond = PRECOND_MULTIGRID if args.precond == "mg" else PRECOND_JACOBI
    slabs = (world > 1 and args.parallel != "replicas") or (world == 1 and args.split_phase)
    if world > 1 and slabs:
        layout = SlabLayout(scenes.slab_box_planes(args.n, world), world, args.ghost)
        sc = scenes.dam_break_slab(args.n, ppc=args.ppc, seed=12345, z_repeat=world, z_planes=layout.extended(rank))
    else:
        sc = scenes.dam_break_slab(args.n, ppc=args.ppc, seed=12345)
        layout = SlabLayout(sc.dims[2], 1, 0)
    # particles this rank owns (its ghost copies belong to the neighbours' counts)
    o0, o1 = layout.owned(rank) if world > 1 and slabs else (0, sc.dims[2])
    e0 = layout.extended(rank)[0] if world > 1 and slabs else 0
    plane = np.floor(sc.pos[:, 2] + 0.5).astype(np.int64) - sc.origin[2] + e0
    n_owned = int(((plane >= o0) & (plane < o1)).sum())
    ctx = FlipContext(sc.desc(), device=local)
    ctx.reserve_particles(sc.pos.shape[0] * 5 // 4 + 4096)
    ctx.upload_scene(sc)
    if slabs:
        overlap = min(4 if args.overlap < 0 else args.overlap, max(args.ghost - 1, 0)) if world > 1 else 0
        step = DistributedSubstep(ctx, layout, Comm(), device=local, tol=args.tol, max_iter=args.max_iter, precond=precond,
                                  overlap=overlap)
    else:
        extra = {}
        if args.full_substep:
            extra = dict(constrain=True, advect=True, levelset_radius=1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0)
        step = FlipSubstep(ctx, tol=args.tol, max_iter=args.max_iter, precond=precond,
                           concurrent_solves=not args.sequential_solves, **extra)

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
    n_local = ctx.particle_count()
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearse else "cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        cnt = torch.tensor([float(sc.dims[0] * sc.dims[1] * (o1 - o0)), float(n_owned), float(infos[0]["divergence"]["cg_rows"])], dtype=torch.float64,
                           device="cpu" if rehearse else "cuda")
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        tot_cells, tot_particles, tot_rows = (int(v) for v in cnt.tolist())
    else:
        tot_cells, tot_particles, tot_rows = sc.n_cells, n_local, infos[0]["divergence"]["cg_rows"]

    # ---- per-kernel measurements on this rank, outside the timed region ----
    solves = [i[k] for i in infos for k in ("divergence", "density") if i[k] is not None]
    rows = solves[0]["cg_rows"]
    iters = sum(s["iterations"] for s in solves)
    paired = not slabs and not args.sequential_solves
    # side by side, the two solves of a step overlap in time: the solver's share of the step is the longer of the two
    pcg_s = sum(max(i[k]["seconds"] for k in ("divergence", "density") if i[k] is not None) for i in infos) if paired \
        else sum(s["seconds"] for s in solves)
    it_bytes = MG_ITERATION_BYTES if args.precond == "mg" else JACOBI_ITERATION_BYTES
    pcg_gbs = iters * rows * it_bytes / pcg_s / 1e9 if pcg_s > 0 else 0.0
    spmv_s = sum(s["spmv_seconds"] for s in solves)
    spmv_n = sum(s["spmv_launches"] for s in solves)
    avg_launch = spmv_s / spmv_n if spmv_n else None
    achieved = rows * DIRECTION_BYTES_PER_ROW / avg_launch / 1e9 if avg_launch else None
    kernels = {}
    jacobi_iters = None
    if rank == 0:
        ctx.bin_particles(want_outside=False)
        t_p2g = time_stage(torch, ctx.p2g)
        band = int(ctx.download("active").astype(np.int64).sum())
        ctx.update_diff()
        t_g2p = time_stage(torch, lambda: ctx.g2p(0.95))
        p2g_bytes = P2G_BYTES_PER_PARTICLE * n_local + P2G_BYTES_PER_BAND_CELL * band
        g2p_bytes = G2P_BYTES_PER_PARTICLE * n_local + G2P_BYTES_PER_BAND_CELL * band
        kernels = {
            "k_p2g": {"bound": "hbm", "achieved": p2g_bytes / t_p2g / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": p2g_bytes / t_p2g / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": t_p2g * 1e6,
                      "algorithmic_bytes": p2g_bytes, "note": "the same launch also accumulates the density field"},
            "k_g2p": {"bound": "hbm", "achieved": g2p_bytes / t_g2p / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": g2p_bytes / t_g2p / 1e9 / HBM_PEAK_GBS, "traffic": None, "avg_launch_us": t_g2p * 1e6,
                      "algorithmic_bytes": g2p_bytes},
        }
        if world == 1 and not args.no_cpu_baseline and not slabs and not args.full_substep:
            # iteration counts of the reference's own solver (Jacobi CG) on this scene, for the CPU extrapolation
            probe = FlipSubstep(ctx, tol=args.tol, max_iter=args.max_iter, precond=PRECOND_JACOBI).step(args.dt)
            jacobi_iters = (probe["divergence"]["iterations"], probe["density"]["iterations"])

    traffic = None
    if rank == 0 and world == 1 and args.n == 256 and args.ppc == 8 and args.precond == "mg" and not args.full_substep:
        traffic = stored_traffic("void k_pcg_direction<true>")
        for name, prefix in (("k_p2g", "void k_p2g_block"), ("k_g2p", "k_g2p")):
            if name in kernels:
                kernels[name]["traffic"] = stored_traffic(prefix)
    if rank == 0:
        parallelism = "single GPU"
        if world > 1:
            parallelism = (f"{world} z-slabs of one {world}x long domain, ghost {args.ghost}, preconditioner overlap {overlap}, RCCL "
                           f"all-reduces + plane exchanges per CG iteration" if slabs else f"{world} independent replicas")
        out = {
            "metric": "FLIP substeps/s (hot path; 256^3-sized slab per MI355X) + PCG GB/s",
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
            "config": {"workload": f"dam break {args.n}^3" + (f" x{world} along z" if world > 1 and slabs else "")
                                   + f", {args.ppc} particles/cell, density projection on, {args.precond} PCG tol {args.tol:g}"
                                   + (", with level set rebuild, constrain_velocity and advection" if args.full_substep else "")
                                   + (", both solves side by side" if paired else ""),
                       "cells": tot_cells, "particles": tot_particles, "rows": tot_rows, "parallelism": parallelism},
            "pcg": {"iterations_per_step": iters / args.steps, "seconds_per_step": pcg_s / args.steps,
                    "effective_GBps": pcg_gbs, "pct_hbm_peak": 100.0 * pcg_gbs / HBM_PEAK_GBS,
                    "bytes_per_row_per_iteration": it_bytes, "rows_this_rank": rows,
                    "converged": all(s["converged"] == 1 for s in solves)},
            "roofline": {"bound": "hbm", "kernel": "k_pcg_direction", "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if achieved else None, "traffic": traffic,
                         "traffic_source": "profiles/r01_d_bench256_mg_pmc_{fetch,write}_size.csv (separate rocprofv3 --pmc "
                                           "passes of this command; 2 x FETCH_SIZE + WRITE_SIZE, bytes per launch)" if traffic else None,
                         "algorithmic_bytes_per_launch": rows * DIRECTION_BYTES_PER_ROW,
                         "bytes_per_row": DIRECTION_BYTES_PER_ROW, "rows_per_launch": rows,
                         "avg_launch_us": avg_launch * 1e6 if avg_launch else None, "launches_timed": spmv_n},
            "roofline_other_kernels": kernels,
        }
        if jacobi_iters is not None:
            # the CPU works on the state the device has reached, not on the fluid at rest: from rest the vectors of the
            # first solve are full of denormals, which the reference's build (no flush-to-zero) crawls through
            sc.pos, sc.vel = ctx.download_particles()
            out["cpu_baseline"] = cpu_baseline(sc, args, jacobi_iters)
        print(json.dumps(out))
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
