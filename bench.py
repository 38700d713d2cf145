"""Benchmark of the FLIP substep on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" is one pass of the loop body of FluidSimulator::advance_flip_pic (src/FluidSimulator.cpp:309-445) without
the fluid source: binning, fluid level set from the particles, P2G + density, extrapolation, constrain_velocity,
forces, labels, face fractions, divergence solve + projection, density solve + projection, extrapolation,
constrain_velocity, G2P, RK3 advection + push-out, displacements -- over the synthetic side-drop dam break at 256^3
nominal resolution with 8 particles per cell (--n/--ppc select another size for quick runs), substep size 0.001 s: the
reference limits a substep by compute_cfl (src/FluidSimulator.cpp:448-458, one cell per substep); at 256^3 the falling
column reaches 1 m/s = 0.26 cells per 0.001 s within the run, while the 0.005 s of earlier rounds exceeded one cell after a
dozen steps and let the particles clump (thousands per cell), which is not a state the reference's loop produces.  Inputs
are generated on the host from a fixed seed and uploaded before the timed region.  Rank 0 prints ONE JSON line.  `hot_path_only` in that
line is the same run without the level-set rebuild, constrain_velocity and advection (the stages outside the
transfer/solve/projection path), timed right after the headline steps.

N = 1: one context; both solves of a step side by side on two streams.
N > 1: one process per GPU (this script starts them itself when it is not already running under torch.distributed.run),
z-slab decomposition driven by libflip_dist.so (include/flip_dist.h): CG scalars all-reduced and a few planes per slab
face exchanged every iteration by RCCL calls enqueued on the context's stream, particles near a face exchanged once per
step (--python-comm: the same loop driven from Python over torch.distributed).
  --scaling weak (default): a container N times as long along z, one 256^3-sized slab per rank, per-GPU work fixed;
    `value` = N x substeps/s.  The decomposition is made for this: a slab keeps 8 ghost planes of either neighbour, a
    fixed cost per rank whatever the slab's thickness.
  --scaling strong: the one 256^3 box cut into N slabs; `value` = substeps of that box per second.  With 139 planes 8
    ranks own 17 planes each beside their 16 ghost planes: most of a rank's work is redundant and latency-bound.
  --full-domain: the container spans the whole nominal grid (with --n 512 --ppc 16: the ~1e9-particle dam break).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E peak
# Algorithmic bytes per unit (minimum traffic if every array were touched once per kernel).
BYTES = {
    # fused direction + SpMV: matrix-free SpMV (read p 8, diagonal 8, 3 float couplings 12, cell type 1; write Ap 8 = 37)
    # plus the direction update folded into it (read z 8, write p 8)
    "k_pcg_direction": 53,
    # CG update + both pre-smoothing steps: read r 8, Ap 8, x 8, p 8, 1/diag 4, diag 4, couplings 12, type 1; write x 8, r 8, z 8
    "k_pcg_update_pre": 85,
    "k_mg_restrict": 44,       # read r 8, z 8, diag 4, 1/diag 4, couplings 12, type 1, row id 4; write coarse b (8 per 8 rows) 1 (+2 slack)
    "k_mg_post": 52,           # read z 8, r 8, diag 4, 1/diag 4, couplings 12, type 1, row id 4, parent x 1 (8 per 8 rows); write t 8 (+2 slack)
    "k_mg_smooth": 52,         # read t 8, r 8, old z 8 (its value at the row), 1/diag 4, diag 4, couplings 12, type 1, row id 4; write z 8 -> 57; 52 kept from the round-1 accounting
}
JACOBI_ITERATION_BYTES = 53 + 56 + 8          # direction, update (56), write z
MG_ITERATION_BYTES = 53 + 56 + 226            # direction, update, V-cycle (level 0: 192, coarse levels ~34)
P2G_BYTES_PER_PARTICLE, P2G_BYTES_PER_BAND_CELL = 48, 24   # read pos + vel; write 3 velocity components
G2P_BYTES_PER_PARTICLE, G2P_BYTES_PER_BAND_CELL = 72, 48   # read pos + vel, write vel; read velocity + difference grids
# flip_update_particles (G2P + RK3 advection + displacements in one pass): read and write pos + vel; read the velocity,
# difference and displacement grids (9 doubles) and the solid level set (a float) of every band cell
UPDATE_BYTES_PER_PARTICLE, UPDATE_BYTES_PER_BAND_CELL = 96, 76
SORT_BYTES_PER_PARTICLE = 128                              # key pass 28, scatter pass 48 + 48 + 4
LEVELSET_BYTES_PER_PARTICLE, LEVELSET_BYTES_PER_CELL = 24, 8   # read pos; read phi_solid 4, write phi_fluid 4


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--n", type=int, default=int(os.environ.get("FLIP_BENCH_N", "256")))
    ap.add_argument("--ppc", type=int, default=8)
    ap.add_argument("--dt", type=float, default=0.001)
    ap.add_argument("--tol", type=float, default=1e-15)
    ap.add_argument("--max-iter", type=int, default=20000)
    ap.add_argument("--precond", default=os.environ.get("FLIP_BENCH_PRECOND", "mg"), choices=["jacobi", "mg"])
    ap.add_argument("--scaling", default="weak", choices=["strong", "weak"],
                    help="N > 1: one box cut into N slabs (strong) or one box-sized slab per rank of an N times longer container (weak)")
    ap.add_argument("--full-domain", action="store_true", help="container over the whole nominal grid instead of main.cpp's half-width box")
    ap.add_argument("--parallel", default="auto", choices=["auto", "slabs", "replicas"],
                    help="N > 1: z-slab decomposition (auto) or independent replicas")
    ap.add_argument("--ghost", type=int, default=8, help="ghost planes on either side of a slab")
    ap.add_argument("--overlap", type=int, default=int(os.environ.get("FLIP_BENCH_OVERLAP", "-1")),
                    help="N > 1: planes of either neighbour's slab the local preconditioner also covers (additive Schwarz); "
                         "0 = disjoint blocks, default 4")
    ap.add_argument("--split-phase", action="store_true",
                    help="N = 1: the slab path (split-phase solver, a one-rank RCCL communicator) through libflip_dist.so")
    ap.add_argument("--python-comm", action="store_true",
                    help="slab runs: drive the communication from Python (torch.distributed) instead of libflip_dist.so")
    ap.add_argument("--hot-path-only", action="store_true",
                    help="time only the transfer/solve/projection path (no level-set rebuild, constrain_velocity, advection)")
    ap.add_argument("--full-substep", action="store_true", help="accepted for compatibility: the whole substep is the default now")
    ap.add_argument("--sequential-solves", action="store_true",
                    help="N = 1: divergence solve, projection, density solve one after the other (the reference's order) instead "
                         "of both solves side by side on two streams")
    ap.add_argument("--python-driver", action="store_true",
                    help="one GPU: the substep as a sequence of C-ABI calls from Python (fluidsimulation_amd.substep) instead of one "
                         "call into libflip_dist.so's flip_dist_substep, which makes the same calls from C")
    ap.add_argument("--warm-start", action="store_true",
                    help="flip_set_warm_start 1: a solve starts from the previous substep's solution instead of zero")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def spawn_ranks(n):
    """Starts n rank processes of this script (before anything here has touched a GPU), with the rendezvous variables
    torch.distributed expects, and returns the first non-zero exit code (0 if all succeeded)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        code = p.wait()
        if code != 0 and rc == 0:
            rc = code
            for q in procs:      # a rank that failed leaves the others waiting in a collective
                if q.poll() is None:
                    q.terminate()
    return rc


def cpu_stage_seconds(o, sc, args):
    """Every non-solver stage of the hot path once at full size on the CPU oracle; returns (seconds per stage, the two CSR
    systems, the liquid mask of the divergence system's rows)."""
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
    systems = []
    disp = None
    for kind in (0, 1):
        t0 = time.perf_counter()
        off, diag, rhs = o.build_system(kind, dt, label, sc.phi_fluid, sc.phi_solid, w, uvw=uvw, density=dens)
        systems.append(o.to_csr(index, off, diag, rhs))
        t[f"build{kind}"] = time.perf_counter() - t0
        p = np.zeros(systems[-1][3].shape)
        t0 = time.perf_counter()
        if kind == 0:
            uvw, valid = o.project_velocity(dt, index, p, sc.phi_fluid, w, uvw)
        else:
            disp = o.project_displacement(dt, index, p, sc.phi_fluid, w)
        t[f"project{kind}"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    diff = o.update_diff(uvw, prev)
    o.g2p(sc.pos, sc.vel.copy(), uvw, diff, 0.95)
    o.apply_displacements(sc.pos.copy(), disp)
    t["g2p"] = time.perf_counter() - t0
    return t, systems, sc.phi_fluid[label == 1] >= 0


def cpu_baseline(sc, args):
    """The CPU oracle (OpenMP port of the same stages, Jacobi CG as the reference runs it) timed on this host's cores, once
    with 8 threads (what the reference pins Eigen to, src/FluidSimulator.cpp:658) and once with the cores available (16 at
    most): one whole hot-path substep MEASURED each time -- every non-solver stage once at full size and both Jacobi-CG
    solves run to the reference's tolerance.  About 10-20 s of CPU work.  A baseline, not a target."""
    from oracle import Oracle
    avail = min(os.cpu_count() or 1, 16)
    out = {}
    reference_solver = None
    iters = None
    for threads in sorted({min(8, avail), avail}):
        o = Oracle(sc.desc(), threads=threads)
        t, systems, dependent = cpu_stage_seconds(o, sc, args)
        t_solve, iters = [], []
        for kind, (rowptr, col, val, b) in enumerate(systems):
            if kind == 0 and reference_solver is None:
                reference_solver = reference_pcgsolver(rowptr, col, val, b, dependent)
            o.cg_jacobi(rowptr, col, val, b, tol=args.tol, max_iter=2)   # thread-pool and page warm-up, not timed
            t0 = time.perf_counter()
            _, it, _ = o.cg_jacobi(rowptr, col, val, b, tol=args.tol, max_iter=args.max_iter)
            t_solve.append(time.perf_counter() - t0)
            iters.append(int(it))
        fixed = sum(t.values())
        total = fixed + t_solve[0] + t_solve[1]
        out[threads] = {"value": 1.0 / total, "fixed_seconds": fixed, "solve_seconds": t_solve,
                        "ms_per_cg_iteration": [1e3 * a / max(b, 1) for a, b in zip(t_solve, iters)],
                        "stage_seconds": {k: round(v, 4) for k, v in t.items()}}
    top = out[avail]
    return {"value": top["value"], "unit": "substeps/s", "cores": avail, "kind": "port",
            "value_8_threads": out[min(8, avail)]["value"], "reference_solver": reference_solver,
            "cg_iterations": iters,
            "sample": f"one whole hot-path substep (no level-set rebuild, constrain, advection) at full size, measured: the stages "
                      f"({top['fixed_seconds']:.2f} s) and both Jacobi-CG solves to tol {args.tol:g} ({iters[0]} + {iters[1]} iterations, "
                      f"{top['solve_seconds'][0]:.2f} + {top['solve_seconds'][1]:.2f} s); particle state as after the timed steps",
            "by_threads": {str(k): v for k, v in out.items()}}


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


def kernel_source_id():
    """Hash of the kernel sources: a stored PMC profile is only quoted for the code it was taken on."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "fluidsimulation_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as f:
                h.update(f.read())
    return h.hexdigest()[:12]


def stored_traffic(args):
    """HBM bytes per launch from the committed rocprofv3 PMC passes of this command (FETCH_SIZE and WRITE_SIZE need
    passes of their own, so they cannot be read live): profiles/traffic.json, written by profiles/summarize_pmc.py, holds
    {kernel: bytes} plus the source hash and the configuration it was taken on, --steps and --warmup included (the scene
    state the counters saw).  Returns ({}, None) when absent or taken on other code or another configuration."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return {}, None
    with open(path) as f:
        d = json.load(f)
    if (d.get("source_id") != kernel_source_id() or d.get("n") != args.n or d.get("ppc") != args.ppc
            or d.get("hot_path_only", False) != args.hot_path_only or d.get("dt") != args.dt
            or d.get("steps") != args.steps or d.get("warmup") != args.warmup):
        return {}, None
    return d.get("bytes_per_launch", {}), d.get("files")


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


def stage_entry(seconds, per_step, algorithmic_bytes, what):
    """A stage of several kernels: reported beside the kernels, never picked as the roofline kernel."""
    gbs = algorithmic_bytes / seconds / 1e9
    return {"seconds": seconds, "per_step": per_step, "ms_per_step": seconds * per_step * 1e3, "algorithmic_bytes": algorithmic_bytes,
            "achieved_GBps": gbs, "frac_hbm_peak": gbs / HBM_PEAK_GBS, "what": what}


def roofline_entry(name, seconds, launches_per_step, algorithmic_bytes, traffic, note=None):
    gbs = algorithmic_bytes / seconds / 1e9
    e = {"kernel": name, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
         "traffic": traffic.get(name), "avg_launch_us": seconds * 1e6, "launches_per_step": launches_per_step,
         "ms_per_step": seconds * launches_per_step * 1e3, "algorithmic_bytes_per_launch": algorithmic_bytes}
    if note:
        e["note"] = note
    return e


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))
    # stdout carries the one JSON line and nothing else: libraries that greet on stdout (RCCL prints its version there
    # when a communicator is created) are sent to stderr, and the line is written to the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a line for another GPU count\n")
        sys.exit(2)
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    # rehearsal on a one-GPU box: FLIP_BENCH_REHEARSE=1 puts every rank on device 0 and uses gloo (RCCL refuses two
    # ranks on one device); the communication layer then stages through the host.  Not a measurement.
    rehearse = os.environ.get("FLIP_BENCH_REHEARSE") == "1"
    if rehearse:
        local = 0
    elif torch.cuda.device_count() < world:
        sys.stderr.write(f"bench.py: {world} ranks but {torch.cuda.device_count()} GPUs visible\n")
        sys.exit(2)
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
    weak = world > 1 and slabs and args.scaling == "weak"
    z_repeat = world if weak else 1
    if world > 1 and slabs:
        nz = scenes.slab_box_planes(args.n, z_repeat, args.full_domain)
        ghost = args.ghost
        if weak:
            layout = SlabLayout(nz, world, ghost)
        else:
            # strong scaling: cut by particle load (the planes inside the walls hold none), slabs at least `ghost` thick
            hw = scenes.SOLID_HALF_WIDTH
            weights = np.full(nz, 0.1)
            weights[hw:nz - hw] = 1.0
            while ghost > 1 and nz < world * ghost:
                ghost //= 2
            layout = SlabLayout.balanced(weights, world, ghost)
        sc = scenes.dam_break_slab(args.n, ppc=args.ppc, seed=12345, z_repeat=z_repeat, z_planes=layout.extended(rank),
                                   full_domain=args.full_domain)
    else:
        ghost = 0
        sc = scenes.dam_break_slab(args.n, ppc=args.ppc, seed=12345, full_domain=args.full_domain)
        layout = SlabLayout(sc.dims[2], 1, 0)
    # particles this rank owns (its ghost copies belong to the neighbours' counts)
    o0, o1 = layout.owned(rank) if world > 1 and slabs else (0, sc.dims[2])
    e0 = layout.extended(rank)[0] if world > 1 and slabs else 0
    plane = np.floor(sc.pos[:, 2] + 0.5).astype(np.int64) - sc.origin[2] + e0
    n_owned = int(((plane >= o0) & (plane < o1)).sum())
    ctx = FlipContext(sc.desc(), device=local)
    ctx.reserve_particles(sc.pos.shape[0] * 5 // 4 + 4096)
    ctx.upload_scene(sc)
    if args.warm_start:
        ctx.set_warm_start(True)
    radius = 1.01 * float(np.float32(sc.dx)) * np.sqrt(3.0) / 2.0   # src/FluidSource.cpp:67
    extra = {} if args.hot_path_only else dict(constrain=True, advect=True, levelset_radius=radius)
    overlap = 0
    if slabs:
        overlap = min(4 if args.overlap < 0 else args.overlap, max(ghost - 1, 0)) if world > 1 else 0
        if args.python_comm:
            step = DistributedSubstep(ctx, layout, Comm(), device=local, tol=args.tol, max_iter=args.max_iter, precond=precond,
                                      overlap=overlap, **extra)
        else:
            from fluidsimulation_amd.dist_native import NativeDistributedSubstep, TorchTransport
            comm = Comm()
            # RCCL refuses several ranks on one device: the rehearsal on a one-GPU box goes through the library's transport
            # callbacks over gloo instead
            step = NativeDistributedSubstep(ctx, layout, comm, device=local, tol=args.tol, max_iter=args.max_iter, precond=precond,
                                            overlap=overlap, transport=TorchTransport(comm, local) if rehearse and world > 1 else "rccl",
                                            force_slab=world == 1, **extra)
    elif args.python_driver or args.sequential_solves or args.hot_path_only:
        step = FlipSubstep(ctx, tol=args.tol, max_iter=args.max_iter, precond=precond,
                           concurrent_solves=not args.sequential_solves, **extra)
    else:
        # one context: flip_dist_substep makes the calls of FlipSubstep.step from C (flip_solve_pair for the two solves)
        from fluidsimulation_amd.dist_native import NativeDistributedSubstep
        step = NativeDistributedSubstep(ctx, layout, Comm(), device=local, tol=args.tol, max_iter=args.max_iter, precond=precond,
                                        overlap=0, transport="rccl", force_slab=False, **extra)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed(stepper, k):
        barrier()
        t0 = time.perf_counter()
        infos = [stepper.step(args.dt) for _ in range(k)]
        barrier()
        return time.perf_counter() - t0, infos

    for _ in range(args.warmup):
        step.step(args.dt)
    elapsed, infos = timed(step, args.steps)
    # The per-kernel timings come from extra substeps after the timed ones: with profiling on, two iterations of the
    # divergence solve run with the device to themselves (the density solve's stream is held back) and are bracketed by
    # events, which a timed substep must not pay for.
    ctx.set_profiling(True)
    prof_infos = [step.step(args.dt) for _ in range(2)]
    ctx.set_profiling(False)
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

    # ---- the same run without the stages outside the transfer/solve/projection path (single context only) ----
    hot = None
    if world == 1 and not slabs and not args.hot_path_only:
        # Without advection and level-set rebuild the particle velocities are not a physical state any more after a few
        # steps (they grow step by step while the positions stay), so the particles are put back afterwards: everything
        # measured below sees the state the timed run reached.  The work per step does not depend on the velocities.
        saved_state = ctx.download_particles() + (ctx.download_mass(),)
        hstep = FlipSubstep(ctx, tol=args.tol, max_iter=args.max_iter, precond=precond, concurrent_solves=not args.sequential_solves)
        # at most 20 steps: without advection the velocities grow without bound, and after some hundred steps the solves
        # of this leg no longer converge
        h_steps = min(args.steps, 20)
        h_elapsed, _ = timed(hstep, h_steps)
        ctx.upload_particles(*saved_state)
        hot = {"value": h_steps / h_elapsed, "unit": "substeps/s", "ms_per_step": 1e3 * h_elapsed / h_steps,
               "what": "binning, P2G + density, extrapolation, forces, labels, face fractions, both solves and projections, G2P, "
                       "displacements; fluid level set as left by the last full step"}

    # ---- per-kernel measurements on this rank, outside the timed region ----
    solves = [i[k] for i in infos for k in ("divergence", "density") if i[k] is not None]
    rows = solves[0]["cg_rows"]
    iters = sum(s["iterations"] for s in solves)
    # the C driver runs the two solves of a step side by side on one context and over slabs alike; the Python slab driver one after the other
    paired = not args.sequential_solves and not (slabs and args.python_comm)
    # side by side, the two solves of a step overlap in time: the solver's share of the step is the longer of the two
    pcg_s = sum(max(i[k]["seconds"] for k in ("divergence", "density") if i[k] is not None) for i in infos) if paired \
        else sum(s["seconds"] for s in solves)
    it_bytes = MG_ITERATION_BYTES if args.precond == "mg" else JACOBI_ITERATION_BYTES
    pcg_gbs = iters * rows * it_bytes / pcg_s / 1e9 if pcg_s > 0 else 0.0
    prof_solves = [i[k] for i in prof_infos for k in ("divergence", "density") if i[k] is not None]
    spmv_s = sum(s["spmv_seconds"] for s in prof_solves)
    spmv_n = sum(s["spmv_launches"] for s in prof_solves)
    # binning as the substep meets it: right after a whole substep (which ends with moved particles, the sort stale);
    # measured first, on the state the run has reached, before anything below touches the particles.  Every rank takes
    # the steps (a substep of a slab run is collective); rank 0 reports its own time.
    t_bin = 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        step.step(args.dt)
        barrier()                     # the substep's own work is done before the clock starts
        e0.record()
        ctx.bin_particles(want_outside=False)
        e1.record()
        e1.synchronize()
        t_bin += e0.elapsed_time(e1) * 1e-3 / 3
    n_local = ctx.particle_count()
    entries, stages = [], {}
    if rank == 0:
        traffic, traffic_files = stored_traffic(args) if world == 1 else ({}, None)
        its_per_step = iters / args.steps
        if spmv_n:
            entries.append(roofline_entry("k_pcg_direction", spmv_s / spmv_n, its_per_step, rows * BYTES["k_pcg_direction"], traffic,
                                          note=f"HIP events around {spmv_n} launches with the device otherwise idle"))
        if args.precond == "mg" and not slabs:
            try:
                kt = ctx.solver_kernel_times()
            except Exception:
                kt = {}
            for name in ("k_pcg_update_pre", "k_mg_restrict", "k_mg_post", "k_mg_smooth"):
                if name in kt and kt[name] > 0:
                    entries.append(roofline_entry(name, kt[name] * 1e-6, its_per_step, rows * BYTES[name], traffic,
                                                  note="one profiled iteration, device otherwise idle"))
            if kt.get("coarse_levels"):
                # 13 launches, not one kernel: a stage.  ~24 B per level-0 row is what the levels below 0 move: 169 B per row of a level
                # that reads float operator copies (the first two), 249 B below, over 1/8 + 1/64 + ... of the rows
                t_c = kt["coarse_levels"] * 1e-6
                stages["coarse_levels"] = stage_entry(t_c, its_per_step, rows * 24, "all launches of the V-cycle below level 0 of one "
                                                      "iteration, run alone (bound by launch and dependent-load latency; beside the other "
                                                      "solve's level-0 kernels they mostly hide)")
        stages["binning"] = stage_entry(t_bin, 1, SORT_BYTES_PER_PARTICLE * n_local, "whole stage after a real substep: cell ids, scan, slots, "
                                        "per-row reorder (stable order + half-cell ordered copy), bands, gather work list")
        # particle-side kernels and stages, timed alone on that state; the repeated G2P calls change the particle
        # velocities, so the particles are put back afterwards
        saved = ctx.download_particles() + (ctx.download_mass(),)
        if not args.hot_path_only:
            t_ls = time_stage(torch, lambda: ctx.fluid_levelset_from_particles(radius))
            stages["fluid_levelset"] = stage_entry(t_ls, 1, LEVELSET_BYTES_PER_PARTICLE * n_local + LEVELSET_BYTES_PER_CELL * sc.n_cells,
                                                   "whole stage: occupancy dilation, shells 0-1 for every cell, compaction, full search of the rest")
        t_p2g = time_stage(torch, ctx.p2g)
        band = int(ctx.download("active").astype(np.int64).sum())
        ctx.update_diff()
        if args.hot_path_only:
            t_g2p = time_stage(torch, lambda: ctx.g2p(0.95))
        else:
            t_upd = time_stage(torch, lambda: ctx.update_particles(0.95, args.dt, True, True))
        entries.append(roofline_entry("k_gather", t_p2g, 1, P2G_BYTES_PER_PARTICLE * n_local + P2G_BYTES_PER_BAND_CELL * band, traffic,
                                      note="flip_p2g = k_gather<true, 2, 2, 2> (P2G and, in the same walk, the density field) after k_gather_clear "
                                           "(zeros outside the band, ~4 % of the time)"))
        if args.hot_path_only:
            entries.append(roofline_entry("k_g2p", t_g2p, 1, G2P_BYTES_PER_PARTICLE * n_local + G2P_BYTES_PER_BAND_CELL * band, traffic))
        else:
            entries.append(roofline_entry("k_update_particles", t_upd, 1, UPDATE_BYTES_PER_PARTICLE * n_local + UPDATE_BYTES_PER_BAND_CELL * band, traffic,
                                          note="flip_update_particles: G2P, RK3 advection with push-out and the density displacements in one pass "
                                               "(k_update_particles<true, true>); the three separate kernels took 0.34 + 0.44 + 0.18 ms"))
        ctx.upload_particles(*saved)
    if rank == 0:
        parallelism = "single GPU"
        if world > 1:
            parallelism = (f"{world} z-slabs ({args.scaling} scaling), ghost {ghost}, preconditioner overlap {overlap}, owned planes "
                           f"{[layout.owned(r)[1] - layout.owned(r)[0] for r in range(world)]}, RCCL all-reduces + plane exchanges per CG "
                           f"iteration" if slabs else f"{world} independent replicas")
        gdims = (sc.dims[0], sc.dims[1], layout.nz if world > 1 and slabs else sc.dims[2])
        workload = (f"side-drop dam break, nominal {args.n}^3" + (", full-domain container" if args.full_domain else "")
                    + (f" x{world} along z" if weak else "")
                    + f" (dense box {gdims[0]}x{gdims[1]}x{gdims[2]} cells), {args.ppc} particles/cell, density projection on, "
                    + ("hot path only (no level-set rebuild, constrain_velocity, advection), " if args.hot_path_only
                       else "whole substep (level set from particles, constrain_velocity, RK3 advection included), ")
                    + f"{args.precond} PCG tol {args.tol:g}, drop_null_rows=1, eliminate_dependent_rows=1"
                    + (", both solves side by side" if paired else ""))
        entries.sort(key=lambda e: -e["ms_per_step"])
        for e in entries:
            e["dominant"] = False
        hbm = [e for e in entries if e.get("bound") == "hbm"]
        dominant = hbm[0] if hbm else None
        if dominant:
            dominant["dominant"] = True
        roofline = dict(dominant) if dominant else {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS,
                                                     "unit": "GB/s", "frac": None, "traffic": None}
        roofline["traffic_source"] = traffic_files if roofline.get("traffic") else None
        roofline["selection"] = ("the kernel (a name rocprofv3's kernel statistics list) with the largest time per step among those timed on "
                                 "their own; stages of several kernels are under `stages`")
        scaling = "weak" if (weak or (world > 1 and not slabs)) else "strong"
        units = world if scaling == "weak" else 1
        out = {
            "metric": "FLIP substeps/s (whole substep, side-drop dam break 256^3) + PCG GB/s",
            "value": units * args.steps / elapsed,
            "unit": "substeps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "cells": tot_cells, "particles": tot_particles, "rows": tot_rows, "parallelism": parallelism},
            "hot_path_only": hot,
            "pcg": {"iterations_per_step": iters / args.steps, "seconds_per_step": pcg_s / args.steps,
                    "warm_start": args.warm_start and not slabs,
                    "iterations_to_pcgsolver_rule_per_step": (sum(q["iterations_inf"] for q in solves) / args.steps
                                                              if all(q["iterations_inf"] >= 0 for q in solves) else None),
                    "effective_GBps": pcg_gbs, "pct_hbm_peak": 100.0 * pcg_gbs / HBM_PEAK_GBS,
                    "bytes_per_row_per_iteration": it_bytes, "rows_this_rank": rows,
                    "converged": all(s["converged"] == 1 for s in solves)},
            "roofline": roofline,
            "roofline_other_kernels": {e["kernel"]: e for e in entries if not e["dominant"]},
            "stages": stages,
        }
        if world == 1 and not args.no_cpu_baseline and not slabs:
            # the CPU works on the state the device has reached, not on the fluid at rest: from rest the vectors of the
            # first solve are full of denormals, which the reference's build (no flush-to-zero) crawls through
            sc.pos, sc.vel = ctx.download_particles()
            sc.phi_fluid = ctx.download("phi_fluid")
            out["cpu_baseline"] = cpu_baseline(sc, args)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
