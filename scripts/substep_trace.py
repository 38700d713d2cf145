"""Reads a rocprofv3 kernel trace (csv) of bench.py and prints one timed substep of median length: every kernel outside
the solver with its start relative to the substep's first kernel, its duration, its queue and whether another kernel was
running when it started; the solver's kernels are summarised (measurement aid).  A substep starts at k_cell_ids."""
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_cell_ids")]
spans = [(ev[starts[q + 1]][0] - ev[starts[q]][0], q) for q in range(len(starts) - 1)]
med = statistics.median(s for s, _ in spans)
span, q = min(spans, key=lambda sq: abs(sq[0] - med))
a, b = starts[q], starts[q + 1]
t0 = ev[a][0]
print(f"substep of {span / 1e6:.3f} ms (median of {len(spans)}), {b - a} kernels")
busy_end = t0
solver = None
for s, e, n, qid in ev[a:b]:
    short = n.split("(")[0].replace("void ", "")[:44]
    is_solver = "k_pcg_" in n or "k_mg_" in n or "k_set_scalars" in n
    if is_solver:
        if solver is None:
            solver = [s, e, 0]
        solver[1] = max(solver[1], e)
        solver[2] += 1
    else:
        if solver is not None and s > solver[1]:
            print(f"{(solver[0] - t0) / 1e3:9.1f} us  {(solver[1] - solver[0]) / 1e3:8.1f} us  -- {solver[2]} solver kernels")
            solver = None
        if e - s > 8000 or (len(sys.argv) > 2 and sys.argv[2] == "all"):
            print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  q{qid} {'|| ' if s < busy_end else '   '}{short}")
    busy_end = max(busy_end, e)
if solver is not None:
    print(f"{(solver[0] - t0) / 1e3:9.1f} us  {(solver[1] - solver[0]) / 1e3:8.1f} us  -- {solver[2]} solver kernels")
