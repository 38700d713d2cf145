"""Reads a rocprofv3 kernel trace (csv) of bench.py and prints, for one timed substep in the middle of the run, every
kernel with its start relative to the substep's first kernel, its duration and the idle gap before it (measurement
aid).  A substep starts at a k_cell_ids launch that follows a k_update_particles launch."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
starts = [i for i, e in enumerate(ev) if e[2].startswith("k_fluid_levelset_occupancy") or e[2].startswith("k_occupancy")]
if len(starts) < 12:
    starts = [i for i, e in enumerate(ev) if e[2].startswith("k_cell_ids")]
a, b = starts[len(starts) // 2], starts[len(starts) // 2 + 1]
t0 = ev[a][0]
busy_end = t0
print(f"substep of {(ev[b][0] - t0) / 1e6:.3f} ms, {b - a} kernels")
agg = {}
for s, e, n in ev[a:b]:
    gap = s - busy_end
    short = n.split("(")[0][:48]
    solver = "k_pcg_" in n or "k_mg_" in n
    if not solver or gap > 20_000:
        print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap / 1e3:7.1f}  {short}")
    k = "solver" if solver else short
    agg[k] = agg.get(k, 0) + (e - s)
    if gap > 0:
        agg["(idle)"] = agg.get("(idle)", 0) + gap
    busy_end = max(busy_end, e)
print("-- totals (kernel time, overlapping kernels both counted)")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1]):
    print(f"{v / 1e3:9.1f} us  {k}")
