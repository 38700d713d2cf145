"""Reads a rocprofv3 kernel trace (csv) of bench.py and classifies the time of the solve phases of the timed substeps:
how long two level-0 kernels ran at once, a level-0 kernel beside coarse-level kernels, coarse kernels only, one kernel
alone, and nothing at all (measurement aid)."""
import csv
import sys

FINE = ("k_pcg_direction", "k_pcg_update_pre", "k_mg_restrict<float, true>", "k_mg_post<float, true>", "k_mg_smooth<float, true, true>",
        "k_mg_smooth<float, false, true>", "k_mg_pre<float, true>")
rows = list(csv.DictReader(open(sys.argv[1])))
iv = []
for r in rows:
    n = r["Kernel_Name"]
    if "k_pcg_" in n or "k_mg_" in n:
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), any(f in n for f in FINE)))
iv.sort()
# split into solve phases: a gap of more than 300 us between solver kernels starts a new one
phases, cur = [], [iv[0]]
for a in iv[1:]:
    if a[0] - max(e for _, e, _ in cur[-50:]) > 300_000:
        phases.append(cur)
        cur = []
    cur.append(a)
phases.append(cur)
phases = [p for p in phases if len(p) > 500]
tot = {"fine+fine": 0, "fine+coarse": 0, "fine alone": 0, "coarse only": 0, "idle": 0}
span = 0
for p in phases[-20:]:
    ev = []
    for s, e, f in p:
        ev.append((s, 1, f)); ev.append((e, -1, f))
    ev.sort()
    nf = nc = 0
    last = ev[0][0]
    for t, d, f in ev:
        dt = t - last
        if nf >= 2: tot["fine+fine"] += dt
        elif nf == 1 and nc >= 1: tot["fine+coarse"] += dt
        elif nf == 1: tot["fine alone"] += dt
        elif nc >= 1: tot["coarse only"] += dt
        else: tot["idle"] += dt
        if f: nf += d
        else: nc += d
        last = t
    span += ev[-1][0] - ev[0][0]
n = len(phases[-20:])
print(f"{n} solve phases, {span / n / 1e6:.2f} ms each")
for k, v in tot.items():
    print(f"  {k:12s} {v / n / 1e6:6.2f} ms  {100.0 * v / span:5.1f} %")
