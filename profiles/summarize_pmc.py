"""Per-kernel mean of rocprofv3 --pmc counters.

    python summarize_pmc.py <counter_collection.csv> <counter>[,<counter>...] [out.csv] [kernel-name filter]

The mean is taken over the launches that did work: a launch whose first counter is below a tenth of that kernel's
largest value is left out (the solver enqueues its iterations in batches, and the launches of a batch that follow
convergence return at once).  `launches` counts the launches averaged, `skipped` the ones left out.
"""
import csv
import sys
from collections import defaultdict

path, counters = sys.argv[1], sys.argv[2].split(",")
only = sys.argv[4] if len(sys.argv) > 4 else None
vals = defaultdict(lambda: defaultdict(list))
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        c = row.get("Counter_Name")
        if c not in counters:
            continue
        name = row["Kernel_Name"].split("(")[0]
        if only and only not in name:
            continue
        vals[name][c].append(float(row["Counter_Value"]))
mean, used, skipped = {}, {}, {}
for k, per in vals.items():
    first = per[counters[0]]
    floor = 0.1 * max(first) if first else 0.0
    keep = [q for q, v in enumerate(first) if v >= floor]
    used[k], skipped[k] = len(keep), len(first) - len(keep)
    mean[k] = {}
    for c in counters:
        v = per[c]
        # the counters of one pass come in the same launch order; a counter collected alone keeps all its launches
        sel = [v[q] for q in keep] if len(v) == len(first) else v
        mean[k][c] = sum(sel) / max(len(sel), 1)
rows = sorted(mean, key=lambda k: -mean[k][counters[0]] * used[k])
out = open(sys.argv[3], "w") if len(sys.argv) > 3 and sys.argv[3] != "-" else sys.stdout
out.write("kernel,launches," + ",".join(f"mean_{c}" for c in counters) + ",skipped\n")
for k in rows:
    out.write(f"\"{k}\",{used[k]}," + ",".join(f"{mean[k][c]:.1f}" for c in counters) + f",{skipped[k]}\n")
