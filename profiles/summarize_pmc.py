"""Per-kernel mean of rocprofv3 --pmc counters.

    python summarize_pmc.py <counter_collection.csv> <counter>[,<counter>...] [out.csv] [kernel-name filter]
"""
import csv
import sys
from collections import defaultdict

path, counters = sys.argv[1], sys.argv[2].split(",")
only = sys.argv[4] if len(sys.argv) > 4 else None
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        c = row.get("Counter_Name")
        if c not in counters:
            continue
        name = row["Kernel_Name"].split("(")[0]
        if only and only not in name:
            continue
        tot[name][c] += float(row["Counter_Value"])
        cnt[name][c] += 1
rows = sorted(tot, key=lambda k: -tot[k][counters[0]])
out = open(sys.argv[3], "w") if len(sys.argv) > 3 and sys.argv[3] != "-" else sys.stdout
out.write("kernel,launches," + ",".join(f"mean_{c}" for c in counters) + "\n")
for k in rows:
    n = max(cnt[k][counters[0]], 1)
    out.write(f"\"{k}\",{n}," + ",".join(f"{tot[k][c] / max(cnt[k][c], 1):.1f}" for c in counters) + "\n")
