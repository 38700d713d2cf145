"""Per-kernel mean of one rocprofv3 --pmc counter: python summarize_pmc.py <counter_collection.csv> <counter> [out.csv]"""
import csv
import sys
from collections import defaultdict

path, counter = sys.argv[1], sys.argv[2]
tot, cnt = defaultdict(float), defaultdict(int)
with open(path, newline="") as f:
    for row in csv.DictReader(f):
        if row.get("Counter_Name") != counter:
            continue
        name = row["Kernel_Name"].split("(")[0]
        tot[name] += float(row["Counter_Value"])
        cnt[name] += 1
rows = sorted(tot, key=lambda k: -tot[k])
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
out.write(f"kernel,launches,mean_{counter},total_{counter}\n")
for k in rows:
    out.write(f"\"{k}\",{cnt[k]},{tot[k] / cnt[k]:.1f},{tot[k]:.1f}\n")
