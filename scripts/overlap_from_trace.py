"""Reads a rocprofv3 kernel trace (csv) and reports, for the solver kernels, the sum of durations, the length of
the union of their intervals and the time two of them ran at once."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
iv = []
for r in rows:
    n = r["Kernel_Name"]
    if "k_pcg_" in n or "k_mg_" in n:
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), n, r.get("Queue_Id", "?")))
iv.sort()
tot = sum(e - s for s, e, _, _ in iv)
union = 0
overlap = 0
cur_s, cur_e = None, None
events = []
for s, e, _, _ in iv:
    events.append((s, 1))
    events.append((e, -1))
events.sort()
depth, last = 0, None
for t, d in events:
    if last is not None and depth >= 1:
        union += t - last
    if last is not None and depth >= 2:
        overlap += t - last
    depth += d
    last = t
queues = sorted({q for _, _, _, q in iv})
print(f"solver kernels: {len(iv)}  sum of durations {tot / 1e6:.2f} ms  union {union / 1e6:.2f} ms  two at once {overlap / 1e6:.2f} ms  queues {queues}")
span = iv[-1][1] - iv[0][0]
print(f"span first..last {span / 1e6:.2f} ms")
