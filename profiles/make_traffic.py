"""Writes profiles/traffic.json from the two per-kernel PMC summaries of one profiling round: HBM-side bytes per launch
= 2 x FETCH_SIZE (the gfx950 correction for wide streaming reads; other widths are uncalibrated) + WRITE_SIZE, both
reported by rocprofv3 in KB.  bench.py quotes these numbers only while the kernel sources hash to `source_id`.

    python profiles/make_traffic.py profiles/r03_x_pmc_fetch_size.csv profiles/r03_x_pmc_write_size.csv [n] [ppc] [dt] [steps] [warmup]
"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_id  # noqa: E402

# bench.py's kernel names -> the profiler's kernel names whose per-launch means add up to one launch of it
GROUPS = {
    "k_pcg_direction": ["void k_pcg_direction<true>"],
    "k_pcg_update_pre": ["k_pcg_update_pre"],
    "k_mg_restrict": ["void k_mg_restrict<float, true>"],
    "k_mg_post": ["void k_mg_post<float, true>"],
    "k_mg_smooth": ["void k_mg_smooth<float, true, true>"],
    "k_gather": ["void k_gather<true, 2, 2, 2>", "void k_gather_clear<true>"],
    "k_g2p": ["k_g2p"],
    "k_update_particles": ["void k_update_particles<true, true>"],
    "k_fluid_levelset": ["k_near_x", "k_near_axis", "void k_fluid_levelset_near<3>", "k_fluid_levelset_list", "k_fluid_levelset_far"],
    "binning": ["k_cell_ids", "k_scatter_slots", "k_reorder_pack", "k_reorder_outside", "k_band_x", "k_band_axis"],
}
# kernels launched more than once per stage launch
MULT = {"k_near_axis": 2, "k_band_axis": 2}


def read(path):
    out = {}
    with open(path, newline="") as f:
        for row in csv.reader(f):
            if row and row[0] != "kernel":
                out[row[0]] = float(row[2])
    return out


fetch, write = read(sys.argv[1]), read(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ppc = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dt = float(sys.argv[5]) if len(sys.argv) > 5 else 0.001
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 20
warmup = int(sys.argv[7]) if len(sys.argv) > 7 else 5
traffic = {}
for name, parts in GROUPS.items():
    if not all(p in fetch and p in write for p in parts if p != "k_reorder_outside"):
        continue
    traffic[name] = sum((2.0 * fetch.get(p, 0.0) + write.get(p, 0.0)) * 1024.0 * MULT.get(p, 1) for p in parts)
json.dump({"source_id": kernel_source_id(), "n": n, "ppc": ppc, "dt": dt, "steps": steps, "warmup": warmup, "hot_path_only": False,
           "files": [os.path.relpath(os.path.abspath(a), ROOT) for a in sys.argv[1:3]],
           "formula": "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes per launch, means over the launches of the profiled run",
           "bytes_per_launch": traffic}, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
