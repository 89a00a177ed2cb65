#!/usr/bin/env python3
"""Per-leg kernel time out of a rocprofv3 kernel trace of `bench_legs.py --pmc-child` (tools/profile_legs.sh): the legs
are told apart by the marker kernel launched in front of each; a leg's time per launch = the sum of its pgsd kernels'
durations / its launches.  Prints one JSON object; copies the stats CSV next to it."""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_legs as L

out = sys.argv[1]
child = json.load(open(os.path.join(out, "child.json")))
trace = sorted(glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)[-1]
rows = []
with open(trace, newline="") as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
rows.sort()
legs, leg = {}, -1
for _, name, ns in rows:
    if L.MARKER in name:
        leg += 1
        continue
    if "pgsd_amd::" not in name or "select_" in name or not (0 <= leg < len(child["order"])):
        continue
    d = legs.setdefault(child["order"][leg], {"ns": 0, "kernels": {}})
    d["ns"] += ns
    k = d["kernels"].setdefault(name.split("(")[0].replace("void ", ""), [0, 0])
    k[0] += 1
    k[1] += ns
N = 10_000_000
algo = {"config2": 56 << 20, "config4_sph": 224 * N, "config4_union": 328 * N, "config5_read": 56 * N,
        "gather_uniform": 60 * N, "gather_hilbert": 60 * N}
res = {}
for name, d in legs.items():
    n = child["launches"][name]
    us = d["ns"] / n / 1e3
    res[name] = {"launches": n, "avg_us_per_launch": round(us, 2), "algorithmic_bytes": algo[name],
                 "frac_of_8TBps": round(algo[name] / (us * 1e-6) / 8e12, 4),
                 "kernels": {k: {"dispatches": v[0], "avg_us": round(v[1] / v[0] / 1e3, 2)} for k, v in d["kernels"].items()}}
print(json.dumps({"source": "rocprofv3 --kernel-trace of bench_legs.py --pmc-child --reps 12", "legs": res}, indent=1))
