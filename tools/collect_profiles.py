#!/usr/bin/env python3
"""Copy the summaries of tools/profile_pack.sh (gpurun_out/prof_<tag>/) into profiles/ and derive
profiles/pack_traffic.json (HBM bytes per pack launch, corrected as MI355X_MICROARCH.md prescribes)."""
import csv
import glob
import hashlib
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")


def first(pattern):
    # gpurun merges new outputs into the directory without removing older ones: take the newest
    hits = sorted(glob.glob(os.path.join(src, pattern), recursive=True), key=os.path.getmtime)
    return hits[-1] if hits else None


def pack_rows(path, counter=None):
    out = {}
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            name = row.get("Kernel_Name", "")
            if "pgsd_amd::pack_" in name and (counter is None or row.get("Counter_Name") == counter):
                out.setdefault(name, []).append(row)
    return out


for size in ("10M", "1M"):
    st = first("stats_%s/**/*kernel_stats.csv" % size)
    if st:
        shutil.copy(st, os.path.join(dst, "%s_kernel_stats_%s.csv" % (tag, size)))
    tr = first("stats_%s/**/*kernel_trace.csv" % size)
    if tr:
        rows = pack_rows(tr)
        with open(tr, newline="") as f:
            header = next(csv.reader(f))
        with open(os.path.join(dst, "%s_kernel_trace_pack_%s.csv" % (tag, size)), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=header)
            w.writeheader()
            for v in rows.values():
                w.writerows(v)
    for kind, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        cc = first("%s_%s/**/*counter_collection.csv" % (kind, size))
        if cc:
            rows = pack_rows(cc, counter)
            with open(cc, newline="") as f:
                header = next(csv.reader(f))
            with open(os.path.join(dst, "%s_pmc_%s_size_%s.csv" % (tag, kind, size)), "w", newline="") as f:
                w = csv.DictWriter(f, fieldnames=header)
                w.writeheader()
                for v in rows.values():
                    w.writerows(v)
    for b in ("bench_under_rocprof", "bench_plain"):
        p = os.path.join(src, "%s_%s.json" % (b, size))
        if os.path.exists(p) and os.path.getsize(p):
            shutil.copy(p, os.path.join(dst, "%s_%s_%s.json" % (tag, b, size)))

# per-launch HBM bytes at 10 M particles (the bench layout)
fc, wc = first("fetch_10M/**/*counter_collection.csv"), first("write_10M/**/*counter_collection.csv")
if fc and wc:
    fetch = {k: [float(r["Counter_Value"]) for r in v] for k, v in pack_rows(fc, "FETCH_SIZE").items()}
    write = {k: [float(r["Counter_Value"]) for r in v] for k, v in pack_rows(wc, "WRITE_SIZE").items()}
    fk = sum(statistics.median(v) for v in fetch.values())
    wk = sum(statistics.median(v) for v in write.values())
    h = hashlib.sha256()       # = bench.py's pack_source_sha256()
    for name in ("pgsd_pack.hip", "pgsd_kernels.hpp"):
        with open(os.path.join(ROOT, "pgsd-sph_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    sha = h.hexdigest()
    tj = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python3 bench.py --particles "
                  "10000000 --steps 10 --warmup 2`, round %s (median over the dispatches); raw rows in profiles/%s_pmc_"
                  "fetch_size_10M.csv and profiles/%s_pmc_write_size_10M.csv" % (tag, tag, tag),
        "kernel": sorted(fetch),
        "layout": "HOOMD Scalar4 arrays (typeid in position.w)",
        "particles": 10000000,
        "pack_source_sha256": sha,
        "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
        "corrections": "FETCH_SIZE is in KiB and on gfx950 reports half of a wide coalesced 16 B/lane read stream "
                       "(MI355X_MICROARCH.md, HBM): bytes = KB * 1024 * 2. WRITE_SIZE: bytes = KB * 1024 (exact for "
                       "16 B/lane streaming stores; for this kernel's 12 B and 4 B per lane stores the known "
                       "280 000 000 chunk bytes are the calibration point).",
        "read_bytes_per_launch": int(fk * 2048), "write_bytes_per_launch": int(wk * 1024),
        "hbm_bytes_per_launch": int(fk * 2048) + int(wk * 1024),
    }
    with open(os.path.join(dst, "pack_traffic.json"), "w") as f:
        json.dump(tj, f, indent=2)
    print(json.dumps(tj, indent=1))
for f in sorted(os.listdir(dst)):
    if f.startswith(tag + "_kernel_stats"):
        print("==", f)
        print(open(os.path.join(dst, f)).read()[:1500])
