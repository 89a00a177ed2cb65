#!/bin/bash
# Which unit is busy in the tag-order gather?  rocprofv3 --pmc passes (one counter set per pass, --kernel-trace only)
# over tools/pack_bench.py for three permutations: identity, Hilbert memory order x lattice tags, uniformly random.
# Output: gpurun_out/gather_counters/<workload>_<set>/... + a one-line-per-pass summary.
set -u
OUT=$PWD/gpurun_out/gather_counters
rm -rf "$OUT"; mkdir -p "$OUT"
BENCH=$GRAFT_REPO_ROOT/tools/pack_bench.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1 || true
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum" "MemUnitBusy MemUnitStalled" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM" "TCP_TA_TCP_STATE_READ_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"; do
  i=$((i+1))
  for w in gather_identity gather_hilbert gather; do
    d="$OUT/${w}_set$i"
    rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$d" -- python3 $BENCH --workloads $w --N 10000000 --iters 6 --warmup 2 > "$d.out" 2> "$d.err" || echo "pass $w set$i ($set) failed: $(tail -2 $d.err | tr '\n' ' ')"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/*_set*")):
    if not os.path.isdir(d):
        continue
    files = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if "pack_tiles" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(os.path.basename(d), {k: round(sum(v) / len(v), 1) for k, v in acc.items()}, "launches", max([len(v) for v in acc.values()] or [0]))
PY
