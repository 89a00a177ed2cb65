#!/bin/bash
# config 2 (2^20 particles, pack kernel only, SURVEY 8(d) protocol) measured 11 067 ns in round 2 and
# 11 715 ns in round 3 for the same pack_rows_kernel<256, 4, 4, true>.  Box spread or a regression?  ONE box, ONE
# session: the round-2 tree (git archive of 7eb681b built under tools/build/r02_tree, not tracked) and the current
# tree take turns under rocprofv3 --kernel-trace --stats; then the launch shapes the row kernel is instantiated for
# are interleaved launch by launch in one process (PGSD_PACK_ROWS_CFG), at 2^19, 2^20 and 2^21 rows.
set -u
OUT=$PWD/gpurun_out/r04/config2_ab
rm -rf "$OUT"; mkdir -p "$OUT"
CUR=$GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/tools/build/r02_tree
cd /tmp && export TMPDIR=/tmp
ARGS="--N 1048576 --iters 200 --warmup 20 --workloads hoomd_pvi,double4"
for pass in 1 2; do
  for tag in r02 cur; do
    tree=$CUR; [ $tag = r02 ] && tree=$OLD
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_${tag}_$pass" -- python3 $tree/tools/pack_bench.py $ARGS \
        > "$OUT/under_rocprof_${tag}_$pass.jsonl" 2> "$OUT/stats_${tag}_$pass.err" || echo "stats $tag $pass failed"
    f=$(find "$OUT/stats_${tag}_$pass" -name "*kernel_stats.csv" | head -1)
    echo "== $tag pass $pass" >> "$OUT/summary.txt"
    grep "pgsd_amd::pack" "$f" | cut -c1-220 >> "$OUT/summary.txt"
    python3 $tree/tools/pack_bench.py $ARGS --kernel-only 2>/dev/null | cut -c1-260 >> "$OUT/summary.txt"
  done
done
V="PGSD_PACK_ROWS_CFG=256x4,PGSD_PACK_ROWS_CFG=256x2,PGSD_PACK_ROWS_CFG=256x1,PGSD_PACK_ROWS_CFG=128x2,PGSD_PACK_ROWS_CFG=128x1,PGSD_PACK_ROWS_CFG=64x2,PGSD_PACK_ROWS_CFG=256x8,PGSD_PACK_ROWS_CFG=512x2"
for N in 524288 1048576 2097152 4194304; do
  python3 $CUR/tools/pack_bench.py --N $N --iters 1600 --warmup 80 --workloads hoomd_pvi,double4,pos_vel_id --kernel-only --variants "$V" \
      >> "$OUT/shapes.jsonl" 2>> "$OUT/shapes.err"
done
cat "$OUT/summary.txt"
python3 - "$OUT/shapes.jsonl" <<'PY'
import json, sys
for ln in open(sys.argv[1]):
    d = json.loads(ln)
    v = {k[len("varPGSD_PACK_ROWS_CFG="):]: x["median_us"] for k, x in d.items() if k.startswith("var")}
    print(d.get("workload"), d.get("N"), " ".join("%s=%.2f" % kv for kv in sorted(v.items(), key=lambda kv: kv[1])))
PY
