#!/bin/bash
# BASELINE config 2 as SURVEY 8(d) defines it: 2^20 particles, pack kernel only, >= 200 launches after 20 warm-ups,
# buffer sets rotated past the 256 MiB Infinity Cache, rocprofv3 kernel durations + FETCH_SIZE / WRITE_SIZE.
# float4 sources (HOOMD layout) and the double4 conversion variant.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_${TAG}_config2
rm -rf "$OUT"; mkdir -p "$OUT"
PB=$GRAFT_REPO_ROOT/tools/pack_bench.py
cd /tmp && export TMPDIR=/tmp
ARGS="--N 1048576 --iters 200 --warmup 20 --workloads hoomd_pvi,double4"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $PB $ARGS > "$OUT/pack_bench_under_rocprof.jsonl" 2> "$OUT/stats.err" || echo "stats failed"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- python3 $PB $ARGS > /dev/null 2> "$OUT/fetch.err" || echo "fetch failed"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- python3 $PB $ARGS > /dev/null 2> "$OUT/write.err" || echo "write failed"
python3 $PB $ARGS --kernel-only > "$OUT/pack_bench_plain.jsonl" 2>/dev/null
grep "pgsd_amd::pack" $(find "$OUT/stats" -name "*kernel_stats.csv" | head -1) | cut -c1-200
for k in fetch write; do f=$(find "$OUT/$k" -name "*counter_collection.csv" | head -1); grep "pgsd_amd::pack" $f | awk -F, '{n[$9]++; s[$9]+=$(NF-2)} END {for (k in n) print k, n[k], s[k]/n[k]}' ; done
cat "$OUT/pack_bench_plain.jsonl" | cut -c1-220
