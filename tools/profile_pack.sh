#!/bin/bash
# Round evidence for the pack kernel (run on the GPU box through gpurun):
#   rocprofv3 --kernel-trace --stats of `bench.py` at 10 M particles (config 3 size) and at 2^20 (config 2),
#   separate --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of the same commands,
#   the bench line printed under the profiler, and an un-profiled bench line.
# Outputs under gpurun_out/prof_<tag>/; tools/collect_profiles.py copies the summaries into profiles/.
set -u
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH=$GRAFT_REPO_ROOT/bench.py
for cfg in "10M 10000000" "1M 1048576"; do
  set -- $cfg; name=$1; n=$2
  common="--particles $n --steps 10 --warmup 2 --no-cpu-baseline --traffic off --no-stall-test --no-exchange-probe --no-legs"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_$name" -- python3 $BENCH $common > "$OUT/bench_under_rocprof_$name.json" 2> "$OUT/stats_$name.err" || echo "stats $name failed"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch_$name" -- python3 $BENCH $common > /dev/null 2> "$OUT/fetch_$name.err" || echo "fetch $name failed"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write_$name" -- python3 $BENCH $common > /dev/null 2> "$OUT/write_$name.err" || echo "write $name failed"
  python3 $BENCH --particles $n --steps 20 --warmup 5 --no-cpu-baseline --traffic off --no-exchange-probe --no-legs > "$OUT/bench_plain_$name.json" 2> "$OUT/bench_plain_$name.err" || echo "plain $name failed"
done
find "$OUT" -name "*.csv" | head -40
