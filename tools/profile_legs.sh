#!/bin/bash
# Round evidence for bench.py's legs (run on the GPU box through gpurun): rocprofv3 --kernel-trace --stats over
# `bench_legs.py --pmc-child` (every leg's kernels, 12 launches each, a marker launch in front of each leg), and the
# per-leg averages derived from the trace (tools/legs_trace_summary.py).  Outputs under gpurun_out/prof_legs_<tag>/.
set -u
TAG=${1:-r05}
OUT=$PWD/gpurun_out/prof_legs_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
LEGS=$GRAFT_REPO_ROOT/bench_legs.py
SUMMARY=$GRAFT_REPO_ROOT/tools/legs_trace_summary.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python3 $LEGS --pmc-child "$OUT/child.json" --reps 12 > "$OUT/child.out" 2> "$OUT/stats.err" || echo "legs stats failed"
python3 $SUMMARY "$OUT" > "$OUT/legs_trace_summary.json" || echo "summary failed"
cat "$OUT/legs_trace_summary.json"
