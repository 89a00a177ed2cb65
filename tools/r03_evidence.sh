#!/bin/bash
# round 3: the evidence run behind profiles/r03_* (gpurun -- bash tools/r03_evidence.sh): kernel stats + PMC (10 M, 2^20), config 2, bench lines, rehearsals,
# the phase timeline of a 10 M frame, the unpack fill under the profiler
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03_evidence
mkdir -p $O
export TMPDIR=/tmp
bash tools/profile_pack.sh r03 > $O/profile_pack.log 2>&1; echo "profile_pack rc=$?" | tee -a $O/summary.txt
bash tools/profile_config2.sh r03 > $O/profile_config2.log 2>&1; echo "profile_config2 rc=$?" | tee -a $O/summary.txt
cd "$GRAFT_REPO_ROOT"
timeout -k 10 400 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench n1 rc=$?" | tee -a $O/summary.txt
timeout -k 10 400 python bench.py --gpus 2 --rehearse-shared-gpu --particles 5000000 --traffic off > $O/rehearse_2ranks_shared_gpu.json 2> $O/rehearse2.err; echo "rehearse2 rc=$?" | tee -a $O/summary.txt
timeout -k 10 400 python bench.py --gpus 4 --rehearse-shared-gpu --particles 2500000 --traffic off > $O/rehearse_4ranks_shared_gpu.json 2> $O/rehearse4.err; echo "rehearse4 rc=$?" | tee -a $O/summary.txt
PGSD_RCCL_LIBRARY=$GRAFT_REPO_ROOT/pgsd-sph_amd/csrc/build/libpgsd_fake_rccl.so timeout -k 10 400 python bench.py --gpus 2 --rehearse-shared-gpu --particles 5000000 --traffic off --no-cpu-baseline > $O/rehearse_2ranks_rccl_glue.json 2> $O/rehearse2g.err; echo "rehearse2 glue rc=$?" | tee -a $O/summary.txt
cd /tmp
PGSD_TRACE=1 timeout -k 10 300 rocprofv3 --marker-trace --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/timeline_10M -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic off --no-stall-test --no-exchange-probe > $O/timeline_bench.json 2> $O/timeline.err; echo "timeline rc=$?" | tee -a $O/summary.txt
python3 $GRAFT_REPO_ROOT/tools/phase_timeline.py $O/timeline_10M > $O/phase_timeline_10M.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/unpack_fill_stats -- python3 $GRAFT_REPO_ROOT/tools/unpack_bench.py 10000000 fill > $O/unpack_fill_under_rocprof.json 2> $O/unpack_fill.err
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/unpack_fill_fetch -- python3 $GRAFT_REPO_ROOT/tools/unpack_bench.py 10000000 fill > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/unpack_fill_write -- python3 $GRAFT_REPO_ROOT/tools/unpack_bench.py 10000000 fill > /dev/null 2>&1
for n in 1024 16384; do
  PGSD_TRACE=1 timeout -k 10 200 rocprofv3 --hip-trace --kernel-trace --marker-trace --stats --output-format csv -d $O/trace_$n -- python3 $GRAFT_REPO_ROOT/tools/append_trace.py $n 100 > $O/trace_$n.log 2>&1
  python3 $GRAFT_REPO_ROOT/tools/phase_timeline.py $O/trace_$n 60 > $O/phase_timeline_append_$n.txt 2>&1
done
# elision of GPU-resident arrays: the comparison kernel under the profiler (duration, FETCH_SIZE, WRITE_SIZE), then the tool's own run
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/elision_stats -- python3 $GRAFT_REPO_ROOT/tools/elision_bench.py --no-append > $O/elision_under_rocprof.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/elision_fetch -- python3 $GRAFT_REPO_ROOT/tools/elision_bench.py --no-append > /dev/null 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/elision_write -- python3 $GRAFT_REPO_ROOT/tools/elision_bench.py --no-append > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
timeout -k 10 300 python tools/elision_bench.py > $O/elision_bench.log 2>&1; echo "elision bench rc=$?" | tee -a $O/summary.txt
cat $O/summary.txt; tail -1 $O/bench_n1.json | cut -c1-600; tail -12 $O/phase_timeline_10M.txt
