#!/bin/bash
# round 3, first GPU pass: suite, bench lines, the "before" picture of the small-frame latency
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_pass1
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench n1 rc=$?" | tee -a $O/summary.txt
timeout -k 10 300 python bench.py --gpus 2 --rehearse-shared-gpu --particles 5000000 --traffic off > $O/bench_n2_rehearse.json 2> $O/bench_n2.err; echo "bench n2 rc=$?" | tee -a $O/summary.txt
timeout -k 10 200 python pgsd-sph_amd/examples/benchmark_hoomd.py --size 256 > $O/benchmark_hoomd_before.log 2>&1
timeout -k 10 200 python pgsd-sph_amd/examples/benchmark_hoomd.py --size 256 --device >> $O/benchmark_hoomd_before.log 2>&1; echo "hoomd bench rc=$?" | tee -a $O/summary.txt
cd /tmp
for n in 1024 16384 1048576; do
  PGSD_TRACE=1 timeout -k 10 200 rocprofv3 --hip-trace --kernel-trace --marker-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_$n -- python3 $GRAFT_REPO_ROOT/tools/append_trace.py $n 100 > $GRAFT_REPO_ROOT/$O/trace_$n.log 2>&1
  echo "trace $n rc=$?" | tee -a $GRAFT_REPO_ROOT/$O/summary.txt
done
cd $GRAFT_REPO_ROOT
python tools/append_trace.py 1024 2000 host >> $O/summary.txt 2>&1
python tools/append_trace.py 1024 2000 >> $O/summary.txt 2>&1
tail -3 $O/pytest_gpu.log
cat $O/summary.txt
