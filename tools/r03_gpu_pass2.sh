#!/bin/bash
# round 3, second GPU pass: suite with the direct small-frame path, after-picture, threshold sweep
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_pass2
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt
timeout -k 10 200 python pgsd-sph_amd/examples/benchmark_hoomd.py --size 256 > $O/benchmark_hoomd_after.log 2>&1
timeout -k 10 200 python pgsd-sph_amd/examples/benchmark_hoomd.py --size 256 --device >> $O/benchmark_hoomd_after.log 2>&1; echo "hoomd bench rc=$?" | tee -a $O/summary.txt
cat $O/benchmark_hoomd_after.log
for mode in host hbm; do
  timeout -k 10 120 python tools/append_cprofile.py 1024 $mode > $O/cprofile_1024_$mode.log 2>&1
done
for kib in 0 256 2048 8192 32768; do
  for n in 1024 16384 65536 262144 1048576; do
    echo -n "direct_max_kib=$kib " >> $O/direct_sweep.log
    PGSD_DIRECT_MAX_KIB=$kib timeout -k 10 120 python tools/append_trace.py $n 300 2>/dev/null | tail -1 >> $O/direct_sweep.log
  done
done
for n in 1024 16384 65536 262144 1048576; do
  echo -n "host " >> $O/direct_sweep.log
  timeout -k 10 120 python tools/append_trace.py $n 300 host 2>/dev/null | tail -1 >> $O/direct_sweep.log
done
cat $O/direct_sweep.log
cd /tmp
PGSD_TRACE=1 timeout -k 10 200 rocprofv3 --hip-trace --kernel-trace --marker-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_1024 -- python3 $GRAFT_REPO_ROOT/tools/append_trace.py 1024 100 > $GRAFT_REPO_ROOT/$O/trace_1024.log 2>&1
echo "trace rc=$?"
