#!/bin/bash
# round 3, seventh GPU pass: the Cython file layer on the GPU box -- suite, smoke, bench, small frames
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_pass7
mkdir -p $O
export TMPDIR=/tmp
ls -la pgsd-sph_amd/pgsd/*.so | tee $O/summary.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/summary.txt
timeout -k 10 400 python bench.py --traffic off > $O/bench_n1.json 2> $O/bench_n1.err; echo "bench rc=$?" | tee -a $O/summary.txt
for mode in host hbm hbm-via-host hbm-async; do
  timeout -k 10 300 python pgsd-sph_amd/examples/benchmark_hoomd.py --size 1024 --mode $mode 2>&1 | grep -v amdgpu.ids | tee -a $O/benchmark_hoomd.log
done
for mode in host hbm; do timeout -k 10 120 python tools/append_cprofile.py 1024 $mode > $O/cprofile_1024_$mode.log 2>&1; done
head -24 $O/cprofile_1024_hbm.log
