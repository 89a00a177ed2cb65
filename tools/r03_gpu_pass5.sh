#!/bin/bash
# round 3, fifth GPU pass: whole suite, unpack fill timing, gather lab under rocprofv3, benchmark_hoomd in four modes
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_pass5
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/summary.txt
tail -3 $O/pytest_gpu.log | tee -a $O/summary.txt
for m in "" mass fill; do python tools/unpack_bench.py 10000000 $m 2>/dev/null | tail -1 | tee -a $O/unpack_bench.jsonl; done
for m in "" mass fill; do PGSD_UNPACK_KERNEL=tiles python tools/unpack_bench.py 10000000 $m 2>/dev/null | tail -1 | sed 's/^/tiles /' | tee -a $O/unpack_bench.jsonl; done
mkdir -p tools/build && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/gather_lab.hip -o tools/build/gather_lab 2>/dev/null
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/gather_prof -- $GRAFT_REPO_ROOT/tools/build/gather_lab 10000000 20 > $GRAFT_REPO_ROOT/$O/gather_lab.jsonl 2> $GRAFT_REPO_ROOT/$O/gather_lab.err ); echo "gather rc=$?" | tee -a $O/summary.txt
cat $O/gather_lab.jsonl
for mode in host hbm hbm-via-host hbm-async; do
  timeout -k 10 300 python pgsd-sph_amd/examples/benchmark_hoomd.py --size 1024 --mode $mode 2>&1 | grep -v amdgpu.ids | tee -a $O/benchmark_hoomd.log
done
for mode in host hbm; do timeout -k 10 120 python tools/append_cprofile.py 1024 $mode > $O/cprofile_1024_$mode.log 2>&1; done
head -30 $O/cprofile_1024_hbm.log
