#!/bin/bash
# round 3, third GPU pass: new BASELINE-size tests, the warmer experiment, small-frame numbers after the Python trims
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03_pass3
mkdir -p $O
export TMPDIR=/tmp
{
  echo "nproc: $(nproc)"; echo "allowed: $(taskset -cp $$ 2>/dev/null)"
  lscpu | grep -E "Model name|Socket|Core|Thread|L3|NUMA" 
  echo "L3 of cpu0: $(cat /sys/devices/system/cpu/cpu0/cache/index3/shared_cpu_list)"
  echo "gpu numa: $(cat /sys/class/drm/card*/device/numa_node 2>/dev/null | tr '\n' ' ')"
} > $O/topology.txt 2>&1
timeout -k 10 900 python -m pytest tests/test_gpu_eight_ranks.py tests/test_gpu_config5.py tests/test_gpu_bench_contract.py "tests/test_gpu_pack.py" -x -q -k "eight or 80M or bench or gpus or exchange_probe or config2" > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_new.log | tee -a $O/summary.txt
B="python bench.py --no-cpu-baseline --traffic off --no-stall-test --no-exchange-probe --steps 20 --warmup 3"
run() { # label, env..., -- args
  label=$1; shift
  out=$(env "$@" 2>/dev/null | tail -1)
  echo "$label $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("value", d["value"], "ms", d["ms_per_step"], "pipe", d["pipeline"])' 2>/dev/null)" | tee -a $O/warm_sweep.log
}
run "base slab16" $B
run "base slab4" $B --slab-mib 4 --slabs 16
for w in 1 2 4; do for mib in 1 2 4 8 16; do
  run "warmers=$w slab=$mib" PGSD_WARMERS=$w $B --slab-mib $mib --slabs 16
done; done
for n in 1024 16384 262144 1048576; do
  echo -n "hbm " >> $O/append.log; python tools/append_trace.py $n 300 2>/dev/null | tail -1 >> $O/append.log
  echo -n "hbm warmers=2 " >> $O/append.log; PGSD_WARMERS=2 python tools/append_trace.py $n 300 2>/dev/null | tail -1 >> $O/append.log
  echo -n "host " >> $O/append.log; python tools/append_trace.py $n 300 host 2>/dev/null | tail -1 >> $O/append.log
done
cat $O/append.log
timeout -k 10 200 python pgsd-sph_amd/examples/benchmark_hoomd.py --size 256 > $O/benchmark_hoomd.log 2>&1
timeout -k 10 200 python pgsd-sph_amd/examples/benchmark_hoomd.py --size 256 --device >> $O/benchmark_hoomd.log 2>&1
cat $O/benchmark_hoomd.log
cat $O/topology.txt
