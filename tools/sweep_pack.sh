#!/bin/bash
# Sweep the pack-kernel launch variants on one workload set; one JSON line per run.
# usage: tools/sweep_pack.sh "<workloads>" "<kernels>" "<variants>" "<blocks_per_cu>" "<tiles>"
W=${1:-pos_vel_id}; KS=${2:-tiles}; VS=${3:-0}; BS=${4:-4}; TS=${5:-1024}
for k in $KS; do for v in $VS; do for b in $BS; do for t in $TS; do
  PGSD_PACK_KERNEL=$k PGSD_PACK_VARIANT=$v PGSD_PACK_BLOCKS_PER_CU=$b PGSD_PACK_TILE=$t \
    timeout -k 10 120 python tools/pack_bench.py --workloads "$W" --iters 40 --warmup 6 2>/dev/null || exit 1
done; done; done; done
