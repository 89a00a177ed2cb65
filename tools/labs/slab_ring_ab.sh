#!/bin/bash
# How deep must the ring of pinned slabs be?  bench.py's headline (10 M particles, pos + vel + typeid, /dev/shm and the
# disk-backed root) with rings of 3 ... 16 slabs of 16 MiB, alternating, two rounds.
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05b; mkdir -p $O; : > $O/slab_ring_ab.jsonl
for round in 1 2; do
  for dir in /dev/shm /tmp; do
    for n in 16 3 4 6 8; do
      timeout -k 10 200 python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --slabs $n --dir $dir --no-legs --no-cpu-baseline --traffic off --no-stall-test --no-exchange-probe 2>/dev/null \
        | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'slabs': $n, 'dir': '$dir', 'round': $round, 'value_GBps': d['value'], 'ms_per_step': d['ms_per_step'], 'd2h_GBps': d['pipeline']['d2h_GBps'], 'write_GBps': d['pipeline']['write_GBps_per_writer']}))" | tee -a $O/slab_ring_ab.jsonl
    done
  done
done
