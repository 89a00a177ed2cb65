#!/bin/bash
# Launch shapes of pack_rows_kernel for launches of three or more source arrays below 2 M rows (the size rule gives
# them 256 x 2): pos + vel + id (SURVEY 8(d)'s config-2 layout), the SPH schema, double4 sources; 2^19 ... 1.5 M rows;
# interleaved launch by launch, dispatch-stamped.
O=$GRAFT_REPO_ROOT/gpurun_out/r05b; mkdir -p $O; : > $O/rows_shape_small_frames.jsonl
V=PGSD_PACK_ROWS_CFG=256x2,PGSD_PACK_ROWS_CFG=256x4,PGSD_PACK_ROWS_CFG=256x8
for w in pos_vel_id sph_full double4; do
  for n in 524288 1048576 1572864; do
    python3 $GRAFT_REPO_ROOT/tools/pack_bench.py --workloads $w --N $n --iters 300 --warmup 30 --kernel-only --variants $V >> $O/rows_shape_small_frames.jsonl
  done
done
python3 - $O/rows_shape_small_frames.jsonl <<'PY'
import sys, json
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d = json.loads(l)
        print(d["workload"], d["N"], {k.split('=')[1]: v["median_us"] for k, v in d.items() if k.startswith("var")})
PY
