#!/bin/bash
# Multi-rank rehearsals of bench.py on a ONE-GPU box (the ranks share cuda:0 over a gloo group): 2 and 4 ranks with
# torch.distributed callbacks, the same through the native RCCL back end's code over the tests' stand-in librccl
# (asynchronous on the stream, ranks as processes), and a declared partition.  Not a measurement of RCCL: what runs here
# is everything above the seven librccl calls.  Outputs: gpurun_out/<tag>/rehearse/*.json (REHEARSE_TAG, default r05).
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${REHEARSE_TAG:-r05}/rehearse
rm -rf $O; mkdir -p $O; cd $O
B="python3 $R/bench.py --particles 5000000 --steps 8 --warmup 2 --traffic off --rehearse-shared-gpu"
for n in 2 4; do
  timeout -k 10 240 $B --gpus $n > shared_gpu_$n.json 2> shared_gpu_$n.err
  PGSD_RCCL_LIBRARY=$R/tests/build/libpgsd_fake_rccl.so timeout -k 10 240 $B --gpus $n > rccl_glue_$n.json 2> rccl_glue_$n.err
done
timeout -k 10 240 $B --gpus 2 --declared-partition --no-cpu-baseline > declared_2.json 2> declared_2.err
python3 - <<'PY'
import glob, json
for f in sorted(glob.glob("*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "NO LINE", e)
        continue
    x = d["exchange_us"] or {}
    print(f, d["n_gpus"], d["value"], d["comm_backend"], d["collectives_per_frame"], x.get("mean"), x.get("min"),
          d["config"]["target_fstype"], (d.get("cpu_baseline") or {}).get("value"))
PY
tail -n 2 *.err | tail -n 14
