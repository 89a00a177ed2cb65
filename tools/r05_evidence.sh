#!/bin/bash
# round 5: evidence behind profiles/r05_* on the FINAL tree (gpurun -- bash tools/r05_evidence.sh <part>).  Each step
# bounded; outputs under gpurun_out/r05/evidence (+ gpurun_out/prof_r05, prof_legs_r05 from the profile scripts).
#   part 1: bench line, headline kernel profile + counters, legs profile
#   part 2: rank rehearsals on the one GPU, elision bench, full-size eight-rank runs
#   part 3: wide GPU fuzz campaign
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05/evidence
mkdir -p $O
PART=${1:-1}
if [ "$PART" = 1 ]; then
  ( cd $R && timeout -k 10 300 python3 bench.py > $O/bench_n1_final.json 2> $O/bench_n1_final.err ); echo "bench rc=$?"
  ( cd $R && timeout -k 10 500 bash tools/profile_pack.sh r05 > $O/profile_pack.log 2>&1 ); echo "profile_pack rc=$?"
  ( cd $R && timeout -k 10 300 bash tools/profile_legs.sh r05 > $O/profile_legs.log 2>&1 ); echo "profile_legs rc=$?"
  # config 4 as a headline run of its own, with the reference's CPU path of the same schema beside it
  for schema in sph union; do
    ( cd $R && timeout -k 10 300 python3 bench.py --schema $schema --steps 6 --warmup 2 --traffic off --no-stall-test --no-exchange-probe > $O/bench_$schema.json 2> $O/bench_$schema.err ); echo "bench $schema rc=$?"
  done
elif [ "$PART" = 2 ]; then
  ( cd $R && timeout -k 10 600 bash tools/rehearse_ranks.sh > $O/rehearse.log 2>&1 ); echo "rehearse rc=$?"
  cd /tmp && export TMPDIR=/tmp
  timeout -k 10 300 python3 $R/tools/elision_bench.py --json $O/elision_10M.jsonl > $O/elision_10M.log 2>&1; echo "elision rc=$?"
  for what in config3 config4; do
    PGSD_RCCL_LIBRARY=$R/tests/build/libpgsd_fake_rccl.so PGSD_FAKE_RCCL_SYNC=1 timeout -k 10 300 python3 $R/tests/fullsize_ranks_worker.py rccl 8 10000000 $what /dev/shm 2>/dev/null | grep RESULT >> $O/fullsize_eight_ranks.txt
    timeout -k 10 300 python3 $R/tests/fullsize_ranks_worker.py shm 8 10000000 $what /dev/shm 2>/dev/null | grep RESULT >> $O/fullsize_eight_ranks.txt
  done
  cat $O/fullsize_eight_ranks.txt
  timeout -k 10 200 $R/tools/labs/build/copy_ceiling 40 > $O/copy_ceiling.jsonl 2> $O/copy_ceiling.err; echo "copy_ceiling rc=$?"
else
  ( cd $R && PGSD_FUZZ_SEEDS=400 timeout -k 10 1000 python3 -m pytest tests/test_gpu_fuzz.py tests/test_gpu_file_fuzz.py tests/test_gpu_elision_fuzz.py -q -x > $O/gpu_fuzz_campaign.log 2>&1 ); echo "fuzz rc=$?"
  tail -3 $O/gpu_fuzz_campaign.log
fi
ls $O
