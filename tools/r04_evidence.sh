#!/bin/bash
# round 4: evidence behind profiles/r04_* (gpurun -- bash tools/r04_evidence.sh).  Each step bounded; outputs under
# gpurun_out/r04/evidence.
set -u
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04/evidence
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py"
# headline: plain run, then the same command under rocprofv3 --kernel-trace --stats
timeout -k 10 300 $B > $O/bench_n1.json 2> $O/bench_n1.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_10M -- $B --steps 10 --warmup 2 --no-cpu-baseline --traffic off --no-stall-test --no-exchange-probe > $O/bench_under_rocprof_10M.json 2> $O/stats_10M.err
# config 4: full schemas to /dev/shm and to the disk-backed root, live PMC traffic; kernel stats of the 15- and 20-chunk launches
for schema in pvi sph union; do
  for dir in /dev/shm /tmp; do
    timeout -k 10 400 $B --schema $schema --dir $dir --steps 6 --warmup 2 --no-cpu-baseline --no-exchange-probe $([ $dir = /tmp ] && echo --traffic off) > $O/config4_${schema}_$(basename $dir).json 2>> $O/config4.err
  done
  [ $schema = pvi ] || timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$schema -- $B --schema $schema --steps 6 --warmup 2 --no-cpu-baseline --traffic off --no-stall-test --no-exchange-probe > /dev/null 2> $O/stats_$schema.err
done
# elision: the comparison at 10 M particles and what the modes cost a 1 024-particle frame
timeout -k 10 300 python3 $R/tools/elision_bench.py --json $O/elision_10M.jsonl > $O/elision_10M.log 2>&1
timeout -k 10 300 python3 $R/tools/elision_bench.py --N 1024 --frames 600 --json $O/elision_1024.jsonl > $O/elision_1024.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_elision -- python3 $R/tools/elision_bench.py --no-append > /dev/null 2> $O/stats_elision.err
# overlap: a kernel queue beside a draining snapshot, with the kernel trace of the run
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_overlap -- python3 $R/tools/overlap_probe.py --frames 16 > $O/overlap_under_rocprof.jsonl 2> $O/stats_overlap.err
# full-size eight-rank runs (the tests' worker), for the record
for what in config3 config4; do
  PGSD_RCCL_LIBRARY=$R/tests/build/libpgsd_fake_rccl.so PGSD_FAKE_RCCL_SYNC=1 timeout -k 10 300 python3 $R/tests/fullsize_ranks_worker.py rccl 8 10000000 $what /dev/shm 2>/dev/null | grep RESULT >> $O/fullsize_eight_ranks.txt
  timeout -k 10 300 python3 $R/tests/fullsize_ranks_worker.py shm 8 10000000 $what /dev/shm 2>/dev/null | grep RESULT >> $O/fullsize_eight_ranks.txt
done
ls $O
