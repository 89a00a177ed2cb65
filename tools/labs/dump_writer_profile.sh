#!/bin/bash
# rocprofv3 --kernel-trace --stats over examples/dump_writer (10 M particles, Hilbert memory order, a snapshot every 500 steps):
# what runs on the GPU per snapshot -- the fused gather + pack, the comparison of the static chunks, the copies -- next to the
# simulation's own step kernel.  Output: gpurun_out/r05b/dump_writer_prof/ (kernel_stats.csv is the summary).
set -u
O=$GRAFT_REPO_ROOT/gpurun_out/r05b/dump_writer_prof; rm -rf "$O"; mkdir -p "$O"
B=$GRAFT_REPO_ROOT/pgsd-sph_amd/csrc/build/dump_writer
cd /tmp && export TMPDIR=/tmp
export PGSD_RANK=0 PGSD_NRANKS=1 DUMP_WRITER_PREALLOC_MIB=2048
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/all" -- $B 10000000 4000 500 /dev/shm/dw_prof.gsd all - hilbert > "$O/all.json" 2> "$O/all.err" || echo "profile (all) failed"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/fluid" -- $B 10000000 4000 500 /dev/shm/dw_prof.gsd fluid - hilbert > "$O/fluid.json" 2> "$O/fluid.err" || echo "profile (fluid) failed"
for w in all fluid; do f=$(find "$O/$w" -name "*kernel_stats.csv" | head -1); echo "== $w"; cut -c1-150 "$f" | head -12; done
