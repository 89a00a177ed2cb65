# examples/dump_writer on the GPU box: the tests, then snapshots of 10 M particles at a realistic period (500 steps = 50 ms:
# the file keeps up) in both memory orders, all particles and a group; 1 M particles; staging and ring allocated up front
set -e
mkdir -p gpurun_out/r05b
timeout -k 10 600 python -m pytest tests/test_gpu_native_harness.py -x -q -k dump_writer > gpurun_out/r05b/dump_tests.log 2>&1 || { tail -40 gpurun_out/r05b/dump_tests.log; exit 1; }
tail -3 gpurun_out/r05b/dump_tests.log
B=pgsd-sph_amd/csrc/build/dump_writer
rm -f gpurun_out/r05b/dump_writer.jsonl
for args in "10000000 4000 500 /dev/shm/dw.gsd all - hilbert" "10000000 4000 500 /dev/shm/dw.gsd all - random" "10000000 4000 500 /dev/shm/dw.gsd fluid - hilbert" "10000000 4000 500 /dev/shm/dw.gsd fluid - random" "1000000 4000 500 /dev/shm/dw.gsd all - hilbert"; do
  PGSD_RANK=0 PGSD_NRANKS=1 DUMP_WRITER_PREALLOC_MIB=2048 timeout -k 10 300 $B $args | tee -a gpurun_out/r05b/dump_writer.jsonl
done
