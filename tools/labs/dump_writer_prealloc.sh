# examples/dump_writer with and without pgsd_device_configure(prealloc_mib): snapshots every 100 steps of a 1 M-particle
# run, i.e. as fast as the file takes them (frames pile up; staging blocks and ring slabs are wanted during the run)
set -e
mkdir -p gpurun_out/r05b
timeout -k 10 900 python -m pytest tests/test_gpu_edges.py tests/test_gpu_elision.py tests/test_gpu_native_harness.py -x -q 2>&1 | tail -3
B=pgsd-sph_amd/csrc/build/dump_writer
rm -f gpurun_out/r05b/dump_writer_prealloc.jsonl
for pre in 0 2048 0 2048; do
  PGSD_RANK=0 PGSD_NRANKS=1 DUMP_WRITER_PREALLOC_MIB=$pre timeout -k 10 200 $B 1000000 4000 100 /dev/shm/dw.gsd all \
    | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); d['prealloc_mib']=$pre; print(json.dumps(d))" | tee -a gpurun_out/r05b/dump_writer_prealloc.jsonl
done
