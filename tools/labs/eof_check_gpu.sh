mkdir -p gpurun_out/r05b
PGSD_CHECK_EOF=1 timeout -k 10 900 python -m pytest tests/test_gpu_golden_device.py tests/test_gpu_edges.py tests/test_gpu_file.py tests/test_gpu_file_fuzz.py tests/test_gpu_native_harness.py tests/test_gpu_eight_ranks.py -x -q > gpurun_out/r05b/gpu_check_eof.log 2>&1; tail -3 gpurun_out/r05b/gpu_check_eof.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r05b/gpu_suite_8.log 2>&1; tail -3 gpurun_out/r05b/gpu_suite_8.log
B=pgsd-sph_amd/csrc/build/dump_writer
PGSD_RANK=0 PGSD_NRANKS=1 DUMP_WRITER_PREALLOC_MIB=2048 timeout -k 10 200 $B 1000000 4000 100 /dev/shm/dw.gsd all - hilbert | tee gpurun_out/r05b/dump_writer_noreloc_stall.jsonl
PGSD_RANK=0 PGSD_NRANKS=1 DUMP_WRITER_PREALLOC_MIB=1024 timeout -k 10 400 $B 1000000 200000 500 /dev/shm/dw_soak.gsd fluid - hilbert | tee -a gpurun_out/r05b/dump_writer_noreloc_stall.jsonl
