set -e
mkdir -p gpurun_out/r05b
B=pgsd-sph_amd/csrc/build/dump_writer
PGSD_RANK=0 PGSD_NRANKS=1 DUMP_WRITER_PREALLOC_MIB=1024 timeout -k 10 400 $B 1000000 200000 500 /dev/shm/dw_soak.gsd fluid - hilbert | tee gpurun_out/r05b/dump_writer_soak.jsonl
SHM=dwsoak_$$
for r in 0 1 2; do
  PGSD_RANK=$r PGSD_NRANKS=3 PGSD_SHM_NAME=$SHM timeout -k 10 400 $B 700000 60000 300 /dev/shm/dw_soak3.gsd all - hilbert > gpurun_out/r05b/dw_soak3_$r.out 2> gpurun_out/r05b/dw_soak3_$r.err &
done
wait
cat gpurun_out/r05b/dw_soak3_0.out | tee -a gpurun_out/r05b/dump_writer_soak.jsonl
for r in 0 1 2; do tail -n 2 gpurun_out/r05b/dw_soak3_$r.err; done
