// benchmark_write.hip -- native (no Python, no torch) harness of the device write path.
//
// The counterpart of the reference's pgsd/scripts/benchmark-write.cc for this library: every
// rank (one process per GPU; ranks meet through the shm communicator or stay alone) holds
// HOOMD-style Scalar4 arrays in HBM, and per frame calls
//   pgsd_write_chunk     -> configuration/step (replicated small chunk; queued)
//   pgsd_write_chunks_device(..., PGSD_PARTITION_AUTO) -> position, velocity, typeid in one fused pack
//                           launch (queued: the kernel runs, the file offsets come with the exchange)
//   pgsd_end_frame       -> ONE allgather of the ranks' chunk sizes (it also yields the row partition, the
//                           MPI_Allgather of benchmark-write.cc:39-45), placement, copies, writes, index
// (pgsd_set_frame_exchange; pass "perchunk" as 4th argument for one exchange per chunk and the caller-side
// pgsd_partition_rows instead; "declared": the row counts exchanged ONCE before the first frame are declared with
// pgsd_set_partition and the frames then cost no collective at all; "elide": declared, plus the elision test a
// snapshot writer with static per-particle arrays would run -- every frame stages position, velocity, type id and a
// charge that changes with the step, compares the packed chunks with frame 0's rows kept in HBM
// (pgsd_copy_staged_chunks / pgsd_compare_staged_chunks), agrees the outcome over the ranks with one small
// allgather and writes only what differs: after frame 0 that is the charge alone) and prints MB/s the way the
// reference's benchmark does, as one JSON line on rank 0.
// With "rccl" as 5th argument the ranks bootstrap the library's RCCL communicator themselves -- rank 0's
// ncclUniqueId travels over the shm communicator they met on -- and every exchange of the run is an
// ncclAllGather over xGMI: the path a C++ caller (HOOMD-SPH's dump writer) takes without MPI or torch.
//
//   hipcc --offload-arch=gfx950 -O2 -I include benchmark_write.hip -L pgsd-sph_amd/pgsd -lpgsd_amd
// With "keep" as 6th argument the file is left behind (benchmark_read.hip reads it back and checks the values).
//
//   PGSD_RANK=r PGSD_NRANKS=P PGSD_SHM_NAME=job ./benchmark_write [particles_per_rank] [frames] [file] [batched|perchunk|declared|elide] [shm|rccl] [keep]
#include "pgsd.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <vector>

__global__ void fill_scalar4(float4* pos, float4* vel, uint64_t n, uint64_t row0)
    {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    uint64_t g = row0 + i;
    float x = (float)(g % 1000) * 0.1f - 50.f;
    pos[i] = make_float4(x, x * 0.5f, -x, __uint_as_float((uint32_t)(g % 5))); // w = type id bits
    vel[i] = make_float4(0.001f * (float)(g % 77), 1.f, -1.f, 2.5f);             // w = mass
    }

__global__ void fill_charge(float* charge, uint64_t n, uint64_t row0, uint64_t step)
    {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        charge[i] = (float)((row0 + i) % 13) + 100.f * (float)step;
    }

#define CHECK(x)                                                                     \
    do                                                                               \
        {                                                                            \
        int rc_ = (x);                                                               \
        if (rc_ != 0)                                                                \
            {                                                                        \
            fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, pgsd_last_error_string()); \
            return 1;                                                                \
            }                                                                        \
        } while (0)

int main(int argc, char** argv)
    {
    const uint64_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 10000000ull;
    const int frames = argc > 2 ? atoi(argv[2]) : 10;
    const char* path = argc > 3 ? argv[3] : "/dev/shm/pgsd_benchmark_write.gsd";
    const bool elide = argc > 4 && strcmp(argv[4], "elide") == 0;
    const bool declared = elide || (argc > 4 && strcmp(argv[4], "declared") == 0);
    const bool batched = !declared && !(argc > 4 && strcmp(argv[4], "perchunk") == 0);
    const bool keep = argc > 6 && strcmp(argv[6], "keep") == 0;
    CHECK(pgsd_comm_init_from_env());
    const int rank = pgsd_comm_rank(), P = pgsd_comm_size();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        {
        fprintf(stderr, "no GPU\n");
        return 1;
        }
    (void)hipSetDevice(rank % ndev);
    const char* comm_name = P > 1 ? "shm" : "self";
    if (argc > 5 && strcmp(argv[5], "rccl") == 0)
        {
        // 128-byte id from rank 0 to everybody over the communicator of the launch, then switch.  The same exchange
        // carries every rank's "librccl loaded, device there" (byte 128): nobody enters ncclCommInitRank -- which
        // waits for all ranks -- unless every rank can
        unsigned char id[129] = {0};
        int ready = pgsd_comm_rccl_available(rank % ndev);
        if (ready == PGSD_SUCCESS && rank == 0)
            ready = pgsd_comm_rccl_unique_id(id);
        id[128] = ready == PGSD_SUCCESS;
        std::vector<unsigned char> all((size_t)P * sizeof(id));
        CHECK(pgsd_comm_allgather(id, all.data(), sizeof(id)));
        for (int r = 0; r < P; r++)
            if (!all[(size_t)r * sizeof(id) + 128])
                {
                fprintf(stderr, "rank %d: the RCCL back end is not available on rank %d%s%s\n", rank, r,
                        r == rank ? ": " : "", r == rank ? pgsd_last_error_string() : "");
                return 1;
                }
        // make the RCCL communicator and install it as the process default (the default takes it over)
        struct pgsd_comm rccl;
        CHECK(pgsd_comm_create_rccl(all.data(), rank, P, rank % ndev, &rccl));
        CHECK(pgsd_comm_set_default(&rccl));
        comm_name = "rccl";
        }

    uint64_t row0, n_global;
    std::vector<uint64_t> counts((size_t)P);
    CHECK(pgsd_partition_rows(n, &row0, &n_global, counts.data()));
    float4 *pos, *vel;
    (void)hipMalloc((void**)&pos, n * sizeof(float4));
    (void)hipMalloc((void**)&vel, n * sizeof(float4));
    hipLaunchKernelGGL(fill_scalar4, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, pos, vel, n, row0);
    (void)hipDeviceSynchronize();

    struct pgsd_handle h;
    CHECK(pgsd_create_and_open(&h, path, "benchmark_write", "hoomd", pgsd_make_version(1, 4), PGSD_OPEN_READWRITE, 0));
    float* charge = NULL;
    if (elide)
        (void)hipMalloc((void**)&charge, (n ? n : 1) * sizeof(float));
    struct pgsd_chunk_req req[4];
    memset(req, 0, sizeof(req));
    req[3].name = "particles/charge";
    req[3].type = PGSD_TYPE_FLOAT;
    req[3].M = 1;
    req[3].src = {charge, NULL, PGSD_TYPE_FLOAT, 1, 0, 0};
    void* frame0_rows[4] = {NULL, NULL, NULL, NULL}; // elide: frame 0's packed rows of this rank, kept in HBM
    const size_t packed_bytes[4] = {(size_t)n * 12, (size_t)n * 12, (size_t)n * 4, (size_t)n * 4};
    unsigned long long chunks_written = 0, chunks_elided = 0;
    req[0].name = "particles/position";
    req[0].type = PGSD_TYPE_FLOAT;
    req[0].M = 3;
    req[0].src = {pos, NULL, PGSD_TYPE_FLOAT, 4, 0, 0};
    req[1].name = "particles/velocity";
    req[1].type = PGSD_TYPE_FLOAT;
    req[1].M = 3;
    req[1].src = {vel, NULL, PGSD_TYPE_FLOAT, 4, 0, 0};
    req[2].name = "particles/typeid";
    req[2].type = PGSD_TYPE_UINT32;
    req[2].M = 1;
    req[2].src = {pos, NULL, PGSD_TYPE_FLOAT, 4, 3, 1};

    CHECK(pgsd_set_frame_exchange(&h, batched ? 1 : 0));
    if (declared) // the counts pgsd_partition_rows brought above, once: no particle migrates in this harness
        CHECK(pgsd_set_partition(&h, counts.data(), (uint32_t)P));
    auto frame = [&](uint64_t step) -> int
    {
        if (!batched && !declared)
            CHECK(pgsd_partition_rows(n, &row0, &n_global, NULL));
        CHECK(pgsd_write_chunk(&h, "configuration/step", PGSD_TYPE_UINT64, 1, 1, 1, 1, 0, 1, false, 0, &step));
        if (elide)
            {
            hipLaunchKernelGGL(fill_charge, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, charge, n, row0, step);
            uint64_t ticket = 0;
            uint8_t same[4] = {0, 0, 0, 0};
            CHECK(pgsd_stage_chunks_device(&h, 4, req, n, &ticket)); // ONE fused pack launch for the four arrays
            if (step == 0)
                {
                for (int i = 0; i < 4; i++)
                    (void)hipMalloc(&frame0_rows[i], packed_bytes[i] ? packed_bytes[i] : 4);
                CHECK(pgsd_copy_staged_chunks(&h, ticket, 0, 4, frame0_rows));
                }
            else
                CHECK(pgsd_compare_staged_chunks(&h, ticket, 0, 4, frame0_rows, NULL, same));
            // a chunk is skipped only when EVERY rank found its rows unchanged
            std::vector<uint8_t> votes((size_t)P * 4);
            CHECK(pgsd_comm_allgather(same, votes.data(), 4));
            for (uint32_t i = 0; i < 4; i++)
                {
                bool skip = true;
                for (int r = 0; r < P; r++)
                    skip = skip && votes[(size_t)r * 4 + i] != 0;
                if (skip)
                    chunks_elided++;
                else
                    {
                    CHECK(pgsd_write_staged_chunks(&h, ticket, i, 1, PGSD_PARTITION_AUTO, 0));
                    chunks_written++;
                    }
                }
            }
        else if (batched || declared)
            CHECK(pgsd_write_chunks_device(&h, 3, req, n, PGSD_PARTITION_AUTO, 0));
        else
            CHECK(pgsd_write_chunks_device(&h, 3, req, n, n_global, row0));
        CHECK(pgsd_end_frame(&h));
        return 0;
    };
    if (frame(0))
        return 1;
    CHECK(pgsd_comm_barrier());
    auto t0 = std::chrono::steady_clock::now();
    for (int f = 0; f < frames; f++)
        if (frame((uint64_t)f + 1))
            return 1;
    CHECK(pgsd_comm_barrier());
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    struct pgsd_device_stats st;
    pgsd_device_get_stats(&h, &st, 0);
    struct pgsd_exchange_stats xs;
    pgsd_get_exchange_stats(&h, &xs, 0);
    const unsigned long long collectives = (unsigned long long)xs.collectives;
    CHECK(pgsd_close(&h));
    if (rank == 0)
        {
        printf("{\"ranks\": %d, \"particles_per_rank\": %llu, \"frames\": %d, \"seconds\": %.4f, \"MBps\": %.1f, "
               "\"pack_launches\": %llu, \"written_bytes_rank0\": %llu, \"exchange\": \"%s\", "
               "\"collectives_rank0\": %llu, \"comm\": \"%s\", \"chunks_written\": %llu, \"chunks_elided\": %llu}\n",
               P, (unsigned long long)n, frames, dt, (double)frames * (double)n_global * 28.0 / dt / 1e6,
               (unsigned long long)st.pack_launches, (unsigned long long)st.written_bytes,
               elide ? "none for the chunks (declared partition); one vote allgather per frame"
               : declared ? "none (declared partition)" : batched ? "one per frame" : "one per chunk",
               collectives, comm_name, chunks_written, chunks_elided);
        if (!keep)
            unlink(path);
        }
    pgsd_comm_finalize();
    (void)hipFree(pos);
    (void)hipFree(vel);
    (void)hipFree(charge);
    for (int i = 0; i < 4; i++)
        (void)hipFree(frame0_rows[i]);
    return 0;
    }
