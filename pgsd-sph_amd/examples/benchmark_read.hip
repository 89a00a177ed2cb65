// benchmark_read.hip -- native (no Python, no torch) harness of the device read path.
//
// The counterpart of the reference's pgsd/scripts/benchmark-read.cc for this library: every rank opens the
// file read-only, takes an even share of the rows of every per-particle chunk (floor + one more for the first
// N mod P ranks, benchmark-read.cc:60-75) and reads its slab of every frame -- here not into a host vector but
// straight into HOOMD-style Scalar4 arrays in HBM:
//   particles/position (N x 3 f32) + particles/typeid (N x 1 u32)  ->  pos4 = (x, y, z, type id bits)
//   particles/velocity (N x 3 f32)                                  ->  vel4.xyz (w = mass is not in the file)
// pgsd_find_chunk is valid on every rank in this library (the reference returns the entry on rank 0 only and
// the benchmark broadcasts N, M and type by hand, benchmark-read.cc:92-99), the reads are independent preads of
// disjoint row slabs, and one fused unpack launch per frame restores the arrays.  With "verify" the harness checks
// the restored arrays against the closed-form values benchmark_write.hip writes (run that with "keep").
//
//   hipcc --offload-arch=gfx950 -O2 -I include benchmark_read.hip -L pgsd-sph_amd/pgsd -lpgsd_amd
//   PGSD_RANK=r PGSD_NRANKS=P PGSD_SHM_NAME=job ./benchmark_read [file] [verify]
#include "pgsd.h"

#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// the values benchmark_write.hip's fill_scalar4 produces for global row g
__global__ void count_mismatches(const float4* pos, const float4* vel, uint64_t n, uint64_t row0, unsigned long long* bad)
    {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    uint64_t g = row0 + i;
    float x = (float)(g % 1000) * 0.1f - 50.f;
    float4 p = pos[i], v = vel[i];
    bool ok = p.x == x && p.y == x * 0.5f && p.z == -x && __float_as_uint(p.w) == (uint32_t)(g % 5)
              && v.x == 0.001f * (float)(g % 77) && v.y == 1.f && v.z == -1.f;
    if (!ok)
        atomicAdd(bad, 1ull);
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
    const char* path = argc > 1 ? argv[1] : "/dev/shm/pgsd_benchmark_write.gsd";
    const bool verify = argc > 2 && strcmp(argv[2], "verify") == 0;
    CHECK(pgsd_comm_init_from_env());
    const int rank = pgsd_comm_rank(), P = pgsd_comm_size();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        {
        fprintf(stderr, "no GPU\n");
        return 1;
        }
    (void)hipSetDevice(rank % ndev);

    struct pgsd_handle h;
    CHECK(pgsd_open(&h, path, PGSD_OPEN_READONLY));
    const uint64_t frames = pgsd_get_nframes(&h);
    const struct pgsd_index_entry* e0 = pgsd_find_chunk(&h, 0, "particles/position");
    if (!e0 || e0->M != 3 || e0->type != PGSD_TYPE_FLOAT)
        {
        fprintf(stderr, "%s: no N x 3 float particles/position in frame 0\n", path);
        return 1;
        }
    // even row shares; the particle count may change from frame to frame, the arrays hold the largest share
    const uint64_t n_global0 = e0->N;
    uint64_t cap = n_global0 / (uint64_t)P + 1;
    float4 *pos, *vel;
    unsigned long long* bad;
    (void)hipMalloc((void**)&pos, cap * sizeof(float4));
    (void)hipMalloc((void**)&vel, cap * sizeof(float4));
    (void)hipMalloc((void**)&bad, sizeof(*bad));
    (void)hipMemset(bad, 0, sizeof(*bad));

    const struct pgsd_field_dst pos_xyz = {pos, NULL, PGSD_TYPE_FLOAT, 4, 0, 0};
    const struct pgsd_field_dst pos_w = {pos, NULL, PGSD_TYPE_FLOAT, 4, 3, 1}; // u32 id bits into the float slot
    const struct pgsd_field_dst vel_xyz = {vel, NULL, PGSD_TYPE_FLOAT, 4, 0, 0};

    uint64_t rows_read = 0, bytes_read = 0;
    CHECK(pgsd_comm_barrier());
    auto t0 = std::chrono::steady_clock::now();
    for (uint64_t f = 0; f < frames; f++)
        {
        const struct pgsd_index_entry* ep = pgsd_find_chunk(&h, f, "particles/position");
        const struct pgsd_index_entry* ev = pgsd_find_chunk(&h, f, "particles/velocity");
        const struct pgsd_index_entry* et = pgsd_find_chunk(&h, f, "particles/typeid");
        if (!ep)
            continue; // a frame without particle data
        const uint64_t n_global = ep->N;
        uint64_t n = n_global / (uint64_t)P, row0 = n * (uint64_t)rank;
        const uint64_t rem = n_global % (uint64_t)P;
        if ((uint64_t)rank < rem)
            {
            n += 1;
            row0 += (uint64_t)rank;
            }
        else
            row0 += rem;
        if (n > cap)
            {
            fprintf(stderr, "frame %llu holds more particles than frame 0\n", (unsigned long long)f);
            return 1;
            }
        // find_chunk hands out pointers into the handle's index: copy before the next lookup is not needed here
        // (the index of a read-only handle does not move), the three entries stay valid together
        CHECK(pgsd_read_chunk_device(&h, ep, n, row0, &pos_xyz));
        bytes_read += n * 12;
        if (et)
            {
            CHECK(pgsd_read_chunk_device(&h, et, n, row0, &pos_w));
            bytes_read += n * 4;
            }
        if (ev)
            {
            CHECK(pgsd_read_chunk_device(&h, ev, n, row0, &vel_xyz));
            bytes_read += n * 12;
            }
        CHECK(pgsd_device_wait_read(&h)); // one fused unpack launch for the frame
        rows_read += n;
        if (verify)
            hipLaunchKernelGGL(count_mismatches, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, pos, vel, n, row0, bad);
        }
    (void)hipDeviceSynchronize();
    CHECK(pgsd_comm_barrier());
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    unsigned long long h_bad = 0;
    (void)hipMemcpy(&h_bad, bad, sizeof(h_bad), hipMemcpyDeviceToHost);
    // totals over the ranks for the report
    uint64_t mine[3] = {rows_read, bytes_read, (uint64_t)h_bad};
    std::vector<uint64_t> all((size_t)P * 3);
    CHECK(pgsd_comm_allgather(mine, all.data(), sizeof(mine)));
    uint64_t rows = 0, bytes = 0, mismatches = 0;
    for (int r = 0; r < P; r++)
        {
        rows += all[(size_t)r * 3];
        bytes += all[(size_t)r * 3 + 1];
        mismatches += all[(size_t)r * 3 + 2];
        }
    CHECK(pgsd_close(&h));
    if (rank == 0)
        printf("{\"ranks\": %d, \"frames\": %llu, \"particles\": %llu, \"rows_read\": %llu, \"bytes_read\": %llu, "
               "\"seconds\": %.4f, \"MBps\": %.1f, \"verified\": %s, \"mismatches\": %llu}\n",
               P, (unsigned long long)frames, (unsigned long long)n_global0, (unsigned long long)rows,
               (unsigned long long)bytes, dt, (double)bytes / dt / 1e6, verify ? (mismatches == 0 ? "true" : "false") : "null",
               (unsigned long long)mismatches);
    pgsd_comm_finalize();
    (void)hipFree(pos);
    (void)hipFree(vel);
    (void)hipFree(bad);
    return mismatches == 0 ? 0 : 1;
    }
