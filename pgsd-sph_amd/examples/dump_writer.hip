// dump_writer.hip -- what a GPU simulation's snapshot writer looks like on top of libpgsd_amd.so (C ABI only).
//
// A toy "simulation" advances HOOMD-style Scalar4 arrays in HBM on a stream of its own, one kernel per step.  Its
// particles sit in MEMORY order (along a Hilbert curve through their lattice sites: the order a space-filling-curve
// sorter leaves them in; or, "random", in an adversarial uniform-like order), the file wants them in TAG
// order, and optionally only a group of them (everything that is not a wall particle).  Every `period` steps the
// writer takes a snapshot the way HOOMD-SPH's dump writer would, without leaving the GPU:
//
//   group filter      flags in tag order -> pgsd_select_rows (stream compaction: wave ballots + scans) -> the kept tags;
//                     composed with the reverse-tag array into the gather index of the frame
//   pack              pgsd_stage_chunks_device: ONE fused launch gathers position, type id (the bits of position.w),
//                     velocity, mass (velocity.w) and density through that index into packed GSD chunks, behind the
//                     simulation's stream (pgsd_device_set_source_stream)
//   elision           type id and mass never change: after frame 0 their packed rows are compared in HBM with frame 0's
//                     (pgsd_copy_staged_chunks / pgsd_compare_staged_chunks) and not written (hoomd.py:654-694)
//   placement         ONE allgather per frame (pgsd_handle_allgather) carries every rank's row count and its elision
//                     votes; the counts are declared (pgsd_set_partition), so placing the chunks exchanges nothing more
//   seal              pgsd_end_frame_async + pgsd_device_wait_packed: the simulation goes on as soon as the pack
//                     kernel is through; device->host copies and file writes run behind the next steps
//
// Afterwards the file is opened again through the reference's own entry points (pgsd_open / pgsd_find_chunk /
// pgsd_read_chunk) and every frame is compared with a host model of the simulation, bit for bit.  Rank 0 prints one
// JSON line: what a snapshot cost the simulation (`sim_gap_us`: the idle gap on the simulation's stream between the last
// step before a snapshot and the first one after) next to a step's time.
//
//   hipcc --offload-arch=gfx950 -O2 -I include dump_writer.hip -L pgsd-sph_amd/pgsd -lpgsd_amd
//   PGSD_RANK=r PGSD_NRANKS=P PGSD_SHM_NAME=job ./dump_writer [particles_per_rank] [steps] [period] [file] [all|fluid] [keep|-] [hilbert|random]
//   DUMP_WRITER_PREALLOC_MIB=n: pgsd_device_configure(prealloc_mib = n) -- no allocation meets a snapshot;
//   DUMP_WRITER_TIMING=1: where the host spends each snapshot call (stderr)
#include "pgsd.h"

#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <vector>

#define DT 0.005f
#define WALL_TYPE 2u

// ---------------------------------------------------------------- the model (device and host run the same arithmetic)
__host__ __device__ inline uint32_t type_of(uint64_t g)
    {
    return (uint32_t)((g * 2654435761ull >> 7) % 3);
    }
__host__ __device__ inline void initial_row(uint64_t g, float* p, float* v)
    {
    const float x = (float)(g % 4093) * 0.03125f - 60.f;
    p[0] = x;
    p[1] = x * 0.5f + (float)(g % 17);
    p[2] = -x;
    v[0] = 0.25f * (float)(g % 77) - 9.f;
    v[1] = 1.f;
    v[2] = -0.125f * (float)(g % 5);
    v[3] = 1.f + 0.5f * (float)(g % 3); // mass
    }
__host__ __device__ inline float density_at(uint64_t g, uint64_t step)
    {
    return 1000.f + 0.5f * (float)step + (float)(g % 7);
    }

// ---- memory order.  "random": slot s holds local tag (s * a + b) % n (a bijection because gcd(a, n) == 1) -- every
// gathered row a DRAM sector of its own, the adversarial case.  "hilbert": particles are created in lattice order (tag t at
// site (t % m, t / m % m, t / m^2), m = ceil(cbrt n)) and kept in memory along the 3-D Hilbert curve through their sites
// (Skilling's transform, AIP Conf. Proc. 707, 2004) -- what a space-filling-curve sorter leaves a dump writer with.
__global__ void order_random_kernel(uint32_t* tag, uint64_t n, uint64_t a, uint64_t b)
    {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n)
        tag[s] = (uint32_t)((s * a + b) % n);
    }

__global__ void hilbert_key_kernel(uint64_t* key, uint32_t* tags, uint64_t n, uint32_t m, int bits)
    {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n)
        return;
    uint32_t X[3] = {(uint32_t)(t % m), (uint32_t)(t / m % m), (uint32_t)(t / ((uint64_t)m * m))};
    const uint32_t M = 1u << (bits - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) // inverse undo
        {
        const uint32_t P = Q - 1;
        for (int i = 0; i < 3; i++)
            if (X[i] & Q)
                X[0] ^= P;
            else
                {
                const uint32_t tt = (X[0] ^ X[i]) & P;
                X[0] ^= tt;
                X[i] ^= tt;
                }
        }
    X[1] ^= X[0]; // Gray encode
    X[2] ^= X[1];
    uint32_t tt = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (X[2] & Q)
            tt ^= Q - 1;
    for (int i = 0; i < 3; i++)
        X[i] ^= tt;
    uint64_t k = 0;
    for (int bit = bits - 1; bit >= 0; bit--)
        for (int i = 0; i < 3; i++)
            k = (k << 1) | ((X[i] >> bit) & 1u);
    key[t] = k;
    tags[t] = (uint32_t)t;
    }

// slot s holds local tag tag[s]: the particle's state, and the reverse tag
__global__ void init_kernel(float4* pos, float4* vel, float* density, const uint32_t* tag, uint32_t* rtag, uint64_t n, uint64_t row0)
    {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n)
        return;
    const uint64_t t = tag[s], g = row0 + t;
    float p[3], v[4];
    initial_row(g, p, v);
    pos[s] = make_float4(p[0], p[1], p[2], __uint_as_float(type_of(g)));
    vel[s] = make_float4(v[0], v[1], v[2], v[3]);
    density[s] = density_at(g, 0);
    rtag[t] = (uint32_t)s;
    }

__global__ void step_kernel(float4* pos, const float4* vel, float* density, const uint32_t* tag, uint64_t n, uint64_t row0,
                            uint64_t step)
    {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n)
        return;
    float4 p = pos[s];
    const float4 v = vel[s];
    if (__float_as_uint(p.w) != WALL_TYPE) // walls stand still
        {
        p.x = __fmaf_rn(v.x, DT, p.x);
        p.y = __fmaf_rn(v.y, DT, p.y);
        p.z = __fmaf_rn(v.z, DT, p.z);
        pos[s] = p;
        }
    density[s] = density_at(row0 + tag[s], step);
    }

__global__ void group_flags_kernel(const float4* pos, const uint32_t* rtag, uint8_t* flags, uint64_t n)
    {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n)
        flags[t] = __float_as_uint(pos[rtag[t]].w) != WALL_TYPE ? 1 : 0;
    }

__global__ void compose_kernel(const uint32_t* kept_tags, const uint32_t* rtag, uint32_t* order, uint64_t n_kept)
    {
    const uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_kept)
        order[k] = rtag[kept_tags[k]];
    }

struct Vote // what a rank tells the others about its part of a frame
    {
    uint64_t kept;   // rows it writes
    uint8_t same[8]; // per chunk: packed rows equal to frame 0's
    };

#define CHECK(x)                                                                        \
    do                                                                                  \
        {                                                                               \
        int rc_ = (x);                                                                  \
        if (rc_ != 0)                                                                   \
            {                                                                           \
            fprintf(stderr, "%s failed: %d (%s)\n", #x, rc_, pgsd_last_error_string()); \
            return 1;                                                                   \
            }                                                                           \
        } while (0)
#define HIP(x)                                                                 \
    do                                                                         \
        {                                                                      \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess)                                                  \
            {                                                                  \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));     \
            return 1;                                                          \
            }                                                                  \
        } while (0)

static uint64_t gcd64(uint64_t a, uint64_t b)
    {
    while (b)
        {
        const uint64_t t = a % b;
        a = b;
        b = t;
        }
    return a;
    }

static unsigned blocks_for(uint64_t n)
    {
    return (unsigned)std::max<uint64_t>(1, (n + 255) / 256);
    }

int main(int argc, char** argv)
    {
    const uint64_t n = argc > 1 ? strtoull(argv[1], NULL, 10) : 1000000ull;
    const int steps = argc > 2 ? atoi(argv[2]) : 200;
    const int period = argc > 3 ? std::max(1, atoi(argv[3])) : 20;
    const char* path = argc > 4 ? argv[4] : "/dev/shm/pgsd_dump_writer.gsd";
    const bool fluid_only = argc > 5 && strcmp(argv[5], "fluid") == 0;
    const bool keep = argc > 6 && strcmp(argv[6], "keep") == 0;
    const bool hilbert = !(argc > 7 && strcmp(argv[7], "random") == 0);
    if (n == 0 || n >= (1ull << 31))
        {
        fprintf(stderr, "particles_per_rank must be in [1, 2^31)\n");
        return 1;
        }
    CHECK(pgsd_comm_init_from_env());
    const int rank = pgsd_comm_rank(), P = pgsd_comm_size();
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        {
        fprintf(stderr, "no GPU\n");
        return 1;
        }
    HIP(hipSetDevice(rank % ndev));
    uint64_t row0 = 0, n_all = 0;
    CHECK(pgsd_partition_rows(n, &row0, &n_all, NULL)); // the tags this rank owns: [row0, row0 + n)

    // ---- the simulation's state
    hipStream_t sim;
    HIP(hipStreamCreateWithFlags(&sim, hipStreamNonBlocking));
    float4 *pos, *vel;
    float* density;
    uint32_t *tag, *rtag, *kept_tags, *order;
    uint8_t* flags;
    HIP(hipMalloc((void**)&pos, n * sizeof(float4)));
    HIP(hipMalloc((void**)&vel, n * sizeof(float4)));
    HIP(hipMalloc((void**)&density, n * sizeof(float)));
    HIP(hipMalloc((void**)&tag, n * sizeof(uint32_t)));
    HIP(hipMalloc((void**)&rtag, n * sizeof(uint32_t)));
    HIP(hipMalloc((void**)&kept_tags, n * sizeof(uint32_t)));
    HIP(hipMalloc((void**)&order, n * sizeof(uint32_t)));
    HIP(hipMalloc((void**)&flags, n));
    if (hilbert)
        {
        uint32_t m = 1;
        while ((uint64_t)m * m * m < n)
            m++;
        int bits = 1;
        while ((1u << bits) < m)
            bits++;
        uint64_t *key, *key_sorted;
        uint32_t* tag_in;
        HIP(hipMalloc((void**)&key, n * sizeof(uint64_t)));
        HIP(hipMalloc((void**)&key_sorted, n * sizeof(uint64_t)));
        HIP(hipMalloc((void**)&tag_in, n * sizeof(uint32_t)));
        hipLaunchKernelGGL(hilbert_key_kernel, dim3(blocks_for(n)), dim3(256), 0, sim, key, tag_in, n, m, bits);
        void* temp = NULL;
        size_t temp_bytes = 0; // memory row i holds the tag with the i-th smallest key
        HIP(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, key, key_sorted, tag_in, tag, (int)n, 0, 3 * bits, sim));
        HIP(hipMalloc(&temp, temp_bytes ? temp_bytes : 16));
        HIP(hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, key, key_sorted, tag_in, tag, (int)n, 0, 3 * bits, sim));
        HIP(hipStreamSynchronize(sim));
        (void)hipFree(temp);
        (void)hipFree(key);
        (void)hipFree(key_sorted);
        (void)hipFree(tag_in);
        }
    else
        {
        uint64_t a = 2654435761ull % n;
        while (a < 2 ? n > 2 : gcd64(a, n) != 1)
            a = a < 2 ? 2 : a + 1;
        if (n <= 2)
            a = 1;
        hipLaunchKernelGGL(order_random_kernel, dim3(blocks_for(n)), dim3(256), 0, sim, tag, n, a, n / 3);
        }
    hipLaunchKernelGGL(init_kernel, dim3(blocks_for(n)), dim3(256), 0, sim, pos, vel, density, tag, rtag, n, row0);
    HIP(hipGetLastError());

    // ---- the file
    struct pgsd_handle h;
    CHECK(pgsd_create_and_open(&h, path, "dump_writer", "hoomd", pgsd_make_version(1, 4), PGSD_OPEN_READWRITE, 0));
    if (const char* mib = getenv("DUMP_WRITER_PREALLOC_MIB")) // staging and the whole pinned ring now, nothing during the run
        {
        struct pgsd_device_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.device = rank % ndev;
        cfg.prealloc_mib = (uint32_t)atoi(mib);
        CHECK(pgsd_device_configure(&h, &cfg));
        }
    CHECK(pgsd_set_frame_exchange(&h, 1));          // one allgather per frame places every chunk
    CHECK(pgsd_device_set_source_stream(&h, sim));  // the pack waits for the simulation's kernels, not the host
    enum { POSITION, TYPEID, VELOCITY, MASS, DENSITY, N_CHUNKS };
    struct pgsd_chunk_req req[N_CHUNKS];
    memset(req, 0, sizeof(req));
    req[POSITION] = {"particles/position", PGSD_TYPE_FLOAT, 3, {pos, order, PGSD_TYPE_FLOAT, 4, 0, 0}};
    req[TYPEID] = {"particles/typeid", PGSD_TYPE_UINT32, 1, {pos, order, PGSD_TYPE_FLOAT, 4, 3, 1}};
    req[VELOCITY] = {"particles/velocity", PGSD_TYPE_FLOAT, 3, {vel, order, PGSD_TYPE_FLOAT, 4, 0, 0}};
    req[MASS] = {"particles/mass", PGSD_TYPE_FLOAT, 1, {vel, order, PGSD_TYPE_FLOAT, 4, 3, 0}};
    req[DENSITY] = {"particles/density", PGSD_TYPE_FLOAT, 1, {density, order, PGSD_TYPE_FLOAT, 1, 0, 0}};
    const bool is_static[N_CHUNKS] = {false, true, false, true, false};
    void* frame0_rows[N_CHUNKS] = {NULL, NULL, NULL, NULL, NULL};
    uint64_t frame0_kept = 0;

    std::vector<double> stall_us;
    std::vector<uint64_t> frame_step;
    std::vector<uint64_t> frame_kept; // this rank's rows in every frame
    unsigned long long chunks_written = 0, chunks_elided = 0;

    // DUMP_WRITER_TIMING=1: where the host spends a snapshot call, per frame, on stderr
    const bool lap_on = getenv("DUMP_WRITER_TIMING") != NULL;
    auto snapshot = [&](uint64_t step) -> int
    {
        const auto t0 = std::chrono::steady_clock::now();
        auto lap_t = t0;
        char laps[400] = "";
        size_t lap_at = 0;
        auto lap = [&](const char* what)
        {
            if (!lap_on)
                return;
            const auto t = std::chrono::steady_clock::now();
            lap_at += (size_t)snprintf(laps + lap_at, sizeof(laps) - lap_at, " %s %.0f", what,
                                       std::chrono::duration<double, std::micro>(t - lap_t).count());
            lap_t = t;
        };
        // which rows, in which order: the group's tags ascending, each looked up in the reverse-tag array
        uint64_t kept = n;
        const uint32_t* gather = rtag;
        if (fluid_only)
            {
            hipLaunchKernelGGL(group_flags_kernel, dim3(blocks_for(n)), dim3(256), 0, sim, pos, rtag, flags, n);
            CHECK(pgsd_select_rows(flags, n, kept_tags, &kept, sim));
            hipLaunchKernelGGL(compose_kernel, dim3(blocks_for(kept)), dim3(256), 0, sim, kept_tags, rtag, order, kept);
            gather = order;
            }
        for (int i = 0; i < N_CHUNKS; i++)
            req[i].src.order = gather;
        lap("select");
        const uint32_t frame = (uint32_t)frame_step.size();
        uint64_t ticket = 0;
        CHECK(pgsd_stage_chunks_device(&h, N_CHUNKS, req, kept, &ticket)); // ONE fused gather + pack launch
        lap("stage");
        Vote mine;
        memset(&mine, 0, sizeof(mine));
        mine.kept = kept;
        if (frame == 0)
            {
            for (int i = 0; i < N_CHUNKS; i++)
                if (is_static[i])
                    HIP(hipMalloc(&frame0_rows[i], std::max<uint64_t>(kept, 1) * req[i].M * 4));
            CHECK(pgsd_copy_staged_chunks(&h, ticket, 0, N_CHUNKS, frame0_rows));
            frame0_kept = kept;
            }
        else if (kept == frame0_kept)
            CHECK(pgsd_compare_staged_chunks(&h, ticket, 0, N_CHUNKS, frame0_rows, NULL, mine.same));
        lap("compare");
        // THE collective of the frame: every rank's row count and its "unchanged since frame 0" votes travel together;
        // the counts are then DECLARED (pgsd_set_partition), so placing the chunks needs no exchange of its own
        std::vector<Vote> votes((size_t)P);
        CHECK(pgsd_handle_allgather(&h, &mine, votes.data(), sizeof(Vote)));
        std::vector<uint64_t> rows((size_t)P);
        uint64_t n_global = 0;
        for (int r = 0; r < P; r++)
            n_global += rows[(size_t)r] = votes[(size_t)r].kept;
        CHECK(pgsd_set_partition(&h, rows.data(), (uint32_t)P));
        CHECK(pgsd_write_chunk(&h, "configuration/step", PGSD_TYPE_UINT64, 1, 1, 1, 1, 0, 1, false, 0, &step));
        const uint32_t n32 = (uint32_t)n_global;
        CHECK(pgsd_write_chunk(&h, "particles/N", PGSD_TYPE_UINT32, 1, 1, 1, 1, 0, 1, false, 0, &n32));
        for (uint32_t i = 0; i < N_CHUNKS; i++)
            {
            bool skip = frame > 0; // a chunk stays out of the frame only when EVERY rank found its rows unchanged
            for (int r = 0; r < P; r++)
                skip = skip && votes[(size_t)r].same[i] != 0;
            if (skip)
                chunks_elided++;
            else
                {
                CHECK(pgsd_write_staged_chunks(&h, ticket, i, 1, PGSD_PARTITION_AUTO, 0));
                chunks_written++;
                }
            }
        lap("place");
        CHECK(pgsd_end_frame_async(&h));    // the frame is sealed; its bytes follow in the background
        lap("seal");
        CHECK(pgsd_device_wait_packed(&h)); // ... and the arrays are the simulation's again
        lap("wait_packed");
        if (lap_on)
            fprintf(stderr, "frame %u:%s us\n", frame, laps);
        stall_us.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6);
        frame_step.push_back(step);
        frame_kept.push_back(kept);
        return 0;
    };

    // ---- the run
    hipEvent_t e0, e1, gap0, gap1;
    HIP(hipEventCreate(&e0));
    HIP(hipEventCreate(&e1));
    HIP(hipEventCreate(&gap0));
    HIP(hipEventCreate(&gap1));
    std::vector<double> gap_us;
    if (snapshot(0))
        return 1;
    CHECK(pgsd_comm_barrier());
    const auto run0 = std::chrono::steady_clock::now();
    float step_ms_sum = 0;
    int step_ms_n = 0;
    for (int s = 1; s <= steps; s++)
        {
        const bool timed = s % period == 1 || period == 1;
        if (timed)
            HIP(hipEventRecord(e0, sim));
        hipLaunchKernelGGL(step_kernel, dim3(blocks_for(n)), dim3(256), 0, sim, pos, vel, density, tag, n, row0, (uint64_t)s);
        if (timed)
            {
            HIP(hipEventRecord(e1, sim));
            HIP(hipEventSynchronize(e1));
            float ms = 0;
            HIP(hipEventElapsedTime(&ms, e0, e1));
            step_ms_sum += ms;
            step_ms_n++;
            }
        if (s % period == 0)
            {
            // what the snapshot costs the SIMULATION: the gap on its stream between the last step before and the first
            // step after (the host-side figure also holds the steps that were still queued when the snapshot was called)
            HIP(hipEventRecord(gap0, sim));
            if (snapshot((uint64_t)s))
                return 1;
            HIP(hipEventRecord(gap1, sim));
            HIP(hipEventSynchronize(gap1));
            float ms = 0;
            HIP(hipEventElapsedTime(&ms, gap0, gap1));
            gap_us.push_back(ms * 1e3);
            }
        }
    HIP(hipStreamSynchronize(sim));
    const double run_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - run0).count();
    CHECK(pgsd_frame_sync(&h)); // every sealed frame is in the file
    const double drained_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - run0).count();
    struct pgsd_device_stats st;
    pgsd_device_get_stats(&h, &st, 0);
    struct pgsd_exchange_stats xs;
    pgsd_get_exchange_stats(&h, &xs, 0);
    CHECK(pgsd_close(&h));

    // ---- read it back through the reference's entry points and compare with the host model
    CHECK(pgsd_open(&h, path, PGSD_OPEN_READONLY));
    const uint64_t n_frames = pgsd_get_nframes(&h);
    int bad = n_frames == frame_step.size() ? 0 : 1;
    std::vector<uint64_t> my_tags; // the local tags this rank wrote, ascending
    for (uint64_t t = 0; t < n; t++)
        if (!fluid_only || type_of(row0 + t) != WALL_TYPE)
            my_tags.push_back(t);
    std::vector<float> model_pos(my_tags.size() * 3), model_vel(my_tags.size() * 4);
    for (size_t k = 0; k < my_tags.size(); k++)
        initial_row(row0 + my_tags[k], &model_pos[3 * k], &model_vel[4 * k]);
    // positions come out of `steps` dependent roundings per particle and are modelled step by step: for every row when
    // that is cheap, else for every stride-th one (everything else has a closed form and is compared in full)
    const uint64_t stride = std::max<uint64_t>(1, (uint64_t)((double)my_tags.size() * (double)steps / 3e8 + 0.999));
    uint64_t model_step = 0;
    std::vector<float> got3(my_tags.size() * 3), got1(my_tags.size());
    std::vector<uint32_t> gotu(my_tags.size());
    for (uint64_t f = 0; f < n_frames && !bad; f++)
        {
        const uint64_t kept = frame_kept[(size_t)f];
        if (kept != my_tags.size())
            bad = 2;
        // this rank's first row of the frame: the ranks' counts again (the file does not store them)
        uint64_t first = 0, total = 0;
        CHECK(pgsd_partition_rows(kept, &first, &total, NULL));
        const struct pgsd_index_entry* e = pgsd_find_chunk(&h, f, "configuration/step");
        uint64_t step = ~0ull;
        if (!e || pgsd_read_chunk(&h, &step, e, 1, 1, 0, false) != 0 || step != frame_step[(size_t)f])
            bad = 3;
        e = pgsd_find_chunk(&h, f, "particles/N");
        uint32_t n32 = 0;
        if (!e || pgsd_read_chunk(&h, &n32, e, 1, 1, 0, false) != 0 || n32 != (uint32_t)total)
            bad = 4;
        for (; model_step < step; model_step++) // the host model catches up: the kernel's arithmetic, step by step
            for (size_t k = 0; k < my_tags.size(); k += stride)
                {
                if (type_of(row0 + my_tags[k]) == WALL_TYPE)
                    continue;
                for (int c = 0; c < 3; c++)
                    model_pos[3 * k + c] = fmaf(model_vel[4 * k + c], DT, model_pos[3 * k + c]);
                }
        e = pgsd_find_chunk(&h, f, "particles/position");
        if (!e || e->N != total || e->M != 3 || (kept && pgsd_read_chunk(&h, got3.data(), e, kept, 3, (uint32_t)first, true) != 0))
            bad = bad ? bad : 5;
        for (size_t k = 0; k < my_tags.size() && !bad; k += stride)
            if (memcmp(&got3[3 * k], &model_pos[3 * k], 12) != 0)
                bad = 5;
        e = pgsd_find_chunk(&h, f, "particles/velocity");
        if (!e || (kept && pgsd_read_chunk(&h, got3.data(), e, kept, 3, (uint32_t)first, true) != 0))
            bad = bad ? bad : 6;
        e = pgsd_find_chunk(&h, f, "particles/density");
        if (!e || (kept && pgsd_read_chunk(&h, got1.data(), e, kept, 1, (uint32_t)first, true) != 0))
            bad = bad ? bad : 7;
        for (size_t k = 0; k < my_tags.size() && !bad; k++)
            {
            const uint64_t g = row0 + my_tags[k];
            float p[3], v[4];
            initial_row(g, p, v);
            if (memcmp(&got3[3 * k], v, 12) != 0)
                bad = 6;
            const float d = density_at(g, step);
            if (memcmp(&got1[k], &d, 4) != 0)
                bad = 7;
            }
        // the static arrays: in frame 0, and nowhere else
        const struct pgsd_index_entry* et = pgsd_find_chunk(&h, f, "particles/typeid");
        const struct pgsd_index_entry* em = pgsd_find_chunk(&h, f, "particles/mass");
        if (f == 0)
            {
            if (!et || !em || (kept && pgsd_read_chunk(&h, gotu.data(), et, kept, 1, (uint32_t)first, true) != 0)
                || (kept && pgsd_read_chunk(&h, got1.data(), em, kept, 1, (uint32_t)first, true) != 0))
                bad = bad ? bad : 8;
            for (size_t k = 0; k < my_tags.size() && !bad; k++)
                {
                const uint64_t g = row0 + my_tags[k];
                float p[3], v[4];
                initial_row(g, p, v);
                if (gotu[k] != type_of(g) || memcmp(&got1[k], &v[3], 4) != 0)
                    bad = 8;
                }
            }
        else if (et || em)
            bad = bad ? bad : 9;
        }
    CHECK(pgsd_close(&h));
    // everybody's verdict
    std::vector<int> verdicts((size_t)P);
    CHECK(pgsd_comm_allgather(&bad, verdicts.data(), sizeof(int)));
    int worst = 0;
    for (int r = 0; r < P; r++)
        worst = worst ? worst : verdicts[(size_t)r];

    if (rank == 0)
        {
        std::vector<double> s(stall_us.begin() + (stall_us.size() > 1 ? 1 : 0), stall_us.end()); // frame 0 allocates
        std::sort(s.begin(), s.end());
        char gaps[512] = "";
        for (size_t i = 0, at = 0; i < gap_us.size() && i < 24 && at < sizeof(gaps) - 16; i++)
            at += (size_t)snprintf(gaps + at, sizeof(gaps) - at, "%s%.0f", i ? ", " : "", gap_us[i]);
        std::sort(gap_us.begin(), gap_us.end());
        printf("{\"example\": \"dump_writer\", \"ranks\": %d, \"particles_per_rank\": %llu, \"group\": \"%s\", \"memory_order\": \"%s\", "
               "\"rows_per_frame_rank0\": %llu, \"steps\": %d, \"period\": %d, \"frames\": %llu, "
               "\"position_rows_modelled\": \"every %llu-th\", \"sim_gap_us_first_frames\": [%s], "
               "\"step_us\": %.1f, \"sim_gap_us_median\": %.1f, \"sim_gap_us_max\": %.1f, \"snapshot_call_us_median\": %.1f, "
               "\"snapshot_call_us_max\": %.1f, \"run_s\": %.4f, "
               "\"drained_after_s\": %.4f, \"pack_launches\": %llu, \"written_bytes_rank0\": %llu, "
               "\"collectives_rank0\": %llu, \"chunks_written\": %llu, \"chunks_elided\": %llu, "
               "\"verified_frames\": %llu, \"ok\": %s, \"failed_check\": %d}\n",
               P, (unsigned long long)n, fluid_only ? "fluid" : "all",
               hilbert ? "Hilbert curve x lattice tags" : "multiplicative permutation (uniform-like)", (unsigned long long)my_tags.size(), steps, period,
               (unsigned long long)n_frames, (unsigned long long)stride, gaps, step_ms_n ? step_ms_sum / step_ms_n * 1e3 : 0.0,
               gap_us.empty() ? 0.0 : gap_us[gap_us.size() / 2], gap_us.empty() ? 0.0 : gap_us.back(),
               s.empty() ? 0.0 : s[s.size() / 2], s.empty() ? 0.0 : s.back(), run_s, drained_s, (unsigned long long)st.pack_launches,
               (unsigned long long)st.written_bytes, (unsigned long long)xs.collectives, chunks_written, chunks_elided,
               (unsigned long long)n_frames, worst == 0 ? "true" : "false", worst);
        if (!keep)
            unlink(path);
        }
    pgsd_comm_finalize();
    for (int i = 0; i < N_CHUNKS; i++)
        (void)hipFree(frame0_rows[i]);
    (void)hipFree(pos);
    (void)hipFree(vel);
    (void)hipFree(density);
    (void)hipFree(tag);
    (void)hipFree(rtag);
    (void)hipFree(kept_tags);
    (void)hipFree(order);
    (void)hipFree(flags);
    return worst == 0 ? 0 : 1;
    }
