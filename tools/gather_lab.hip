// tools/gather_lab.hip -- kernel-structure experiments for the tag-order gather pack (gfx950).
//
// Not product code.  The product's gather (chunk row i = source row order[i], a random permutation: snapshots in
// tag order) runs through the LDS-tiled kernel at 414 us for 10 M particles: every random 16-byte row costs a 64-byte
// fetch (FETCH_SIZE 1.3 GB against 320 MB of rows), and pos and vel rows are fetched side by side, so 320 MB of
// randomly touched lines compete for the 256 MiB Infinity Cache.  Question: does streaming ARRAY AFTER ARRAY (the
// row kernel's group-major grid) keep one array's lines cache-resident until their other rows are asked for?
//
//   both      one lane: idx = order[i]; pos row and vel row loaded together; x3 + x3 + x1 stores
//   major     2-D grid, blockIdx.y = array: all of pos, then all of vel (one launch)
//   twice     two launches: pos (+ id), then vel
// each with cached or non-temporal row loads; stores are non-temporal.  Dispatch begin/end stamps, 3 buffer sets.
//
//   hipcc -O3 --offload-arch=gfx950 tools/gather_lab.hip -o tools/build/gather_lab
//   ./gather_lab [N=10000000] [reps=20]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

#define CK(x)                                                                                 \
    do                                                                                        \
        {                                                                                     \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess)                                                                 \
            {                                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                          \
            }                                                                                 \
        } while (0)

template<bool NT> __device__ __forceinline__ u32x4 ld_row(const u32x4* p)
    {
    if (NT)
        return __builtin_nontemporal_load(p);
    return *p;
    }

__device__ __forceinline__ void st3(u32x4 r, uint32_t* out, uint64_t i)
    {
    u32x3 v = {r.x, r.y, r.z};
    __builtin_nontemporal_store(v, (u32x3*)(out + i * 3));
    }

template<int U, bool NT>
__global__ __launch_bounds__(256) void both_kernel(const u32x4* __restrict__ pos, const u32x4* __restrict__ vel,
                                                   const uint32_t* __restrict__ order, uint32_t* opos, uint32_t* ovel,
                                                   uint32_t* oid, uint64_t N)
    {
    const uint64_t base = (uint64_t)blockIdx.x * (256 * U) + threadIdx.x;
    uint32_t idx[U];
    u32x4 p[U], v[U];
#pragma unroll
    for (int k = 0; k < U; k++)
        {
        const uint64_t i = base + (uint64_t)k * 256;
        idx[k] = i < N ? __builtin_nontemporal_load(order + i) : 0;
        }
#pragma unroll
    for (int k = 0; k < U; k++)
        {
        p[k] = ld_row<NT>(pos + idx[k]);
        v[k] = ld_row<NT>(vel + idx[k]);
        }
#pragma unroll
    for (int k = 0; k < U; k++)
        {
        const uint64_t i = base + (uint64_t)k * 256;
        if (i < N)
            {
            st3(p[k], opos, i);
            st3(v[k], ovel, i);
            __builtin_nontemporal_store(p[k].w, oid + i);
            }
        }
    }

// blockIdx.y = 0: pos -> opos + oid;  1: vel -> ovel
template<int U, bool NT>
__global__ __launch_bounds__(256) void major_kernel(const u32x4* __restrict__ pos, const u32x4* __restrict__ vel,
                                                    const uint32_t* __restrict__ order, uint32_t* opos, uint32_t* ovel,
                                                    uint32_t* oid, uint64_t N, int only)
    {
    const int which = only >= 0 ? only : (int)blockIdx.y;
    const u32x4* src = which == 0 ? pos : vel;
    uint32_t* out = which == 0 ? opos : ovel;
    const uint64_t base = (uint64_t)blockIdx.x * (256 * U) + threadIdx.x;
    uint32_t idx[U];
    u32x4 r[U];
#pragma unroll
    for (int k = 0; k < U; k++)
        {
        const uint64_t i = base + (uint64_t)k * 256;
        idx[k] = i < N ? __builtin_nontemporal_load(order + i) : 0;
        }
#pragma unroll
    for (int k = 0; k < U; k++)
        r[k] = ld_row<NT>(src + idx[k]);
#pragma unroll
    for (int k = 0; k < U; k++)
        {
        const uint64_t i = base + (uint64_t)k * 256;
        if (i < N)
            {
            st3(r[k], out, i);
            if (which == 0)
                __builtin_nontemporal_store(r[k].w, oid + i);
            }
        }
    }

__global__ void check_kernel(const uint32_t* pos, const uint32_t* vel, const uint32_t* order, const uint32_t* opos,
                             const uint32_t* ovel, const uint32_t* oid, uint64_t N, unsigned long long* bad)
    {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N)
        return;
    uint64_t s = order[i];
    bool ok = oid[i] == pos[s * 4 + 3];
    for (int c = 0; c < 3; c++)
        ok = ok && opos[i * 3 + c] == pos[s * 4 + c] && ovel[i * 3 + c] == vel[s * 4 + c];
    if (!ok)
        atomicAdd(bad, 1ull);
    }

__global__ void fill_kernel(uint32_t* p, uint64_t n, uint32_t seed)
    {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = (uint32_t)(i * 2654435761u) ^ seed;
    }

struct Set
    {
    u32x4 *pos, *vel;
    uint32_t *order, *opos, *ovel, *oid;
    };

int main(int argc, char** argv)
    {
    const uint64_t N = argc > 1 ? strtoull(argv[1], NULL, 10) : 10000000ull;
    const int reps = argc > 2 ? atoi(argv[2]) : 20;
    const int NSETS = 3;
    std::vector<uint32_t> perm(N);
    std::iota(perm.begin(), perm.end(), 0u);
    std::mt19937_64 rng(1234);
    std::shuffle(perm.begin(), perm.end(), rng);
    Set sets[NSETS];
    for (int s = 0; s < NSETS; s++)
        {
        CK(hipMalloc((void**)&sets[s].pos, N * 16));
        CK(hipMalloc((void**)&sets[s].vel, N * 16));
        CK(hipMalloc((void**)&sets[s].order, N * 4));
        CK(hipMalloc((void**)&sets[s].opos, N * 12));
        CK(hipMalloc((void**)&sets[s].ovel, N * 12));
        CK(hipMalloc((void**)&sets[s].oid, N * 4));
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((N * 4 + 255) / 256)), dim3(256), 0, 0, (uint32_t*)sets[s].pos, N * 4, 17u + s);
        hipLaunchKernelGGL(fill_kernel, dim3((unsigned)((N * 4 + 255) / 256)), dim3(256), 0, 0, (uint32_t*)sets[s].vel, N * 4, 99u + s);
        CK(hipMemcpy(sets[s].order, perm.data(), N * 4, hipMemcpyHostToDevice));
        }
    unsigned long long* bad;
    CK(hipMalloc((void**)&bad, 8));
    hipEvent_t e0, e1, e2;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipEventCreate(&e2));

    auto run = [&](const char* name, auto launch, bool two)
    {
        std::vector<float> ms;
        for (int it = 0; it < reps + 3; it++)
            {
            Set& s = sets[it % NSETS];
            CK(hipMemsetAsync(s.oid, 0, N * 4, 0));
            float t = launch(s);
            if (it >= 3)
                ms.push_back(t);
            }
        CK(hipDeviceSynchronize());
        Set& s = sets[(reps + 2) % NSETS];
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(check_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, 0, (const uint32_t*)s.pos,
                           (const uint32_t*)s.vel, s.order, s.opos, s.ovel, s.oid, N, bad);
        unsigned long long h = 0;
        CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        std::sort(ms.begin(), ms.end());
        double med = ms[ms.size() / 2];
        printf("{\"variant\": \"%s\", \"N\": %llu, \"median_us\": %.1f, \"best_us\": %.1f, \"algorithmic_TBps\": %.2f, \"ok\": %s}\n",
               name, (unsigned long long)N, med * 1e3, ms[0] * 1e3, (double)N * 60.0 / (med * 1e-3) / 1e12, h == 0 ? "true" : "false");
        (void)two;
    };

#define LAUNCH1(K, GRID, ...)                                                              \
    [&](Set& s) -> float                                                                   \
    {                                                                                      \
        hipExtLaunchKernelGGL(K, GRID, dim3(256), 0, 0, e0, e1, 0, __VA_ARGS__);           \
        CK(hipEventSynchronize(e1));                                                       \
        float t;                                                                           \
        CK(hipEventElapsedTime(&t, e0, e1));                                               \
        return t;                                                                          \
    }
    const unsigned b2 = (unsigned)((N + 511) / 512), b4 = (unsigned)((N + 1023) / 1024), b1 = (unsigned)((N + 255) / 256);
    run("both U=2 cached", LAUNCH1((both_kernel<2, false>), dim3(b2), s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N), false);
    run("both U=2 nt", LAUNCH1((both_kernel<2, true>), dim3(b2), s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N), false);
    run("both U=4 cached", LAUNCH1((both_kernel<4, false>), dim3(b4), s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N), false);
    run("major U=1 cached", LAUNCH1((major_kernel<1, false>), dim3(b1, 2), s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N, -1), false);
    run("major U=2 cached", LAUNCH1((major_kernel<2, false>), dim3(b2, 2), s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N, -1), false);
    run("major U=2 nt", LAUNCH1((major_kernel<2, true>), dim3(b2, 2), s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N, -1), false);
    run("major U=4 cached", LAUNCH1((major_kernel<4, false>), dim3(b4, 2), s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N, -1), false);
    run("major U=8 cached", LAUNCH1((major_kernel<8, false>), dim3((unsigned)((N + 2047) / 2048), 2), s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N, -1), false);
    // two launches: the sum of both dispatches
    auto twice = [&](Set& s) -> float
    {
        hipExtLaunchKernelGGL((major_kernel<4, false>), dim3(b4), dim3(256), 0, 0, e0, e1, 0, s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N, 0);
        hipExtLaunchKernelGGL((major_kernel<4, false>), dim3(b4), dim3(256), 0, 0, nullptr, e2, 0, s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N, 1);
        CK(hipEventSynchronize(e2));
        float t;
        CK(hipEventElapsedTime(&t, e0, e2));
        return t;
    };
    run("twice U=4 cached (2 launches, begin of 1st to end of 2nd)", twice, true);
    // round 3: two launches, pos rows only then vel rows only, EIGHT rows in flight per lane --
    // the eight index loads, then the eight row loads issued back to back, the stores behind staggered
    // s_waitcnt vmcnt(7..0) as the compiler schedules this loop shape (checked in the ISA)
    const unsigned b8 = (unsigned)((N + 2047) / 2048);
    auto twice8 = [&](Set& s) -> float
    {
        hipExtLaunchKernelGGL((major_kernel<8, false>), dim3(b8), dim3(256), 0, 0, e0, e1, 0, s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N, 0);
        hipExtLaunchKernelGGL((major_kernel<8, false>), dim3(b8), dim3(256), 0, 0, nullptr, e2, 0, s.pos, s.vel, s.order, s.opos, s.ovel, s.oid, N, 1);
        CK(hipEventSynchronize(e2));
        float t;
        CK(hipEventElapsedTime(&t, e0, e2));
        return t;
    };
    run("twice U=8 cached (2 launches, 8 rows in flight per lane)", twice8, true);
    return 0;
    }
