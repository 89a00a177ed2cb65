// tools/pack_lab.hip -- kernel-structure experiments for the pos+vel+typeid pack (gfx950).
//
// Not product code: a stand-alone bench that answers "what is the streaming ceiling of this box,
// and which kernel structure reaches it" for the traffic shape of the headline workload
// (read two float4 arrays, write N x 3, N x 3, N x 1 words):
//
//   copy      bare register float4 copy (no LDS, no barrier) moving the same number of bytes
//   direct    one particle row per lane: global_load_dwordx4 -> global_store_dwordx3 + dword
//             (each wave instruction still covers one contiguous span: 1 KiB / 768 B / 256 B)
//   lds96     wave-private LDS window: rows written as 12-byte pieces (the LDS image IS the chunk
//             stream), read back linearly, 16-byte stores; no workgroup barrier
//   product   pgsd_pack_fields() of libpgsd_amd.so for comparison (pass the .so path)
//
// Every launch is timed with the dispatch's own begin/end stamps (hipExtLaunchKernelGGL events);
// buffer sets are rotated so nothing is re-read from the 256 MiB Infinity Cache.
//
//   hipcc -O3 --offload-arch=gfx950 tools/pack_lab.hip -o gpurun_out/pack_lab -ldl
//   ./pack_lab [N=10000000] [reps=30] [libpgsd_amd.so]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <string>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));

#define CK(x)                                                                                   \
    do                                                                                          \
        {                                                                                       \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess)                                                                   \
            {                                                                                   \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(2);                                                                            \
            }                                                                                   \
        } while (0)

template<bool NT> __device__ __forceinline__ u32x4 ld(const u32x4* p)
    {
    if constexpr (NT)
        return __builtin_nontemporal_load(p);
    else
        return *p;
    }
template<bool NT> __device__ __forceinline__ void st(u32x4 v, u32x4* p)
    {
    if constexpr (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
    }
template<bool NT> __device__ __forceinline__ void st3(u32x3 v, uint32_t* p)
    {
    if constexpr (NT)
        __builtin_nontemporal_store(v, (u32x3*)p);
    else
        *(u32x3*)p = v;
    }
template<bool NT> __device__ __forceinline__ void st1(uint32_t v, uint32_t* p)
    {
    if constexpr (NT)
        __builtin_nontemporal_store(v, p);
    else
        *p = v;
    }

// ---------------------------------------------------------------- bare copy
// U independent 16-byte loads in flight per lane; a workgroup walks contiguous U*4 KiB pieces
template<int U, bool NT> __global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ a,
                                                                            u32x4* __restrict__ b, uint64_t nvec)
    {
    const uint64_t piece = (uint64_t)U * 256;
    for (uint64_t base = (uint64_t)blockIdx.x * piece; base < nvec; base += (uint64_t)gridDim.x * piece)
        {
        u32x4 r[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t v = base + (uint64_t)k * 256 + threadIdx.x;
            if (v < nvec)
                r[k] = ld<NT>(a + v);
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t v = base + (uint64_t)k * 256 + threadIdx.x;
            if (v < nvec)
                st<NT>(r[k], b + v);
            }
        }
    }

// ---------------------------------------------------------------- reference (obviously correct)
__global__ void ref_kernel(const uint32_t* pos, const uint32_t* vel, uint32_t* opos, uint32_t* ovel, uint32_t* oid,
                           uint64_t N)
    {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (uint64_t)gridDim.x * blockDim.x)
        {
        for (int c = 0; c < 3; c++)
            {
            opos[i * 3 + c] = pos[i * 4 + c];
            ovel[i * 3 + c] = vel[i * 4 + c];
            }
        oid[i] = pos[i * 4 + 3];
        }
    }

__global__ void cmp_kernel(const uint32_t* a, const uint32_t* b, uint64_t n, unsigned long long* bad)
    {
    unsigned long long c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        c += a[i] != b[i];
    if (c)
        atomicAdd(bad, c);
    }

// ---------------------------------------------------------------- direct: row per lane
template<int U, bool NT>
__global__ __launch_bounds__(256) void direct_kernel(const u32x4* __restrict__ pos, const u32x4* __restrict__ vel,
                                                     uint32_t* __restrict__ opos, uint32_t* __restrict__ ovel,
                                                     uint32_t* __restrict__ oid, uint64_t N)
    {
    const uint64_t piece = (uint64_t)U * 256;
    for (uint64_t base = (uint64_t)blockIdx.x * piece; base < N; base += (uint64_t)gridDim.x * piece)
        {
        u32x4 p[U], v[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
            if (i < N)
                {
                p[k] = ld<NT>(pos + i);
                v[k] = ld<NT>(vel + i);
                }
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
            if (i < N)
                {
                u32x3 a = {p[k].x, p[k].y, p[k].z};
                u32x3 b = {v[k].x, v[k].y, v[k].z};
                st3<NT>(a, opos + i * 3);
                st3<NT>(b, ovel + i * 3);
                st1<NT>(p[k].w, oid + i);
                }
            }
        }
    }

// direct, arrays one after the other inside a piece (all position loads+stores, then velocity):
// fewer distinct streams open at once per wave
template<int U, bool NT>
__global__ __launch_bounds__(256) void direct_seq_kernel(const u32x4* __restrict__ pos, const u32x4* __restrict__ vel,
                                                         uint32_t* __restrict__ opos, uint32_t* __restrict__ ovel,
                                                         uint32_t* __restrict__ oid, uint64_t N)
    {
    const uint64_t piece = (uint64_t)U * 256;
    for (uint64_t base = (uint64_t)blockIdx.x * piece; base < N; base += (uint64_t)gridDim.x * piece)
        {
        u32x4 p[U], v[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
            if (i < N)
                p[k] = ld<NT>(pos + i);
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
            if (i < N)
                v[k] = ld<NT>(vel + i);
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
            if (i < N)
                {
                u32x3 a = {p[k].x, p[k].y, p[k].z};
                st3<NT>(a, opos + i * 3);
                st1<NT>(p[k].w, oid + i);
                }
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
            if (i < N)
                {
                u32x3 b = {v[k].x, v[k].y, v[k].z};
                st3<NT>(b, ovel + i * 3);
                }
            }
        }
    }

// ---------------------------------------------------------------- direct, parametrised
// THREADS per workgroup, U rows per lane, cache policy of loads / stores chosen separately,
// SPLIT: 0 = every workgroup handles both arrays of its rows; 1 = first half of the grid packs
// position (+id), second half velocity; 2 = even workgroups position, odd workgroups velocity
template<int THREADS, int U, bool NTL, bool NTS, int SPLIT>
__global__ __launch_bounds__(THREADS) void direct2_kernel(const u32x4* __restrict__ pos, const u32x4* __restrict__ vel,
                                                          uint32_t* __restrict__ opos, uint32_t* __restrict__ ovel,
                                                          uint32_t* __restrict__ oid, uint64_t N)
    {
    const uint64_t piece = (uint64_t)U * THREADS;
    uint64_t b = blockIdx.x, nb = gridDim.x;
    int which = 0; // 0 both, 1 position only, 2 velocity only
    if (SPLIT == 1)
        {
        nb >>= 1;
        which = b < nb ? 1 : 2;
        if (b >= nb)
            b -= nb;
        }
    else if (SPLIT == 2)
        {
        which = (b & 1) ? 2 : 1;
        b >>= 1;
        nb >>= 1;
        }
    for (uint64_t base = b * piece; base < N; base += nb * piece)
        {
        u32x4 p[U], v[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * THREADS + threadIdx.x;
            if (i < N)
                {
                if (which != 2)
                    p[k] = ld<NTL>(pos + i);
                if (which != 1)
                    v[k] = ld<NTL>(vel + i);
                }
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * THREADS + threadIdx.x;
            if (i < N)
                {
                if (which != 2)
                    {
                    u32x3 a = {p[k].x, p[k].y, p[k].z};
                    st3<NTS>(a, opos + i * 3);
                    st1<NTS>(p[k].w, oid + i);
                    }
                if (which != 1)
                    {
                    u32x3 c = {v[k].x, v[k].y, v[k].z};
                    st3<NTS>(c, ovel + i * 3);
                    }
                }
            }
        }
    }

// two-array 16-byte copy in one kernel (what several concurrent streams cost without the 12-byte stores)
template<int U, bool NT>
__global__ __launch_bounds__(256) void copy2_kernel(const u32x4* __restrict__ a0, const u32x4* __restrict__ a1,
                                                    u32x4* __restrict__ b0, u32x4* __restrict__ b1, uint64_t nvec)
    {
    const uint64_t piece = (uint64_t)U * 256;
    for (uint64_t base = (uint64_t)blockIdx.x * piece; base < nvec; base += (uint64_t)gridDim.x * piece)
        {
        u32x4 r[U], q[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t v = base + (uint64_t)k * 256 + threadIdx.x;
            if (v < nvec)
                {
                r[k] = ld<NT>(a0 + v);
                q[k] = ld<NT>(a1 + v);
                }
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t v = base + (uint64_t)k * 256 + threadIdx.x;
            if (v < nvec)
                {
                st<NT>(r[k], b0 + v);
                st<NT>(q[k], b1 + v);
                }
            }
        }
    }

// ---------------------------------------------------------------- lds96: wave-private LDS window
// A wave owns 256 consecutive rows per step: 4 loads of 1 KiB per source array, rows written to LDS
// as 12-byte pieces at row*12 (the image is the chunk stream), w words at 3072 + row*4; read back as
// 16-byte vectors and stored with 16 bytes per lane.  Only wave-level ordering is needed.
template<bool NT>
__global__ __launch_bounds__(256) void lds96_kernel(const u32x4* __restrict__ pos, const u32x4* __restrict__ vel,
                                                    u32x4* __restrict__ opos, u32x4* __restrict__ ovel,
                                                    u32x4* __restrict__ oid, uint64_t N)
    {
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[4][1024 + 768];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* L = lds_all[wave];
    const uint64_t n_steps = N / 256; // the lab keeps N a multiple of 256
    const uint64_t gw = (uint64_t)blockIdx.x * 4 + wave, nw = (uint64_t)gridDim.x * 4;
    for (uint64_t step = gw; step < n_steps; step += nw)
        {
        const uint64_t row0 = step * 256;
        u32x4 p[4], v[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            p[k] = ld<NT>(pos + row0 + k * 64 + lane);
#pragma unroll
        for (int k = 0; k < 4; k++)
            v[k] = ld<NT>(vel + row0 + k * 64 + lane);
#pragma unroll
        for (int k = 0; k < 4; k++)
            {
            const uint32_t r = k * 64 + lane;
            u32x3 a = {p[k].x, p[k].y, p[k].z};
            *(u32x3*)(L + r * 3) = a;
            L[768 + r] = p[k].w;
            u32x3 b = {v[k].x, v[k].y, v[k].z};
            *(u32x3*)(L + 1024 + r * 3) = b;
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        u32x4 o[7];
#pragma unroll
        for (int k = 0; k < 3; k++)
            o[k] = *(const u32x4*)(L + (k * 64 + lane) * 4);
        o[3] = *(const u32x4*)(L + 768 + lane * 4);
#pragma unroll
        for (int k = 0; k < 3; k++)
            o[4 + k] = *(const u32x4*)(L + 1024 + (k * 64 + lane) * 4);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int k = 0; k < 3; k++)
            st<NT>(o[k], opos + step * 192 + k * 64 + lane);
        st<NT>(o[3], oid + step * 64 + lane);
#pragma unroll
        for (int k = 0; k < 3; k++)
            st<NT>(o[4 + k], ovel + step * 192 + k * 64 + lane);
        }
    }

// ---------------------------------------------------------------- unpack direction (read path)
// chunks (N x 3 position, N x 1 id, N x 3 velocity, N x 1 mass) -> two Scalar4 arrays, 640 MB at 10 M rows
// (a) row per lane: 12-byte + 4-byte loads, 16-byte stores
template<int T, int U>
__global__ __launch_bounds__(T) void unrows_kernel(const uint32_t* __restrict__ cpos, const uint32_t* __restrict__ cid,
                                                   const uint32_t* __restrict__ cvel, const uint32_t* __restrict__ cmass,
                                                   u32x4* __restrict__ pos4, u32x4* __restrict__ vel4, uint64_t N)
    {
    const bool second = blockIdx.y != 0; // y = destination array
    const uint32_t* c3 = second ? cvel : cpos;
    const uint32_t* c1 = second ? cmass : cid;
    u32x4* dst = second ? vel4 : pos4;
    const uint64_t base = (uint64_t)blockIdx.x * (T * U) + threadIdx.x;
    u32x3 a[U];
    uint32_t w[U];
#pragma unroll
    for (int k = 0; k < U; k++)
        {
        const uint64_t i = base + (uint64_t)k * T;
        if (i < N)
            {
            a[k] = __builtin_nontemporal_load((const u32x3*)(c3 + i * 3));
            w[k] = __builtin_nontemporal_load(c1 + i);
            }
        }
#pragma unroll
    for (int k = 0; k < U; k++)
        {
        const uint64_t i = base + (uint64_t)k * T;
        if (i < N)
            {
            u32x4 o = {a[k].x, a[k].y, a[k].z, w[k]};
            __builtin_nontemporal_store(o, dst + i);
            }
        }
    }

// (b) wave-private LDS: every load is a linear 16-byte vector of the dense chunk stream (3 of position + 1 of
// the id per 256 rows and lane), rows are read back as 12-byte + 4-byte pieces; no workgroup barrier
__global__ __launch_bounds__(256) void unlds_kernel(const u32x4* __restrict__ cpos, const u32x4* __restrict__ cid,
                                                    const u32x4* __restrict__ cvel, const u32x4* __restrict__ cmass,
                                                    u32x4* __restrict__ pos4, u32x4* __restrict__ vel4, uint64_t N)
    {
    __shared__ __attribute__((aligned(16))) uint32_t lds_all[4][1024];
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t* L = lds_all[wave];
    const bool second = blockIdx.y != 0;
    const u32x4* c3 = second ? cvel : cpos;
    const u32x4* c1 = second ? cmass : cid;
    u32x4* dst = second ? vel4 : pos4;
    const uint64_t step = (uint64_t)blockIdx.x * 4 + wave; // 256 rows per wave
    if (step * 256 >= N)
        return;
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 3; k++)
        v[k] = __builtin_nontemporal_load(c3 + step * 192 + k * 64 + lane);
    v[3] = __builtin_nontemporal_load(c1 + step * 64 + lane);
#pragma unroll
    for (int k = 0; k < 3; k++)
        *(u32x4*)(L + (k * 64 + lane) * 4) = v[k];
    *(u32x4*)(L + 768 + lane * 4) = v[3];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int k = 0; k < 4; k++)
        {
        const uint32_t r = k * 64 + lane;
        const u32x3 a = *(const u32x3*)(L + r * 3);
        u32x4 o = {a.x, a.y, a.z, L[768 + r]};
        __builtin_nontemporal_store(o, dst + step * 256 + r);
        }
    }

// ---------------------------------------------------------------- harness
struct Set
    {
    uint32_t *pos, *vel, *opos, *ovel, *oid;
    };

static double med(std::vector<float> v)
    {
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
    }

struct pgsd_field_desc
    {
    const void* src;
    const uint32_t* order;
    uint32_t src_type, src_stride, src_col0, bitcast;
    };
struct pgsd_pack_job
    {
    void* dst;
    uint32_t dst_type, M;
    pgsd_field_desc src;
    };

int main(int argc, char** argv)
    {
    uint64_t N = argc > 1 ? strtoull(argv[1], 0, 10) : 10000000ull;
    int reps = argc > 2 ? atoi(argv[2]) : 30;
    const char* so = argc > 3 ? argv[3] : nullptr;
    N = (N / 256) * 256;
    const int NSETS = N * 60 > (1ull << 30) ? 3 : (int)std::min<uint64_t>(24, ((1ull << 30) / (N * 60)) + 2);
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"N\": %llu, \"sets\": %d, \"reps\": %d}\n", prop.gcnArchName, cus,
           (unsigned long long)N, NSETS, reps);

    std::vector<Set> sets(NSETS);
    for (auto& s : sets)
        {
        CK(hipMalloc(&s.pos, N * 16));
        CK(hipMalloc(&s.vel, N * 16));
        CK(hipMalloc(&s.opos, N * 12));
        CK(hipMalloc(&s.ovel, N * 12));
        CK(hipMalloc(&s.oid, N * 4));
        }
    // fill: pseudo-random words (host LCG for set 0, device copies + xor for the others is not needed:
    // the values do not matter for timing, only set 0 is verified)
        {
        std::vector<uint32_t> h(N * 4);
        uint32_t x = 12345;
        for (auto& w : h)
            {
            x = x * 1664525u + 1013904223u;
            w = x;
            }
        for (auto& s : sets)
            {
            CK(hipMemcpy(s.pos, h.data(), N * 16, hipMemcpyHostToDevice));
            for (auto& w : h)
                w ^= 0x9e3779b9u;
            CK(hipMemcpy(s.vel, h.data(), N * 16, hipMemcpyHostToDevice));
            }
        }
    uint32_t *rpos, *rvel, *rid;
    CK(hipMalloc(&rpos, N * 12));
    CK(hipMalloc(&rvel, N * 12));
    CK(hipMalloc(&rid, N * 4));
    unsigned long long* bad;
    CK(hipMalloc(&bad, 8));
    hipLaunchKernelGGL(ref_kernel, dim3(4096), dim3(256), 0, 0, sets[0].pos, sets[0].vel, rpos, rvel, rid, N);
    CK(hipDeviceSynchronize());

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const double pack_moved = (double)N * 60.0, pack_algo = (double)N * 56.0;

    auto verify = [&](const Set& s) -> unsigned long long
    {
        CK(hipMemset(bad, 0, 8));
        hipLaunchKernelGGL(cmp_kernel, dim3(2048), dim3(256), 0, 0, s.opos, rpos, N * 3, bad);
        hipLaunchKernelGGL(cmp_kernel, dim3(2048), dim3(256), 0, 0, s.ovel, rvel, N * 3, bad);
        hipLaunchKernelGGL(cmp_kernel, dim3(2048), dim3(256), 0, 0, s.oid, rid, N, bad);
        unsigned long long h = 0;
        CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
        return h;
    };
    auto clear_out = [&](const Set& s)
    {
        CK(hipMemset(s.opos, 0xff, N * 12));
        CK(hipMemset(s.ovel, 0xff, N * 12));
        CK(hipMemset(s.oid, 0xff, N * 4));
    };

    // run(name, launch(set, e0, e1), bytes_moved, bytes_algo, check)
    auto run = [&](const std::string& name, auto&& launch, double moved, double algo, bool check)
    {
        if (check)
            clear_out(sets[0]);
        for (int i = 0; i < 3; i++)
            launch(sets[i % NSETS], (hipEvent_t) nullptr, (hipEvent_t) nullptr);
        CK(hipDeviceSynchronize());
        unsigned long long nbad = check ? verify(sets[0]) : 0;
        std::vector<float> t;
        for (int i = 0; i < reps; i++)
            {
            launch(sets[i % NSETS], e0, e1);
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            t.push_back(ms * 1000.f);
            }
        // back-to-back batch (launch overhead excluded by the in-order queue): wall / reps
        CK(hipDeviceSynchronize());
        hipEvent_t b0, b1;
        CK(hipEventCreate(&b0));
        CK(hipEventCreate(&b1));
        CK(hipEventRecord(b0, 0));
        for (int i = 0; i < reps; i++)
            launch(sets[i % NSETS], (hipEvent_t) nullptr, (hipEvent_t) nullptr);
        CK(hipEventRecord(b1, 0));
        CK(hipEventSynchronize(b1));
        float bms = 0;
        CK(hipEventElapsedTime(&bms, b0, b1));
        CK(hipEventDestroy(b0));
        CK(hipEventDestroy(b1));
        const double m = med(t), mn = *std::min_element(t.begin(), t.end());
        printf("{\"kernel\": \"%s\", \"us_med\": %.2f, \"us_min\": %.2f, \"us_batch\": %.2f, \"moved_TBps\": %.3f, "
               "\"algo_TBps\": %.3f, \"frac_of_8TBps\": %.4f, \"mismatches\": %llu}\n",
               name.c_str(), m, mn, bms * 1000.0 / reps, moved / m / 1e6, algo / m / 1e6, algo / m / 1e6 / 8.0, nbad);
        fflush(stdout);
    };

    const char* only = getenv("LAB_ONLY"); // substring filter on the kernel name
    auto want = [&](const char* name) { return !only || strstr(name, only) != nullptr; };
    // ---- bare copies: 300 MB in, 300 MB out at N = 10 M (same bytes as the pack moves)
        {
        const uint64_t nvec = N * 30 / 16; // N*30 bytes each way
        uint32_t *ca[3], *cb[3];
        for (int i = 0; i < 3; i++)
            {
            CK(hipMalloc(&ca[i], nvec * 16));
            CK(hipMalloc(&cb[i], nvec * 16));
            CK(hipMemset(ca[i], i + 1, nvec * 16));
            }
        int ci = 0;
#define COPY(U, NT, PERCU)                                                                                   \
    if (want("copy U=" #U " nt=" #NT " wg/cu=" #PERCU))                                                      \
    run(std::string("copy U=" #U " nt=" #NT " wg/cu=" #PERCU),                                              \
        [&](const Set&, hipEvent_t a, hipEvent_t b)                                                          \
        {                                                                                                    \
            uint64_t blocks = std::min<uint64_t>((nvec + (U) * 256 - 1) / ((U) * 256),                      \
                                                 (PERCU) ? (uint64_t)cus * (PERCU) : ~0ull);                \
            hipExtLaunchKernelGGL((copy_kernel<U, NT>), dim3((unsigned)blocks), dim3(256), 0, 0, a, b, 0,    \
                                  (const u32x4*)ca[ci % 3], (u32x4*)cb[ci % 3], nvec);                       \
            ci++;                                                                                            \
        },                                                                                                   \
        (double)nvec * 32, (double)nvec * 32, false)
        COPY(1, true, 0);
        COPY(2, true, 0);
        COPY(4, true, 0);
        COPY(2, true, 8);
        COPY(1, false, 0);
#define COPY2(U, NT)                                                                                         \
    if (want("copy2 U=" #U " nt=" #NT))                                                                      \
    run(std::string("copy2 U=" #U " nt=" #NT " one-shot (two arrays, 16-byte stores)"),                     \
        [&](const Set&, hipEvent_t a, hipEvent_t b)                                                          \
        {                                                                                                    \
            const uint64_t h = nvec / 2;                                                                     \
            uint64_t blocks = (h + (U) * 256 - 1) / ((U) * 256);                                            \
            hipExtLaunchKernelGGL((copy2_kernel<U, NT>), dim3((unsigned)blocks), dim3(256), 0, 0, a, b, 0,   \
                                  (const u32x4*)ca[ci % 3], (const u32x4*)ca[ci % 3] + h, (u32x4*)cb[ci % 3], \
                                  (u32x4*)cb[ci % 3] + h, h);                                                \
            ci++;                                                                                            \
        },                                                                                                   \
        (double)(nvec / 2) * 64, (double)(nvec / 2) * 64, false)
        COPY2(1, true);
        COPY2(2, true);
        for (int i = 0; i < 3; i++)
            {
            CK(hipFree(ca[i]));
            CK(hipFree(cb[i]));
            }
        }

#define D2(THREADS, U, NTL, NTS, SPLIT, PERCU)                                                               \
    if (want("direct2 T=" #THREADS " U=" #U " ntl=" #NTL " nts=" #NTS " split=" #SPLIT " wg/cu=" #PERCU))  \
    run(std::string("direct2 T=" #THREADS " U=" #U " ntl=" #NTL " nts=" #NTS " split=" #SPLIT " wg/cu=" #PERCU), \
        [&](const Set& s, hipEvent_t a, hipEvent_t b)                                                        \
        {                                                                                                    \
            uint64_t blocks = (N + (U) * (THREADS) - 1) / ((U) * (THREADS));                                 \
            if (PERCU)                                                                                       \
                blocks = std::min<uint64_t>(blocks, (uint64_t)cus * (PERCU));                               \
            if (SPLIT)                                                                                       \
                blocks *= 2;                                                                                 \
            hipExtLaunchKernelGGL((direct2_kernel<THREADS, U, NTL, NTS, SPLIT>), dim3((unsigned)blocks),     \
                                  dim3(THREADS), 0, 0, a, b, 0, (const u32x4*)s.pos, (const u32x4*)s.vel,    \
                                  s.opos, s.ovel, s.oid, N);                                                 \
        },                                                                                                   \
        pack_moved, pack_algo, true)
    D2(256, 1, true, true, 0, 0);
    D2(256, 2, true, true, 0, 0);
    D2(256, 3, true, true, 0, 0);
    D2(256, 4, true, true, 0, 0);
    D2(64, 1, true, true, 0, 0);
    D2(64, 2, true, true, 0, 0);
    D2(64, 4, true, true, 0, 0);
    D2(128, 1, true, true, 0, 0);
    D2(128, 2, true, true, 0, 0);
    D2(512, 1, true, true, 0, 0);
    D2(512, 2, true, true, 0, 0);
    D2(1024, 1, true, true, 0, 0);
    D2(256, 1, false, true, 0, 0);
    D2(256, 1, true, false, 0, 0);
    D2(256, 1, false, false, 0, 0);
    D2(256, 1, true, true, 1, 0);
    D2(256, 2, true, true, 1, 0);
    D2(256, 1, true, true, 2, 0);
    D2(256, 2, true, true, 2, 0);
    D2(256, 4, true, true, 2, 0);
    D2(256, 1, true, true, 0, 16);
    D2(256, 1, true, true, 0, 32);
    D2(256, 2, true, true, 0, 32);

#define LDS96(NT, PERCU)                                                                                     \
    if (want("lds96 nt=" #NT " wg/cu=" #PERCU))                                                              \
    run(std::string("lds96 nt=" #NT " wg/cu=" #PERCU),                                                      \
        [&](const Set& s, hipEvent_t a, hipEvent_t b)                                                        \
        {                                                                                                    \
            uint64_t blocks = std::min<uint64_t>((N / 256 + 3) / 4, (PERCU) ? (uint64_t)cus * (PERCU) : ~0ull); \
            hipExtLaunchKernelGGL((lds96_kernel<NT>), dim3((unsigned)blocks), dim3(256), 0, 0, a, b, 0,      \
                                  (const u32x4*)s.pos, (const u32x4*)s.vel, (u32x4*)s.opos, (u32x4*)s.ovel,  \
                                  (u32x4*)s.oid, N);                                                         \
        },                                                                                                   \
        pack_moved, pack_algo, true)
    LDS96(true, 0);

    // ---- unpack direction: the packed chunks of set k -> Scalar4 arrays (reusing pos/vel of set k+1 as
    //      destinations); 640 MB moved at 10 M rows.  mass chunk = the id buffer of another set (values do
    //      not matter for timing; verification checks the xyz + id words).
        {
        uint32_t* mass;
        CK(hipMalloc(&mass, N * 4));
        CK(hipMemset(mass, 0x3f, N * 4));
        // reference chunks for every set: packed from its own pos/vel
        for (auto& st : sets)
            hipLaunchKernelGGL(ref_kernel, dim3(4096), dim3(256), 0, 0, st.pos, st.vel, st.opos, st.ovel, st.oid, N);
        CK(hipDeviceSynchronize());
        std::vector<Set> dsts(NSETS);
        for (auto& d : dsts)
            {
            CK(hipMalloc(&d.pos, N * 16));
            CK(hipMalloc(&d.vel, N * 16));
            }
        auto check_un = [&](int k) -> unsigned long long
        {
            // pos4 of the result must equal the source pos (xyz + id in w); vel4.xyz the source vel
            CK(hipMemset(bad, 0, 8));
            hipLaunchKernelGGL(cmp_kernel, dim3(2048), dim3(256), 0, 0, dsts[k].pos, sets[k].pos, N * 4, bad);
            unsigned long long h = 0;
            CK(hipMemcpy(&h, bad, 8, hipMemcpyDeviceToHost));
            return h;
        };
        const double un_moved = (double)N * 64.0;
        auto run_un = [&](const std::string& name, auto&& launch)
        {
            CK(hipMemset(dsts[0].pos, 0xff, N * 16));
            for (int i = 0; i < 3; i++)
                launch(i % NSETS, (hipEvent_t) nullptr, (hipEvent_t) nullptr);
            CK(hipDeviceSynchronize());
            unsigned long long nbad = check_un(0);
            std::vector<float> t;
            for (int i = 0; i < reps; i++)
                {
                launch(i % NSETS, e0, e1);
                CK(hipEventSynchronize(e1));
                float ms = 0;
                CK(hipEventElapsedTime(&ms, e0, e1));
                t.push_back(ms * 1000.f);
                }
            const double m = med(t), mn = *std::min_element(t.begin(), t.end());
            printf("{\"kernel\": \"%s\", \"us_med\": %.2f, \"us_min\": %.2f, \"moved_TBps\": %.3f, "
                   "\"frac_of_8TBps\": %.4f, \"mismatches\": %llu}\n",
                   name.c_str(), m, mn, un_moved / m / 1e6, un_moved / m / 1e6 / 8.0, nbad);
            fflush(stdout);
        };
#define UNROWS(T, U)                                                                                          \
    if (want("unrows T=" #T " U=" #U))                                                                        \
    run_un("unrows T=" #T " U=" #U,                                                                           \
           [&](int k, hipEvent_t a, hipEvent_t b)                                                             \
           {                                                                                                  \
               dim3 grid((unsigned)((N + (T) * (U)-1) / ((T) * (U))), 2);                                     \
               hipExtLaunchKernelGGL((unrows_kernel<T, U>), grid, dim3(T), 0, 0, a, b, 0, sets[k].opos, sets[k].oid, \
                                     sets[k].ovel, mass, (u32x4*)dsts[k].pos, (u32x4*)dsts[k].vel, N);        \
           })
        UNROWS(256, 2);
        UNROWS(64, 2);
        UNROWS(128, 1);
        if (want("unlds"))
            run_un("unlds (wave-private LDS, 16-byte loads)",
                   [&](int k, hipEvent_t a, hipEvent_t b)
                   {
                       dim3 grid((unsigned)((N / 256 + 3) / 4), 2);
                       hipExtLaunchKernelGGL(unlds_kernel, grid, dim3(256), 0, 0, a, b, 0, (const u32x4*)sets[k].opos,
                                             (const u32x4*)sets[k].oid, (const u32x4*)sets[k].ovel, (const u32x4*)mass,
                                             (u32x4*)dsts[k].pos, (u32x4*)dsts[k].vel, N);
                   });
        if (so && want("product unpack"))
            {
            void* lib = dlopen(so, RTLD_NOW);
            struct field_dst
                {
                void* dst;
                const uint32_t* order;
                uint32_t dst_type, dst_stride, dst_col0, bitcast;
                };
            struct unpack_job
                {
                const void* src;
                uint32_t src_type, M;
                field_dst dst;
                };
            auto unpack_fields = lib ? (int (*)(uint32_t, const unpack_job*, uint64_t, void*))dlsym(lib, "pgsd_unpack_fields")
                                     : nullptr;
            if (unpack_fields)
                run_un("product pgsd_unpack_fields (LDS-tiled)",
                       [&](int k, hipEvent_t a, hipEvent_t b)
                       {
                           unpack_job j[4] = {{sets[k].opos, 9, 3, {dsts[k].pos, nullptr, 9, 4, 0, 0}},
                                              {sets[k].oid, 3, 1, {dsts[k].pos, nullptr, 9, 4, 3, 1}},
                                              {sets[k].ovel, 9, 3, {dsts[k].vel, nullptr, 9, 4, 0, 0}},
                                              {mass, 9, 1, {dsts[k].vel, nullptr, 9, 4, 3, 0}}};
                           if (a)
                               CK(hipEventRecord(a, 0));
                           if (unpack_fields(4, j, N, nullptr) != 0)
                               exit(5);
                           if (b)
                               CK(hipEventRecord(b, 0));
                       });
            }
        }

    if (so)
        {
        void* lib = dlopen(so, RTLD_NOW);
        if (!lib)
            {
            fprintf(stderr, "dlopen %s: %s\n", so, dlerror());
            return 3;
            }
        auto pack_fields = (int (*)(uint32_t, const pgsd_pack_job*, uint64_t, void*))dlsym(lib, "pgsd_pack_fields");
        if (!pack_fields)
            return 3;
        // product launch: events recorded around the call on the null stream (includes launch gaps
        // only if several kernels are used; the headline layout is one launch)
        run("product pgsd_pack_fields",
            [&](const Set& s, hipEvent_t a, hipEvent_t b)
            {
                pgsd_pack_job j[3];
                memset(j, 0, sizeof(j));
                j[0] = {s.opos, 9, 3, {s.pos, nullptr, 9, 4, 0, 0}};
                j[1] = {s.ovel, 9, 3, {s.vel, nullptr, 9, 4, 0, 0}};
                j[2] = {s.oid, 3, 1, {s.pos, nullptr, 9, 4, 3, 1}};
                if (a)
                    CK(hipEventRecord(a, 0));
                if (pack_fields(3, j, N, nullptr) != 0)
                    {
                    fprintf(stderr, "pgsd_pack_fields failed\n");
                    exit(4);
                    }
                if (b)
                    CK(hipEventRecord(b, 0));
            },
            pack_moved, pack_algo, true);
        }
    return 0;
    }
