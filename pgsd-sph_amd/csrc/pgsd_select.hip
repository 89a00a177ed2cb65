// pgsd_select.hip -- the smaller gfx950 kernels around the pack path:
//   compare_bytes_kernel   packed chunk == reference rows?  (the GPU-side elision test of pgsd.hoomd: numpy's equality,
//                          repeating references; reads only, 0.82-0.86 of the HBM peak)
//   select_*_kernel        stream compaction for filtered snapshots: wave ballot / popcount scans give each workgroup's
//                          count, a one-block scan turns counts into offsets (= per-chunk row and byte counts) and
//                          hands the total to the host, a scatter pass writes the index list (pgsd_select_rows)
//   pgsd_device_alloc / _free / _copy   device memory owned by the library (pgsd.fl.DeviceBuffer)
// Shared device helpers: pgsd_kernels.hpp.
#include "pgsd_kernels.hpp"

namespace pgsd_amd
    {
// ------------------------------------------------------------------ packed chunk == reference rows ?
// pgsd.hoomd elides a per-particle array that equals frame 0's, or the schema's default where frame 0 has no such chunk
// (hoomd.py:654-694: numpy.array_equal / a broadcast comparison).  For arrays that live in HBM the test runs here: the
// chunk is packed as usual, then its packed bytes are compared with the reference rows (also in device memory) -- 16
// bytes per lane and load, four loads of each side in flight, grid-stride.  Bandwidth-bound when the arrays are equal
// (2 x chunk bytes read -- 1 x against a short REPEATING reference, which stays in the L2 --, nothing written).
// Equality is numpy's: integer chunks by their bytes, float chunks by VALUE -- a NaN differs from everything, itself
// included, +0.0 equals -0.0 -- decided on the bit patterns (no floating-point instruction, so no denormal mode can
// come into it).  Arrays that differ differ early, so a PROBE launch -- four workgroups per job over its first 64
// KiB -- runs first: the full launch's workgroups of a job the probe marked leave at once (had they all found the
// difference themselves, thousands of waves would each have sent their mark across PCIe: 237 us for two moving arrays
// of 10 M rows against 129 us for six equal ones).  A difference further in is still found by the full launch; the
// first workgroup to see it marks the job and the others stop at their next stride.  The flag words are never
// cleared: a launch marks with its own generation number.
// The common case is "equal": the test is shaped for it.  Per 16-byte vector: OR of the XORs (any bit differs?) and, for
// float chunks, the largest |x| bit pattern of the CHUNK's words shifted left by one (sign out): above 0xff000000 it
// is a NaN, which equals nothing -- itself included.  Only when bits differ does the slow look decide whether it is
// a +0.0 / -0.0 pair (equal by value) -- a path an equal array never takes and a different one leaves the kernel on.
template <int MODE> __device__ __forceinline__ uint32_t cmp_differ16(const u32x4 x, const u32x4 y)
    {
    const uint32_t differ = (x.x ^ y.x) | (x.y ^ y.y) | (x.z ^ y.z) | (x.w ^ y.w);
    if (MODE == CMP_BYTES)
        return differ;
    if (MODE == CMP_F32)
        {
        const uint32_t m = max(max(x.x << 1, x.y << 1), max(x.z << 1, x.w << 1));
        uint32_t bad = m > 0xff000000u ? 1u : 0u; // a NaN among the chunk's four floats
        if (differ != 0)
            {
            const uint32_t a[4] = {x.x, x.y, x.z, x.w}, b[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int k = 0; k < 4; k++) // bits differ: equal all the same when both are zeros of either sign
                bad |= ((a[k] ^ b[k]) != 0 && ((a[k] | b[k]) << 1) != 0) ? 1u : 0u;
            }
        return bad;
        }
    // CMP_F64: two doubles per vector, little endian (low word first).  (hi << 1) | (lo != 0) > 0xffe00000: a NaN
    const uint32_t t0 = (x.y << 1) | (x.x != 0 ? 1u : 0u), t1 = (x.w << 1) | (x.z != 0 ? 1u : 0u);
    uint32_t bad = max(t0, t1) > 0xffe00000u ? 1u : 0u;
    if (differ != 0)
        {
        const uint32_t al[2] = {x.x, x.z}, ah[2] = {x.y, x.w}, bl[2] = {y.x, y.z}, bh[2] = {y.y, y.w};
#pragma unroll
        for (int k = 0; k < 2; k++)
            bad |= (((al[k] ^ bl[k]) | (ah[k] ^ bh[k])) != 0 && (((ah[k] | bh[k]) << 1) | al[k] | bl[k]) != 0) ? 1u : 0u;
        }
    return bad;
    }

// one element of `es` bytes (1: a byte of an integer chunk) at byte offset `at`, assembled from bytes: the slow road of
// unaligned pointers and of the last bytes
__device__ __forceinline__ bool cmp_differ_element(const char* pa, const char* pb, uint64_t at, uint64_t at_b, uint32_t es,
                                                   uint32_t mode)
    {
    uint64_t a = 0, b = 0;
    for (uint32_t k = 0; k < es; k++)
        {
        a |= (uint64_t)(uint8_t)pa[at + k] << (8 * k);
        b |= (uint64_t)(uint8_t)pb[at_b + k] << (8 * k);
        }
    if (mode == CMP_F32)
        return ((a ^ b) != 0 && ((a | b) & 0x7fffffffull) != 0) || (a & 0x7fffffffull) > 0x7f800000ull;
    if (mode == CMP_F64)
        return ((a ^ b) != 0 && ((a | b) & 0x7fffffffffffffffull) != 0) || (a & 0x7fffffffffffffffull) > 0x7ff0000000000000ull;
    return a != b;
    }

template <int MODE, bool PERIODIC>
__device__ __forceinline__ bool cmp_vector_loop(const u32x4* a, const u32x4* b, uint64_t n16, uint64_t period16, const uint32_t* df,
                                                uint32_t gen)
    {
    const uint64_t per_block = 256 * 4;
    for (uint64_t base = (uint64_t)blockIdx.x * per_block; base < n16; base += (uint64_t)gridDim.x * per_block)
        {
        if (__hip_atomic_load(df, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen)
            return false; // somebody else has the answer
        // a repeating reference: ONE modulo per lane and stride, the three further vectors by a conditional step back
        // (period16 >= 256 is checked by the host)
        uint64_t bi = PERIODIC ? (base + threadIdx.x) % period16 : 0;
        u32x4 x[4], y[4];
#pragma unroll
        for (int k = 0; k < 4; k++)
            {
            const uint64_t i = base + (uint64_t)k * 256 + threadIdx.x;
            x[k] = (u32x4)(0u);
            y[k] = (u32x4)(0u);
            if (i < n16)
                {
                x[k] = __builtin_nontemporal_load(a + i);
                y[k] = PERIODIC ? b[bi] : __builtin_nontemporal_load(b + i);
                }
            if (PERIODIC)
                {
                bi += 256;
                if (bi >= period16)
                    bi -= period16;
                }
            }
        uint32_t acc = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            acc |= cmp_differ16<MODE>(x[k], y[k]);
        if (acc != 0)
            return true;
        }
    return false;
    }

__global__ __launch_bounds__(256) void compare_bytes_kernel(const CompareArgs args)
    {
    CompareJob jb = args.j[blockIdx.y];
    uint32_t* df = args.dflags + blockIdx.y;
    if (args.limit != 0 && jb.bytes > args.limit)
        jb.bytes = args.limit;
    if (__hip_atomic_load(df, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == args.gen)
        return; // marked by the probe (or by a quicker workgroup)
    bool diff = false;
    const char* pa = (const char*)jb.a;
    const char* pb = (const char*)jb.b;
    uint64_t done = 0; // bytes covered by the vector loop
    if ((((uintptr_t)pa | (uintptr_t)pb) & 15) == 0)
        {
        const u32x4* a = (const u32x4*)pa;
        const u32x4* b = (const u32x4*)pb;
        const uint64_t n16 = jb.bytes >> 4;
        const uint64_t p16 = jb.period >> 4;
        done = n16 << 4;
        if (jb.period == 0)
            diff = jb.mode == CMP_F32   ? cmp_vector_loop<CMP_F32, false>(a, b, n16, 0, df, args.gen)
                   : jb.mode == CMP_F64 ? cmp_vector_loop<CMP_F64, false>(a, b, n16, 0, df, args.gen)
                                        : cmp_vector_loop<CMP_BYTES, false>(a, b, n16, 0, df, args.gen);
        else
            diff = jb.mode == CMP_F32   ? cmp_vector_loop<CMP_F32, true>(a, b, n16, p16, df, args.gen)
                   : jb.mode == CMP_F64 ? cmp_vector_loop<CMP_F64, true>(a, b, n16, p16, df, args.gen)
                                        : cmp_vector_loop<CMP_BYTES, true>(a, b, n16, p16, df, args.gen);
        }
    // what the vector loop left: the last bytes, or everything when a side is not 16-byte aligned -- element by element
    const uint32_t es = jb.mode == CMP_F32 ? 4u : jb.mode == CMP_F64 ? 8u : 1u;
    for (uint64_t i = done + ((uint64_t)blockIdx.x * 256 + threadIdx.x) * es; i + es <= jb.bytes && !diff;
         i += (uint64_t)gridDim.x * 256 * es)
        diff = cmp_differ_element(pa, pb, i, jb.period ? i % jb.period : i, es, jb.mode);
    const uint64_t who = __ballot(diff);
    if (who != 0 && (uint32_t)(__ffsll((unsigned long long)who) - 1) == (threadIdx.x & 63u))
        {
        __hip_atomic_store(df, args.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(args.hflags + blockIdx.y, args.gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

void warm_select_kernels()
    {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, (const void*)compare_bytes_kernel);
    (void)hipGetLastError();
    }

int launch_compare(uint32_t n_jobs, const CompareJob* jobs, uint32_t gen, uint32_t* dflags, uint32_t* hflags,
                   hipStream_t stream, std::string* err)
    {
    if (n_jobs == 0)
        return PGSD_SUCCESS;
    if (n_jobs > CMP_MAX_JOBS || !jobs || !dflags || !hflags)
        return PGSD_ERROR_INVALID_ARGUMENT;
    CompareArgs args;
    memset(&args, 0, sizeof(args));
    args.dflags = dflags;
    args.hflags = hflags;
    args.gen = gen;
    args.n_jobs = n_jobs;
    uint64_t most = 0;
    for (uint32_t i = 0; i < n_jobs; i++)
        {
        args.j[i] = jobs[i];
        most = std::max<uint64_t>(most, jobs[i].bytes);
        }
    // one workgroup per 16 KiB of the longest job, at most eight per CU of the part (2048): grid-stride beyond
    uint64_t blocks = (most + 16383) / 16384;
    blocks = std::min<uint64_t>(std::max<uint64_t>(blocks, 1), 2048);
    // whatever an earlier call of this thread left in the runtime's last-error slot (a failed hipMalloc, the caller's own
    // calls) is not this launch's: the slot is read again right behind the launches
    (void)hipGetLastError();
    if (most > 65536)
        {
        args.limit = 65536;
        hipLaunchKernelGGL(compare_bytes_kernel, dim3(4, n_jobs), dim3(256), 0, stream, args); // 16 KiB per workgroup
        args.limit = 0;
        }
    hipLaunchKernelGGL(compare_bytes_kernel, dim3((unsigned)blocks, n_jobs), dim3(256), 0, stream, args);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        {
        if (err)
            *err = std::string("compare kernel launch failed: ") + hipGetErrorString(e);
        return PGSD_ERROR_DEVICE;
        }
    return PGSD_SUCCESS;
    }

// ------------------------------------------------------------------ select (compaction)
#define SEL_THREADS 256
#define SEL_PER_THREAD 16
#define SEL_PER_BLOCK (SEL_THREADS * SEL_PER_THREAD)

// number of non-zero flag bytes among the 16 this lane owns
__device__ __forceinline__ uint32_t sel_load16(const uint8_t* flags, uint64_t base, uint64_t N, uint32_t* mask)
    {
    uint32_t m = 0;
    if (base + SEL_PER_THREAD <= N && ((uintptr_t)(flags + base) & 15) == 0)
        {
        u32x4 v = *(const u32x4*)(flags + base);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; k++)
            m |= (((w[k >> 2] >> (8 * (k & 3))) & 0xffu) != 0 ? 1u : 0u) << k;
        }
    else
        {
        for (int k = 0; k < 16; k++)
            if (base + k < N && flags[base + k] != 0)
                m |= 1u << k;
        }
    *mask = m;
    return (uint32_t)__popc(m);
    }

// inclusive scan of one value per lane across the 64-lane wavefront
__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t x)
    {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
        {
        uint32_t y = __shfl_up(x, d, 64);
        if (lane >= d)
            x += y;
        }
    return x;
    }

// exclusive scan of the block counts by ONE workgroup; also writes the total -- to device memory and straight into the
// caller's pinned word (a system-scope store: no copy command behind the kernels)
__device__ __forceinline__ void select_scan_block(const uint32_t* block_counts, uint32_t n_blocks, uint64_t* block_offsets,
                                                  uint64_t* out_count, uint64_t* out_count_host, uint32_t* wave_sums,
                                                  uint64_t* carry)
    {
    if (threadIdx.x == 0)
        *carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_blocks; b0 += SEL_THREADS)
        {
        uint32_t i = b0 + threadIdx.x;
        uint32_t c = i < n_blocks ? block_counts[i] : 0;
        uint32_t inc = wave_inclusive_scan(c);
        if ((threadIdx.x & 63) == 63)
            wave_sums[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint32_t wave_off = 0;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); w++)
            wave_off += wave_sums[w];
        if (i < n_blocks)
            block_offsets[i] = *carry + wave_off + inc - c;
        __syncthreads();
        if (threadIdx.x == SEL_THREADS - 1)
            *carry += (uint64_t)wave_off + inc;
        __syncthreads();
        }
    if (threadIdx.x == 0)
        {
        *out_count = *carry;
        __hip_atomic_store(out_count_host, *carry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }

__global__ __launch_bounds__(SEL_THREADS) void select_count_kernel(const uint8_t* flags, uint64_t N, uint32_t* block_counts)
    {
    __shared__ uint32_t wave_sums[SEL_THREADS / 64];
    uint64_t base = ((uint64_t)blockIdx.x * SEL_THREADS + threadIdx.x) * SEL_PER_THREAD;
    uint32_t mask;
    uint32_t c = base < N ? sel_load16(flags, base, N, &mask) : 0;
    uint32_t inc = wave_inclusive_scan(c);
    if ((threadIdx.x & 63) == 63)
        wave_sums[threadIdx.x >> 6] = inc;
    __syncthreads();
    if (threadIdx.x == 0)
        block_counts[blockIdx.x] = wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
    }

// (A launch of its own: letting the LAST counting workgroup scan -- a ticket counter, release / acquire at device scope
// in every workgroup -- was measured at 87-95 us per call against 37-44: on a part whose eight L2s are not coherent
// with each other such fences write back and invalidate whole caches; profiles/r05_select_bench.jsonl.)
__global__ __launch_bounds__(SEL_THREADS) void select_scan_kernel(const uint32_t* block_counts, uint32_t n_blocks,
                                                                  uint64_t* block_offsets, uint64_t* out_count,
                                                                  uint64_t* out_count_host)
    {
    __shared__ uint32_t wave_sums[SEL_THREADS / 64];
    __shared__ uint64_t carry;
    select_scan_block(block_counts, n_blocks, block_offsets, out_count, out_count_host, wave_sums, &carry);
    }

__global__ __launch_bounds__(SEL_THREADS) void select_scatter_kernel(const uint8_t* flags, uint64_t N,
                                                                     const uint64_t* block_offsets,
                                                                     uint32_t* out_index)
    {
    // The kept rows of this block are compacted in LDS first (each lane drops its <= 16 indices at
    // its block-local rank), then the block writes them out as one dense, coalesced run: lane i
    // stores element i of the run instead of 16 scattered stores per lane.
    __shared__ uint32_t wave_sums[SEL_THREADS / 64];
    __shared__ uint32_t local[SEL_PER_BLOCK];
    uint64_t base = ((uint64_t)blockIdx.x * SEL_THREADS + threadIdx.x) * SEL_PER_THREAD;
    uint32_t mask = 0;
    uint32_t c = base < N ? sel_load16(flags, base, N, &mask) : 0;
    uint32_t inc = wave_inclusive_scan(c);
    if ((threadIdx.x & 63) == 63)
        wave_sums[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t wave_off = 0, total = 0;
    for (uint32_t w = 0; w < SEL_THREADS / 64; w++)
        {
        if (w < (threadIdx.x >> 6))
            wave_off += wave_sums[w];
        total += wave_sums[w];
        }
    uint32_t pos = wave_off + inc - c;
    while (mask)
        {
        int k = __ffs((int)mask) - 1;
        mask &= mask - 1;
        local[pos++] = (uint32_t)(base + (uint64_t)k);
        }
    __syncthreads();
    uint32_t* out = out_index + block_offsets[blockIdx.x];
    // 16-byte stores where the run's start allows, 4-byte stores for the ragged ends
    const uint32_t lead = (uint32_t)((4u - (((uintptr_t)out >> 2) & 3u)) & 3u);
    const uint32_t head = lead < total ? lead : total;
    if (threadIdx.x < head)
        out[threadIdx.x] = local[threadIdx.x];
    const uint32_t nvec = (total - head) >> 2;
    for (uint32_t v = threadIdx.x; v < nvec; v += SEL_THREADS)
        {
        const uint32_t e = head + 4 * v;
        u32x4 q = {local[e], local[e + 1], local[e + 2], local[e + 3]};
        *(u32x4*)(out + e) = q;
        }
    for (uint32_t e = head + 4 * nvec + threadIdx.x; e < total; e += SEL_THREADS)
        out[e] = local[e];
    }
    } // namespace pgsd_amd

using namespace pgsd_amd;

namespace
    {
struct SelectScratch
    {
    void* dev = nullptr;
    size_t cap = 0;
    uint64_t* host_count = nullptr;     // pinned, device-mapped: the scan writes the count into it
    uint64_t* host_count_dev = nullptr; // ... through this alias
    };
std::mutex g_select_lock;
std::map<int, SelectScratch> g_select_scratch;
    } // namespace

extern "C" int pgsd_select_rows(const uint8_t* flags, uint64_t N, uint32_t* out_index, uint64_t* out_count_host, void* stream_)
    try
    {
    if (!out_count_host || (N > 0 && (!flags || !out_index)) || N >= (1ull << 32))
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (!pgsd_device_available())
        {
        set_last_error("pgsd_select_rows: no HIP device visible (the HIP path has no CPU fallback)");
        return PGSD_ERROR_NO_DEVICE;
        }
    if (N == 0)
        {
        *out_count_host = 0;
        return PGSD_SUCCESS;
        }
    std::lock_guard<std::mutex> guard(g_select_lock);
    // the scratch space lives on the GPU the FLAGS live on (a process with several GPUs need not have made it current)
    int current = 0, device = 0;
    if (hipGetDevice(&current) != hipSuccess)
        return PGSD_ERROR_DEVICE;
    device = current;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, flags) == hipSuccess && attr.type == hipMemoryTypeDevice)
        device = attr.device;
    else
        (void)hipGetLastError();
    struct DeviceScope // restore the caller's current device on every way out
        {
        int back;
        bool on;
        ~DeviceScope()
            {
            if (on)
                (void)hipSetDevice(back);
            }
        } scope {current, device != current};
    if (scope.on && hipSetDevice(device) != hipSuccess)
        return PGSD_ERROR_DEVICE;
    SelectScratch& sc = g_select_scratch[device];
        {
        uint64_t nb = (N + SEL_PER_BLOCK - 1) / SEL_PER_BLOCK;
        // the count (u64) + block_counts (u32) rounded to 8 bytes + block_offsets (u64)
        const size_t need = 8 + (size_t)(((nb * 4 + 7) & ~7ull) + nb * 8);
        if (need > sc.cap)
            {
            if (sc.dev)
                (void)hipFree(sc.dev);
            sc.dev = nullptr;
            sc.cap = 0;
            const size_t cap = std::max<size_t>(need * 2, 1u << 16);
            if (hipMalloc(&sc.dev, cap) != hipSuccess)
                {
                sc.dev = nullptr;
                set_last_error("pgsd_select_rows: cannot allocate the scratch space");
                return PGSD_ERROR_MEMORY_ALLOCATION_FAILED;
                }
            sc.cap = cap;
            }
        if (!sc.host_count)
            {
            void* alias = nullptr;
            if (hipHostMalloc((void**)&sc.host_count, sizeof(uint64_t), hipHostMallocMapped) != hipSuccess
                || hipHostGetDevicePointer(&alias, sc.host_count, 0) != hipSuccess)
                {
                if (sc.host_count)
                    (void)hipHostFree(sc.host_count);
                sc.host_count = nullptr;
                set_last_error("pgsd_select_rows: cannot allocate pinned memory");
                return PGSD_ERROR_MEMORY_ALLOCATION_FAILED;
                }
            sc.host_count_dev = (uint64_t*)alias;
            }
        }
    // whatever an earlier call of this thread left in the runtime's last-error slot (a failed hipMalloc, the caller's own
    // calls) is not this launch's: the slot is read again right behind the launches
    (void)hipGetLastError();
    uint64_t* out_count = (uint64_t*)sc.dev;
    void* workspace = (char*)sc.dev + 8;
    hipStream_t stream = (hipStream_t)stream_;
    uint64_t n_blocks = (N + SEL_PER_BLOCK - 1) / SEL_PER_BLOCK;
        {
        uint32_t* block_counts = (uint32_t*)workspace;
        uint64_t* block_offsets = (uint64_t*)((char*)workspace + ((n_blocks * 4 + 7) & ~7ull));
        hipLaunchKernelGGL(select_count_kernel, dim3((unsigned)n_blocks), dim3(SEL_THREADS), 0, stream, flags, N,
                           block_counts);
        hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(SEL_THREADS), 0, stream, block_counts,
                           (uint32_t)n_blocks, block_offsets, out_count, sc.host_count_dev);
        hipLaunchKernelGGL(select_scatter_kernel, dim3((unsigned)n_blocks), dim3(SEL_THREADS), 0, stream, flags,
                           N, block_offsets, out_index);
        }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        {
        set_last_error(std::string("select kernel launch failed: ") + hipGetErrorString(e));
        return PGSD_ERROR_DEVICE;
        }
    e = hipStreamSynchronize(stream); // the kernels are through: the count is in the pinned word
    if (e != hipSuccess)
        {
        set_last_error(std::string("pgsd_select_rows: ") + hipGetErrorString(e));
        return PGSD_ERROR_DEVICE;
        }
    *out_count_host = __atomic_load_n(sc.host_count, __ATOMIC_ACQUIRE);
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" void* pgsd_device_alloc(int device, size_t bytes, const void* pattern, size_t pattern_bytes)
    try
    {
    if (!pgsd_device_available())
        {
        set_last_error("pgsd_device_alloc: no HIP device visible (the HIP path has no CPU fallback)");
        return nullptr;
        }
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (device >= 0 && device != prev && hipSetDevice(device) != hipSuccess)
        {
        set_last_error("pgsd_device_alloc: no device " + std::to_string(device));
        return nullptr;
        }
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, std::max<size_t>(bytes, 16));
    if (e == hipSuccess && pattern && pattern_bytes > 0 && bytes > 0)
        {
        // the pattern repeated over a host image of at most 1 MiB (a multiple of the pattern), copied piecewise
        const size_t reps = std::max<size_t>(1, std::min<size_t>((1u << 20) / pattern_bytes, (bytes + pattern_bytes - 1) / pattern_bytes));
        std::vector<char> img(reps * pattern_bytes);
        for (size_t r = 0; r < reps; r++)
            memcpy(img.data() + r * pattern_bytes, pattern, pattern_bytes);
        for (size_t at = 0; at < bytes && e == hipSuccess; at += img.size())
            e = hipMemcpy((char*)p + at, img.data(), std::min(img.size(), bytes - at), hipMemcpyHostToDevice);
        }
    if (e != hipSuccess)
        {
        set_last_error(std::string("pgsd_device_alloc: ") + hipGetErrorString(e));
        if (p)
            (void)hipFree(p);
        p = nullptr;
        }
    if (device >= 0 && prev >= 0 && device != prev)
        (void)hipSetDevice(prev);
    return p;
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return nullptr;
    }

extern "C" int pgsd_device_free(int device, void* ptr)
    try
    {
    if (!ptr)
        return PGSD_SUCCESS;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (device >= 0 && device != prev)
        (void)hipSetDevice(device);
    const hipError_t e = hipFree(ptr);
    if (device >= 0 && prev >= 0 && device != prev)
        (void)hipSetDevice(prev);
    if (e != hipSuccess)
        {
        set_last_error(std::string("pgsd_device_free: ") + hipGetErrorString(e));
        return PGSD_ERROR_DEVICE;
        }
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_device_copy(int device, void* dst, const void* src, size_t bytes)
    try
    {
    if (bytes == 0)
        return PGSD_SUCCESS;
    if (!dst || !src)
        return PGSD_ERROR_INVALID_ARGUMENT;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (device >= 0 && device != prev)
        (void)hipSetDevice(device);
    const hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyDefault); // either side may be host memory
    if (device >= 0 && prev >= 0 && device != prev)
        (void)hipSetDevice(prev);
    if (e != hipSuccess)
        {
        set_last_error(std::string("pgsd_device_copy: ") + hipGetErrorString(e));
        return PGSD_ERROR_DEVICE;
        }
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }
