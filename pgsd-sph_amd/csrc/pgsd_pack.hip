// pgsd_pack.hip -- gfx950 (CDNA4 / MI355X) kernels of the snapshot pack path.
//
// pack:    chunk[i][c] = convert(src[(order ? order[i] : i) * stride + col0 + c])
//          for every field of a frame, from HBM-resident particle arrays (HOOMD-style
//          float4 / double4 / scalar arrays) into dense GSD chunk buffers.
// unpack:  the inverse for restart reads, all chunks of a frame in one launch, whole destination
//          rows assembled in registers where the launch restores every column of an array.
// select:  stream compaction (filtered snapshots): wave ballot / popcount scans give each
//          workgroup's count, a one-block scan turns counts into offsets (= per-chunk row
//          and byte counts), a scatter pass writes the index list.
//
// The path is pure data movement, so the design target is the HBM roofline, not MFMA.  Two families:
//
//   row-per-lane (pack_rows_kernel, the default for 4- and 8-byte elements): lane i loads row i of a
//     source array and stores each chunk's M elements as one contiguous piece at row i of the chunk.  No
//     LDS, no barrier; every wave instruction covers one contiguous span of whole 128-byte lines (1 KiB
//     loads, 768-byte / 256-byte stores).  One launch per frame: blockIdx.y = source array, dense
//     same-type arrays ride along as 16-byte-per-lane copies.  Runs at the rate of a bare register float4
//     copy of the same bytes (the round-2 lab, profiles/r02_pack_lab*).
//   LDS-tiled (pack_tiles_kernel and its variants, round 1): a source tile (contiguous in memory) is
//     streamed into LDS with 16-byte-per-lane loads, re-packed / converted between LDS and registers and
//     streamed out with 16-byte-per-lane stores.  Takes what the row kernel does not: 1- and 2-byte
//     elements, integer <-> float conversions, gathers through a permutation, rows wider than 32 bytes;
//     the unpack of the read path is of this family.
//   * fields that read the same source array (position.xyz and the type id HOOMD keeps
//     in position.w) form one group: the array is fetched from HBM once.
//   * 64-wide wavefronts, 256-thread workgroups; source rows are read once and chunk rows written once:
//     non-temporal hints on both sides (each worth 4-6 % at 10 M particles).
//
// No reference counterpart exists (the reference has no device code); the
// outputs are pinned by oracle_pack_rows() in oracle/pgsd_oracle.c and by the byte layout
// of chunks in the golden files.
#include "pgsd_internal.hpp"
#include "pgsd_pack.hpp"
#include "pgsd_private.h"

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <vector>

namespace pgsd_amd
    {
#define PACK_THREADS 256

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------ device helpers

template<int SSZ> __device__ __forceinline__ uint64_t lds_load(const char* p)
    {
    if constexpr (SSZ == 1)
        return *(const uint8_t*)p;
    else if constexpr (SSZ == 2)
        return *(const uint16_t*)p;
    else if constexpr (SSZ == 4)
        return *(const uint32_t*)p;
    else
        return *(const uint64_t*)p;
    }

// one element, value semantics selected by `kind` (wave-uniform)
template<int SSZ, int DSZ> __device__ __forceinline__ uint64_t convert_elem(uint64_t raw, uint32_t kind)
    {
    switch (kind)
        {
        default:
        case PACK_BITS: // same width, narrowing, zero-extension, bitcast: low bytes
            return raw;
        case PACK_SEXT:
            {
            if constexpr (SSZ == 1)
                return (uint64_t)(int64_t)(int8_t)raw;
            else if constexpr (SSZ == 2)
                return (uint64_t)(int64_t)(int16_t)raw;
            else if constexpr (SSZ == 4)
                return (uint64_t)(int64_t)(int32_t)raw;
            else
                return raw;
            }
        case PACK_F2F:
            {
            if constexpr (SSZ == 8 && DSZ == 4)
                return (uint64_t)__float_as_uint((float)__longlong_as_double((long long)raw)); // RNE
            else if constexpr (SSZ == 4 && DSZ == 8)
                return (uint64_t)__double_as_longlong((double)__uint_as_float((uint32_t)raw));
            else
                return raw;
            }
        case PACK_U2F:
            {
            if constexpr (DSZ == 4)
                return (uint64_t)__float_as_uint((float)(uint32_t)raw);
            else
                return (uint64_t)__double_as_longlong((double)(uint32_t)raw);
            }
        case PACK_S2F:
            {
            int32_t v;
            if constexpr (SSZ == 1)
                v = (int8_t)raw;
            else if constexpr (SSZ == 2)
                v = (int16_t)raw;
            else
                v = (int32_t)raw;
            if constexpr (DSZ == 4)
                return (uint64_t)__float_as_uint((float)v);
            else
                return (uint64_t)__double_as_longlong((double)v);
            }
        }
    }

// LDS image skew: 16 bytes of padding after every 128 bytes.  A staged float4 tile read back
// column-wise (position.xyz with stride 16/3 words, the w column with stride 16 words) hits the
// same few of the 32 banks: 6-way conflicts for xyz, 16-way for w in a linear image; with the
// skew every 8th row shifts by 4 banks and the worst cases drop to 2- and 4-way
// (SQ_LDS_BANK_CONFLICT, profiles/r01_lds_conflicts.md).
__device__ __forceinline__ uint32_t lds_skew(uint32_t byte_off)
    {
    return byte_off + ((byte_off >> 7) << 4);
    }

// source tiles are read once and chunk tiles written once: non-temporal on both sides.  (Round 1 swept
// LDS-DMA staging, default cache policies and a linear LDS image as compile-time variants: none was better,
// profiles/r01_pack_sweep.jsonl; the variants were removed in round 2.)
__device__ __forceinline__ u32x4 stream_load(const u32x4* p)
    {
    return __builtin_nontemporal_load(p);
    }

__device__ __forceinline__ void stream_store(u32x4 v, u32x4* p)
    {
    __builtin_nontemporal_store(v, p);
    }

// Stream the re-packed tile of one output chunk from LDS to global memory:
// 16 bytes per lane per store, lanes consecutive => each wave store covers 1 KiB.
template<int SSZ, int DSZ, int NT, int KIND = -1>
__device__ __forceinline__ void emit_tile(const PackOut& o, const char* lds, uint32_t rows,
                                          uint32_t stride_elems, uint64_t row0, uint32_t tid)
    {
    constexpr uint32_t EPT = 16 / DSZ; // elements per 16-byte vector
    const uint32_t M = o.M;
    const uint32_t nelem = rows * M;
    const uint32_t nvec = nelem / EPT;
    char* gdst = (char*)o.dst + row0 * (uint64_t)M * DSZ;
    const uint32_t kind = KIND >= 0 ? (uint32_t)KIND : o.kind;
    const uint32_t col0 = o.col0;

    for (uint32_t v = tid; v < nvec; v += NT)
        {
        uint32_t e = v * EPT;
        // row = e / M via multiply-high (exact for e < 2^32 / M, host guarantees it)
        uint32_t row = (M == 1) ? e : __umulhi(e, o.magic);
        uint32_t col = e - row * M;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t k = 0; k < EPT; k++)
            {
            uint64_t raw = lds_load<SSZ>(lds + lds_skew((row * stride_elems + col0 + col) * SSZ));
            uint64_t val = convert_elem<SSZ, DSZ>(raw, kind);
            if constexpr (DSZ == 8)
                {
                w[2 * k] = (uint32_t)val;
                w[2 * k + 1] = (uint32_t)(val >> 32);
                }
            else if constexpr (DSZ == 4)
                w[k] = (uint32_t)val;
            else if constexpr (DSZ == 2)
                w[k >> 1] |= ((uint32_t)val & 0xffffu) << (16 * (k & 1));
            else
                w[k >> 2] |= ((uint32_t)val & 0xffu) << (8 * (k & 3));
            if (++col == M)
                {
                col = 0;
                row++;
                }
            }
        u32x4 out = {w[0], w[1], w[2], w[3]};
        stream_store(out, (u32x4*)(gdst + (size_t)v * 16));
        }
    // ragged end of the last tile: element-wise
    for (uint32_t e = nvec * EPT + tid; e < nelem; e += NT)
        {
        uint32_t row = (M == 1) ? e : __umulhi(e, o.magic);
        uint32_t col = e - row * M;
        uint64_t raw = lds_load<SSZ>(lds + lds_skew((row * stride_elems + col0 + col) * SSZ));
        uint64_t val = convert_elem<SSZ, DSZ>(raw, kind);
        char* p = gdst + (size_t)e * DSZ;
        if constexpr (DSZ == 8)
            *(uint64_t*)p = val;
        else if constexpr (DSZ == 4)
            *(uint32_t*)p = (uint32_t)val;
        else if constexpr (DSZ == 2)
            *(uint16_t*)p = (uint16_t)val;
        else
            *(uint8_t*)p = (uint8_t)val;
        }
    }

template<int SSZ, int NT>
__device__ __forceinline__ void emit_dispatch(const PackOut& o, const char* lds, uint32_t rows,
                                              uint32_t stride_elems, uint64_t row0, uint32_t tid)
    {
    switch (o.dsz)
        {
        case 1: emit_tile<SSZ, 1, NT>(o, lds, rows, stride_elems, row0, tid); break;
        case 2: emit_tile<SSZ, 2, NT>(o, lds, rows, stride_elems, row0, tid); break;
        case 4: emit_tile<SSZ, 4, NT>(o, lds, rows, stride_elems, row0, tid); break;
        default: emit_tile<SSZ, 8, NT>(o, lds, rows, stride_elems, row0, tid); break;
        }
    }

template<int NT, int MODE>
__device__ __forceinline__ void emit_group(const PackGroup& g, const char* lds, uint32_t rows, uint64_t row0,
                                           uint32_t tid)
    {
    for (uint32_t oi = 0; oi < g.n_out; oi++)
        {
        const PackOut& o = g.out[oi];
        if constexpr (MODE == PACK_MODE_W32)
            {
            // 32-bit words moved unchanged (float4 -> N x 3 float, typeid in position.w, int3 images)
            emit_tile<4, 4, NT, PACK_BITS>(o, lds, rows, g.stride, row0, tid);
            continue;
            }
        if constexpr (MODE == PACK_MODE_F64_F32)
            {
            // double4 / double sources written as float32 chunks
            emit_tile<8, 4, NT, PACK_F2F>(o, lds, rows, g.stride, row0, tid);
            continue;
            }
        switch (g.ssz)
            {
            case 1: emit_dispatch<1, NT>(o, lds, rows, g.stride, row0, tid); break;
            case 2: emit_dispatch<2, NT>(o, lds, rows, g.stride, row0, tid); break;
            case 4: emit_dispatch<4, NT>(o, lds, rows, g.stride, row0, tid); break;
            default: emit_dispatch<8, NT>(o, lds, rows, g.stride, row0, tid); break;
            }
        }
    }

// Bring `rows` source rows starting at row0 into LDS with NT cooperating lanes:
// a linear 16-byte-per-lane stream, or a row gather through `order`.
template<int NT>
__device__ __forceinline__ void stage_rows(const PackGroup& g, char* lds, uint32_t rows, uint64_t row0, uint32_t tid)
    {
    const uint32_t rowbytes = g.rowbytes;
    if (g.order == nullptr)
        {
        // rows*rowbytes contiguous bytes, 16-byte aligned start
        const char* gsrc = (const char*)g.src + row0 * rowbytes;
        const uint32_t nbytes = rows * rowbytes;
        const uint32_t nvec = nbytes >> 4;
            {
            uint32_t v = tid;
            // four independent 16-byte loads in flight per lane
            for (; v + 3 * NT < nvec; v += 4 * NT)
                {
                u32x4 a = stream_load((const u32x4*)gsrc + v);
                u32x4 b = stream_load((const u32x4*)gsrc + v + NT);
                u32x4 c = stream_load((const u32x4*)gsrc + v + 2 * NT);
                u32x4 d = stream_load((const u32x4*)gsrc + v + 3 * NT);
                *(u32x4*)(lds + lds_skew(v << 4)) = a;
                *(u32x4*)(lds + lds_skew((v + NT) << 4)) = b;
                *(u32x4*)(lds + lds_skew((v + 2 * NT) << 4)) = c;
                *(u32x4*)(lds + lds_skew((v + 3 * NT) << 4)) = d;
                }
            for (; v < nvec; v += NT)
                *(u32x4*)(lds + lds_skew(v << 4)) = stream_load((const u32x4*)gsrc + v);
            }
        for (uint32_t b = (nvec << 4) + tid; b < nbytes; b += NT)
            lds[lds_skew(b)] = gsrc[b];
        }
    else
        {
        // row i of the tile comes from source row order[row0 + i]
        const uint32_t* ord = g.order + row0;
        if (rowbytes == 16)
            {
            // indices first, then four independent row fetches in flight per lane
            uint32_t i = tid;
            for (; i + 3 * NT < rows; i += 4 * NT)
                {
                const uint32_t o0 = ord[i], o1 = ord[i + NT], o2 = ord[i + 2 * NT], o3 = ord[i + 3 * NT];
                u32x4 a = *((const u32x4*)g.src + o0);
                u32x4 b = *((const u32x4*)g.src + o1);
                u32x4 c = *((const u32x4*)g.src + o2);
                u32x4 d = *((const u32x4*)g.src + o3);
                *(u32x4*)(lds + lds_skew(i << 4)) = a;
                *(u32x4*)(lds + lds_skew((i + NT) << 4)) = b;
                *(u32x4*)(lds + lds_skew((i + 2 * NT) << 4)) = c;
                *(u32x4*)(lds + lds_skew((i + 3 * NT) << 4)) = d;
                }
            for (; i < rows; i += NT)
                *(u32x4*)(lds + lds_skew(i << 4)) = *((const u32x4*)g.src + ord[i]);
            }
        else if (rowbytes == 32)
            {
            uint32_t i = tid;
            for (; i + NT < rows; i += 2 * NT)
                {
                const uint64_t o0 = ord[i], o1 = ord[i + NT];
                u32x4 a0 = *((const u32x4*)g.src + 2 * o0), a1 = *((const u32x4*)g.src + 2 * o0 + 1);
                u32x4 b0 = *((const u32x4*)g.src + 2 * o1), b1 = *((const u32x4*)g.src + 2 * o1 + 1);
                *(u32x4*)(lds + lds_skew((2 * i) << 4)) = a0;
                *(u32x4*)(lds + lds_skew((2 * i + 1) << 4)) = a1;
                *(u32x4*)(lds + lds_skew((2 * (i + NT)) << 4)) = b0;
                *(u32x4*)(lds + lds_skew((2 * (i + NT) + 1) << 4)) = b1;
                }
            for (; i < rows; i += NT)
                {
                const uint64_t o0 = ord[i];
                *(u32x4*)(lds + lds_skew((2 * i) << 4)) = *((const u32x4*)g.src + 2 * o0);
                *(u32x4*)(lds + lds_skew((2 * i + 1) << 4)) = *((const u32x4*)g.src + 2 * o0 + 1);
                }
            }
        else if ((rowbytes & 3) == 0)
            {
            const uint32_t wpr = rowbytes >> 2;
            const uint32_t nw = rows * wpr;
            for (uint32_t i = tid; i < nw; i += NT)
                {
                uint32_t r = i / wpr, c = i - r * wpr;
                *(uint32_t*)(lds + lds_skew(i << 2)) = *((const uint32_t*)g.src + (uint64_t)ord[r] * wpr + c);
                }
            }
        else
            {
            const uint32_t nb = rows * rowbytes;
            for (uint32_t i = tid; i < nb; i += NT)
                {
                uint32_t r = i / rowbytes, c = i - r * rowbytes;
                lds[lds_skew(i)] = *((const char*)g.src + (uint64_t)ord[r] * rowbytes + c);
                }
            }
        }
    }

// Fused multi-field pack, workgroup-tiled kernel.  One 256-thread workgroup (four 64-lane
// wavefronts) owns a tile of `tile_rows` particles at a time; tiles are dealt to
// workgroups round-robin, so the workgroups resident at any moment stream one contiguous
// window of every array (DRAM-page friendly).  Per tile and source group: stage -> barrier
// -> emit every chunk fed by that source -> barrier.
template<int MODE> __global__ __launch_bounds__(PACK_THREADS) void pack_tiles_kernel(const PackArgs args)
    {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t TILE = args.tile_rows;

    for (uint64_t tile = blockIdx.x; tile < args.n_tiles; tile += gridDim.x)
        {
        const uint64_t row0 = tile * TILE;
        const uint32_t rows = (uint32_t)((args.N - row0 < (uint64_t)TILE) ? args.N - row0 : TILE);
        // every source array of a batch is in flight before the first byte is consumed:
        // one load latency and two barriers per batch instead of per source array
        for (uint32_t b = 0; b < args.n_batches; b++)
            {
            const uint32_t g0 = args.batch_start[b], g1 = args.batch_start[b + 1];
            for (uint32_t gi = g0; gi < g1; gi++)
                stage_rows<PACK_THREADS>(args.g[gi], lds + args.g[gi].lds_off, rows, row0, tid);
            __syncthreads();
            for (uint32_t gi = g0; gi < g1; gi++)
                emit_group<PACK_THREADS, MODE>(args.g[gi], lds + args.g[gi].lds_off, rows, row0, tid);
            __syncthreads();
            }
        }
    }

// Software-pipelined form of the tiled kernel for launches with at most PF_GROUPS linear source
// arrays whose tiles are at most PF_VECS x 256 vectors: the 16-byte loads of the NEXT tile are
// issued into registers before the current tile is emitted, so that a workgroup has loads in
// flight while it stores.  What it buys is the overlap of the read and the write phase when a
// workgroup only sees one or two tiles (2^20 particles: every tile is resident at once and the
// plain kernel runs "all loads, then all stores").
#define PF_GROUPS 3
#define PF_VECS 4
template<int MODE> __global__ __launch_bounds__(PACK_THREADS) void pack_tiles_prefetch_kernel(const PackArgs args)
    {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t TILE = args.tile_rows;
    u32x4 r[PF_GROUPS * PF_VECS];

    auto issue = [&](uint64_t tile)
    {
        const uint64_t row0 = tile * TILE;
        const uint32_t rows = (uint32_t)((args.N - row0 < (uint64_t)TILE) ? args.N - row0 : TILE);
#pragma unroll
        for (uint32_t gi = 0; gi < PF_GROUPS; gi++)
            {
            if (gi >= args.n_groups)
                break;
            const PackGroup& g = args.g[gi];
            const u32x4* gsrc = (const u32x4*)((const char*)g.src + row0 * g.rowbytes);
            const uint32_t nvec = (rows * g.rowbytes) >> 4;
#pragma unroll
            for (uint32_t k = 0; k < PF_VECS; k++)
                {
                const uint32_t v = tid + k * PACK_THREADS;
                if (v < nvec)
                    r[gi * PF_VECS + k] = __builtin_nontemporal_load(gsrc + v);
                }
            }
    };
    auto commit = [&](uint64_t tile)
    {
        const uint64_t row0 = tile * TILE;
        const uint32_t rows = (uint32_t)((args.N - row0 < (uint64_t)TILE) ? args.N - row0 : TILE);
#pragma unroll
        for (uint32_t gi = 0; gi < PF_GROUPS; gi++)
            {
            if (gi >= args.n_groups)
                break;
            const PackGroup& g = args.g[gi];
            char* l = lds + g.lds_off;
            const uint32_t nbytes = rows * g.rowbytes;
            const uint32_t nvec = nbytes >> 4;
#pragma unroll
            for (uint32_t k = 0; k < PF_VECS; k++)
                {
                const uint32_t v = tid + k * PACK_THREADS;
                if (v < nvec)
                    *(u32x4*)(l + lds_skew(v << 4)) = r[gi * PF_VECS + k];
                }
            // ragged end of the last tile
            const char* gsrc = (const char*)g.src + row0 * g.rowbytes;
            for (uint32_t b = (nvec << 4) + tid; b < nbytes; b += PACK_THREADS)
                l[lds_skew(b)] = gsrc[b];
            }
    };

    uint64_t tile = blockIdx.x;
    if (tile < args.n_tiles)
        issue(tile);
    for (; tile < args.n_tiles; tile += gridDim.x)
        {
        const uint64_t row0 = tile * TILE;
        const uint32_t rows = (uint32_t)((args.N - row0 < (uint64_t)TILE) ? args.N - row0 : TILE);
        commit(tile);
        __syncthreads();
        if (tile + gridDim.x < args.n_tiles)
            issue(tile + gridDim.x);
        for (uint32_t gi = 0; gi < args.n_groups; gi++)
            emit_group<PACK_THREADS, MODE>(args.g[gi], lds + args.g[gi].lds_off, rows, row0, tid);
        __syncthreads();
        }
    }

// ------------------------------------------------------------------ row-per-lane pack (the default)
// One particle row per lane, no LDS, no barrier: lane i loads row i of a source array with one
// (or two) vector loads and stores the M elements of every chunk fed by that array as ONE
// contiguous piece of M * sizeof(element) bytes at row i of the chunk.  Consecutive lanes hold
// consecutive rows, so every wave instruction still covers one contiguous span of memory --
// 1 KiB per float4 load, 768 B per N x 3 float store, 256 B per scalar store, all of them whole
// 128-byte lines -- which is what the memory system needs; the 16-byte-per-lane rule is not.
// Measured on MI355X (the round-2 lab, profiles/r02_pack_lab*.jsonl): this shape runs the
// headline workload at the rate of a bare register float4 copy of the same bytes (93-95 us for
// 600 MB at 10 M particles), where the LDS-staged tile kernel below needs 104-107 us: the
// stage -> barrier -> emit phases cost more than the odd store widths.
// Workgroups are dealt to source arrays group-major (all workgroups of array 0, then array 1 ...):
// a wave keeps one input and at most a few output streams open.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x3 __attribute__((ext_vector_type(3)));
// rows of 12 or 24 bytes are only dword / 8-byte aligned
typedef u32x2 u32x2_a4 __attribute__((aligned(4)));
typedef u32x3 u32x3_a4 __attribute__((aligned(4)));
typedef u32x4 u32x4_a4 __attribute__((aligned(4)));

// a source row in registers: two SSA vectors (a C array indexed by a run-time column would be
// demoted to scratch memory by the compiler)
struct RowRegs
    {
    u32x4 lo, hi;
    };

template<int RW> __device__ __forceinline__ void row_load(const uint32_t* p, RowRegs& r)
    {
    if constexpr (RW == 1)
        r.lo.x = __builtin_nontemporal_load(p);
    else if constexpr (RW == 2)
        {
        u32x2 v = __builtin_nontemporal_load((const u32x2_a4*)p);
        r.lo.x = v.x, r.lo.y = v.y;
        }
    else if constexpr (RW == 3)
        {
        u32x3 v = __builtin_nontemporal_load((const u32x3_a4*)p);
        r.lo.x = v.x, r.lo.y = v.y, r.lo.z = v.z;
        }
    else
        {
        r.lo = __builtin_nontemporal_load((const u32x4_a4*)p);
        if constexpr (RW == 6)
            {
            u32x2 w = __builtin_nontemporal_load((const u32x2_a4*)(p + 4));
            r.hi.x = w.x, r.hi.y = w.y;
            }
        else if constexpr (RW == 8)
            r.hi = __builtin_nontemporal_load((const u32x4_a4*)(p + 4));
        }
    }

// dword `i` (wave-uniform) of a row held in registers
template<int RW> __device__ __forceinline__ uint32_t row_pick(const RowRegs& r, uint32_t i)
    {
    if constexpr (RW == 1)
        return r.lo.x;
    else if constexpr (RW == 2)
        return (i & 1u) ? r.lo.y : r.lo.x;
    else if constexpr (RW <= 4)
        {
        const uint32_t a = (i & 1u) ? r.lo.y : r.lo.x, b = (i & 1u) ? r.lo.w : r.lo.z;
        return (i & 2u) ? b : a;
        }
    else
        {
        const uint32_t a = (i & 1u) ? r.lo.y : r.lo.x, b = (i & 1u) ? r.lo.w : r.lo.z;
        const uint32_t c = (i & 1u) ? r.hi.y : r.hi.x, d = (i & 1u) ? r.hi.w : r.hi.z;
        const uint32_t ab = (i & 2u) ? b : a, cd = (i & 2u) ? d : c;
        return (i & 4u) ? cd : ab;
        }
    }

// nw (1..8, wave-uniform) dwords to row `i` of a chunk whose rows are nw dwords long
template<uint32_t NWMAX>
__device__ __forceinline__ void row_store(uint32_t* p, const uint32_t (&w)[ROWS_MAX_WORDS], uint32_t nw)
    {
    if (NWMAX >= 4 && nw >= 4)
        {
        u32x4 v = {w[0], w[1], w[2], w[3]};
        __builtin_nontemporal_store(v, (u32x4_a4*)p);
        if (NWMAX == 4)
            return;
        if (nw == 8)
            {
            u32x4 q = {w[4], w[5], w[6], w[7]};
            __builtin_nontemporal_store(q, (u32x4_a4*)(p + 4));
            }
        else if (nw == 6)
            {
            u32x2 q = {w[4], w[5]};
            __builtin_nontemporal_store(q, (u32x2_a4*)(p + 4));
            }
        else if (nw == 5)
            __builtin_nontemporal_store(w[4], p + 4);
        else if (nw == 7)
            {
            u32x3 q = {w[4], w[5], w[6]};
            __builtin_nontemporal_store(q, (u32x3_a4*)(p + 4));
            }
        }
    else if (nw == 3)
        {
        u32x3 v = {w[0], w[1], w[2]};
        __builtin_nontemporal_store(v, (u32x3_a4*)p);
        }
    else if (nw == 2)
        {
        u32x2 v = {w[0], w[1]};
        __builtin_nontemporal_store(v, (u32x2_a4*)p);
        }
    else
        __builtin_nontemporal_store(w[0], p);
    }

// BITS_ONLY: every chunk of the launch moves dwords unchanged (the common case: float4 -> N x 3
// floats, the type id in position.w, int3 images); no conversion code is generated
template<int RW, bool BITS_ONLY>
__device__ __forceinline__ void row_emit(const RowsOut& o, const RowRegs& r, uint64_t i)
    {
    uint32_t w[ROWS_MAX_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0};
    const uint32_t nw = o.nw_out, c0 = o.col0; // c0: first source dword
    constexpr uint32_t NWMAX = BITS_ONLY ? (RW < ROWS_MAX_WORDS ? RW : ROWS_MAX_WORDS) : ROWS_MAX_WORDS;
    if (BITS_ONLY || o.kind == ROWS_BITS)
        {
        // picks past the chunk's width are computed and dropped: cheaper than guarding them
#pragma unroll
        for (uint32_t k = 0; k < NWMAX; k++)
            w[k] = row_pick<RW>(r, c0 + k);
        }
    else if (o.kind == ROWS_F64_F32)
        {
        if constexpr (RW >= 2)
            {
#pragma unroll
            for (uint32_t k = 0; k < RW / 2; k++)
                {
                const uint64_t bits = (uint64_t)row_pick<RW>(r, c0 + 2 * k) | ((uint64_t)row_pick<RW>(r, c0 + 2 * k + 1) << 32);
                w[k] = __float_as_uint((float)__longlong_as_double((long long)bits)); // v_cvt_f32_f64, RNE
                }
            }
        }
    else // ROWS_F32_F64
        {
#pragma unroll
        for (uint32_t k = 0; k < (RW < 4 ? RW : 4); k++)
            {
            const uint64_t bits = (uint64_t)__double_as_longlong((double)__uint_as_float(row_pick<RW>(r, c0 + k)));
            w[2 * k] = (uint32_t)bits;
            w[2 * k + 1] = (uint32_t)(bits >> 32);
            }
        }
    row_store<NWMAX>((uint32_t*)o.dst + i * nw, w, nw);
    }

// Dense chunks of the source's own type (the chunk IS the array: scalar arrays, orientation, ...): 16 bytes
// per lane.  The gridDim.x workgroups of the array share its vectors evenly, one contiguous slice each,
// whatever the array's length relative to the row-mode arrays of the same launch.
template<int T, int U> __device__ __forceinline__ void copy_body(const RowsGroup& g)
    {
    const u32x4* src = (const u32x4*)g.src;
    u32x4* dst = (u32x4*)g.out[0].dst;
    const uint64_t nvec = g.copy_vecs;
    const uint64_t per = (nvec + gridDim.x - 1) / gridDim.x;
    const uint64_t first = (uint64_t)blockIdx.x * per;
    const uint64_t last = first + per < nvec ? first + per : nvec;
    for (uint64_t base = first + threadIdx.x; base < last; base += (uint64_t)(T * U))
        {
        u32x4 r[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t v = base + (uint64_t)k * T;
            if (v < last)
                r[k] = __builtin_nontemporal_load(src + v);
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t v = base + (uint64_t)k * T;
            if (v < last)
                __builtin_nontemporal_store(r[k], dst + v);
            }
        }
    if (blockIdx.x == 0 && threadIdx.x < g.copy_tail)
        ((char*)dst)[nvec * 16 + threadIdx.x] = ((const char*)src)[nvec * 16 + threadIdx.x];
    }

// blockIdx.y = source array (group), blockIdx.x = block of T*U consecutive rows (or 16-byte vectors of a
// dense array riding along in the same launch).  The dispatcher walks x fastest, so the launch streams
// array after array (group-major); ONE launch per frame keeps small snapshots from paying a second
// launch's ramp (2^20 particles with a separate id array: 16.5 us as two launches, see profiles/).
template<int T, int U, int RW, bool BITS_ONLY>
__global__ __launch_bounds__(T) void pack_rows_kernel(const RowsArgs args)
    {
    const RowsGroup& g = args.g[blockIdx.y];
    if (g.copy_vecs != 0 || g.copy_tail != 0)
        {
        copy_body<T, U>(g);
        return;
        }
    RowRegs r[U];
    const uint64_t N = args.N;
    const uint64_t base = (uint64_t)blockIdx.x * (uint64_t)(T * U) + threadIdx.x;
    const uint32_t* src = (const uint32_t*)g.src;
#pragma unroll
    for (int k = 0; k < U; k++)
        {
        r[k].lo = u32x4 {0, 0, 0, 0};
        r[k].hi = u32x4 {0, 0, 0, 0};
        const uint64_t i = base + (uint64_t)k * T;
        if (i < N)
            row_load<RW>(src + i * RW, r[k]);
        }
    const uint32_t n_out = g.n_out;
#pragma unroll
    for (int k = 0; k < U; k++)
        {
        const uint64_t i = base + (uint64_t)k * T;
        if (i < N)
            for (uint32_t oi = 0; oi < n_out; oi++)
                row_emit<RW, BITS_ONLY>(g.out[oi], r[k], i);
        }
    }

// a launch of dense arrays only
template<int T, int U> __global__ __launch_bounds__(T) void pack_copy_kernel(const RowsArgs args)
    {
    copy_body<T, U>(args.g[blockIdx.y]);
    }

// Fallback for operands the tiled kernel cannot take (unaligned pointers, very wide
// rows): one output element per lane, grid-stride.  Correct for everything, not tuned.
__global__ __launch_bounds__(PACK_THREADS) void pack_generic_kernel(const PackGenericArgs a)
    {
    const uint64_t total = a.N * a.M;
    for (uint64_t e = (uint64_t)blockIdx.x * PACK_THREADS + threadIdx.x; e < total;
         e += (uint64_t)gridDim.x * PACK_THREADS)
        {
        uint64_t row = e / a.M;
        uint32_t col = (uint32_t)(e - row * a.M);
        uint64_t srow = a.order ? a.order[row] : row;
        const char* sp = (const char*)a.src + (srow * a.stride + a.col0 + col) * a.ssz;
        uint64_t raw = 0;
        for (uint32_t b = 0; b < a.ssz; b++)
            raw |= (uint64_t)(uint8_t)sp[b] << (8 * b);
        uint64_t val;
        switch (a.ssz)
            {
            case 1: val = a.dsz == 8 ? convert_elem<1, 8>(raw, a.kind) : convert_elem<1, 4>(raw, a.kind); break;
            case 2: val = a.dsz == 8 ? convert_elem<2, 8>(raw, a.kind) : convert_elem<2, 4>(raw, a.kind); break;
            case 4: val = a.dsz == 8 ? convert_elem<4, 8>(raw, a.kind) : convert_elem<4, 4>(raw, a.kind); break;
            default: val = a.dsz == 4 ? convert_elem<8, 4>(raw, a.kind) : convert_elem<8, 8>(raw, a.kind); break;
            }
        char* dp = (char*)a.dst + e * a.dsz;
        for (uint32_t b = 0; b < a.dsz; b++)
            dp[b] = (char)(val >> (8 * b));
        }
    }

// ------------------------------------------------------------------ unpack (read path)
// Inverse of the pack.  The dense tiles of ALL chunks of a launch (contiguous bytes) are
// streamed into LDS with 16-byte non-temporal loads, one barrier, then
//   * destination arrays whose rows are completely restored by this launch (position.xyz and
//     the type id into a Scalar4 array, velocity.xyz and mass, double4 builds ...) are
//     assembled row-wise in registers and written with 16-byte non-temporal stores: whole
//     lines, no partial-line writes;
//   * every other chunk is scattered element-wise: lane e converts element e of the tile and
//     stores it to its (row, column); columns no chunk restores are left untouched.
// W32 = every chunk and destination element is 4 bytes wide and moved unchanged.
template<bool W32> __device__ __forceinline__ uint64_t unpack_elem(const char* p, uint32_t ssz, uint32_t dsz, uint32_t kind)
    {
    if constexpr (W32)
        return *(const uint32_t*)p;
    else
        {
        switch (ssz)
            {
            case 1: return dsz == 8 ? convert_elem<1, 8>(lds_load<1>(p), kind) : convert_elem<1, 4>(lds_load<1>(p), kind);
            case 2: return dsz == 8 ? convert_elem<2, 8>(lds_load<2>(p), kind) : convert_elem<2, 4>(lds_load<2>(p), kind);
            case 4: return dsz == 8 ? convert_elem<4, 8>(lds_load<4>(p), kind) : convert_elem<4, 4>(lds_load<4>(p), kind);
            default: return dsz == 4 ? convert_elem<8, 4>(lds_load<8>(p), kind) : convert_elem<8, 8>(lds_load<8>(p), kind);
            }
        }
    }

template<bool W32>
__device__ __forceinline__ void scatter_tile(const UnpackJob& j, const char* lds, uint32_t rows, uint64_t row0)
    {
    const uint32_t M = j.M, ssz = W32 ? 4u : j.ssz, dsz = W32 ? 4u : j.dsz;
    const uint32_t nelem = rows * M;
    for (uint32_t e = threadIdx.x; e < nelem; e += PACK_THREADS)
        {
        uint32_t row = (M == 1) ? e : __umulhi(e, j.magic);
        uint32_t col = e - row * M;
        uint64_t val = unpack_elem<W32>(lds + (size_t)e * ssz, ssz, dsz, j.kind);
        uint64_t drow = j.order ? (uint64_t)j.order[row0 + row] : row0 + row;
        char* p = (char*)j.dst + (drow * j.dst_stride + j.dst_col0 + col) * dsz;
        if (dsz == 8)
            *(uint64_t*)p = val;
        else if (dsz == 4)
            *(uint32_t*)p = (uint32_t)val;
        else if (dsz == 2)
            *(uint16_t*)p = (uint16_t)val;
        else
            *(uint8_t*)p = (uint8_t)val;
        }
    }

// Where one column of an assembled destination row comes from (kept in LDS: lanes that build
// different vectors of a wide row look up different columns).
struct UnpackCol
    {
    uint32_t base; // LDS byte offset of column 0 .. of row 0 of the chunk tile
    uint32_t step; // bytes per chunk row
    uint32_t ssz, kind;
    };
#define UNPACK_TABLE_BYTES (UNPACK_MAX_GROUPS * UNPACK_MAX_ROW_COLS * 16)

// one 16-byte vector of a destination row per lane per step; rows are 16, 32 or 64 bytes
template<bool W32>
__device__ __forceinline__ void assemble_rows(const UnpackGroup& g, const UnpackCol* tab, const char* lds, uint32_t rows,
                                              uint64_t row0)
    {
    const uint32_t shift = g.vec_shift, mask = (1u << shift) - 1u;
    const uint32_t dsz = W32 ? 4u : g.dsz;
    const uint32_t ept = 16u / dsz; // 4 or 2 elements per vector
    const uint32_t nvec = rows << shift;
    const uint32_t rowbytes = g.stride * dsz;
    const uint32_t* order = g.order;
    char* dst = (char*)g.dst;
    if (shift == 0)
        {
        // 16-byte rows (Scalar4 of floats / ints): the four column descriptors are loop invariants
        UnpackCol d[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            d[k] = tab[k < ept ? k : 0];
        for (uint32_t row = threadIdx.x; row < rows; row += PACK_THREADS)
            {
            uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
            for (uint32_t k = 0; k < 4; k++)
                {
                if (k >= ept)
                    break;
                uint64_t val = unpack_elem<W32>(lds + d[k].base + row * d[k].step, d[k].ssz, dsz, d[k].kind);
                if (dsz == 8)
                    {
                    w[2 * k] = (uint32_t)val;
                    w[2 * k + 1] = (uint32_t)(val >> 32);
                    }
                else
                    w[k] = (uint32_t)val;
                }
            const uint64_t drow = order ? (uint64_t)order[row0 + row] : row0 + row;
            u32x4 out = {w[0], w[1], w[2], w[3]};
            __builtin_nontemporal_store(out, (u32x4*)(dst + drow * 16));
            }
        return;
        }
    for (uint32_t v = threadIdx.x; v < nvec; v += PACK_THREADS)
        {
        const uint32_t row = v >> shift, q = v & mask;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (uint32_t k = 0; k < 4; k++)
            {
            if (k >= ept)
                break;
            const UnpackCol d = tab[q * ept + k];
            uint64_t val = unpack_elem<W32>(lds + d.base + row * d.step, d.ssz, dsz, d.kind);
            if (dsz == 8)
                {
                w[2 * k] = (uint32_t)val;
                w[2 * k + 1] = (uint32_t)(val >> 32);
                }
            else
                w[k] = (uint32_t)val;
            }
        const uint64_t drow = order ? (uint64_t)order[row0 + row] : row0 + row;
        u32x4 out = {w[0], w[1], w[2], w[3]};
        __builtin_nontemporal_store(out, (u32x4*)(dst + drow * rowbytes + (size_t)q * 16));
        }
    }

template<bool W32> __global__ __launch_bounds__(PACK_THREADS) void unpack_tiles_kernel(const UnpackArgs args)
    {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const uint32_t tid = threadIdx.x;
    const uint32_t TILE = args.tile_rows;
    // column table of the assembled arrays, once per workgroup (chunk tiles start behind it)
    UnpackCol* tab = (UnpackCol*)lds;
    if (tid < args.n_groups * UNPACK_MAX_ROW_COLS)
        {
        const UnpackGroup& g = args.g[tid / UNPACK_MAX_ROW_COLS];
        const uint32_t col = tid % UNPACK_MAX_ROW_COLS;
        UnpackCol d = {0, 0, 4, 0};
        if (col < g.stride)
            {
            const UnpackJob& j = args.j[g.col_job[col]];
            d.base = j.lds_off + g.col_off[col] * j.ssz;
            d.step = j.rowbytes;
            d.ssz = j.ssz;
            d.kind = j.kind;
            }
        tab[tid] = d;
        }
    // (the first tile's barrier publishes the table)
    for (uint64_t tile = blockIdx.x; tile < args.n_tiles; tile += gridDim.x)
        {
        const uint64_t row0 = tile * TILE;
        const uint32_t rows = (uint32_t)((args.N - row0 < (uint64_t)TILE) ? args.N - row0 : TILE);
        // every chunk tile in flight before the first byte is consumed
        for (uint32_t ji = 0; ji < args.n_jobs; ji++)
            {
            const UnpackJob& j = args.j[ji];
            const char* gsrc = (const char*)j.src + row0 * j.rowbytes;
            char* l = lds + j.lds_off;
            const uint32_t nbytes = rows * j.rowbytes;
            const uint32_t nvec = nbytes >> 4;
            uint32_t v = tid;
            for (; v + 3 * PACK_THREADS < nvec; v += 4 * PACK_THREADS)
                {
                u32x4 a = __builtin_nontemporal_load((const u32x4*)gsrc + v);
                u32x4 b = __builtin_nontemporal_load((const u32x4*)gsrc + v + PACK_THREADS);
                u32x4 c = __builtin_nontemporal_load((const u32x4*)gsrc + v + 2 * PACK_THREADS);
                u32x4 d = __builtin_nontemporal_load((const u32x4*)gsrc + v + 3 * PACK_THREADS);
                ((u32x4*)l)[v] = a;
                ((u32x4*)l)[v + PACK_THREADS] = b;
                ((u32x4*)l)[v + 2 * PACK_THREADS] = c;
                ((u32x4*)l)[v + 3 * PACK_THREADS] = d;
                }
            for (; v < nvec; v += PACK_THREADS)
                ((u32x4*)l)[v] = __builtin_nontemporal_load((const u32x4*)gsrc + v);
            for (uint32_t b = (nvec << 4) + tid; b < nbytes; b += PACK_THREADS)
                l[b] = gsrc[b];
            }
        __syncthreads();
        for (uint32_t gi = 0; gi < args.n_groups; gi++)
            assemble_rows<W32>(args.g[gi], tab + gi * UNPACK_MAX_ROW_COLS, lds, rows, row0);
        for (uint32_t ji = 0; ji < args.n_jobs; ji++)
            if (!args.j[ji].in_group)
                scatter_tile<W32>(args.j[ji], lds + args.j[ji].lds_off, rows, row0);
        __syncthreads();
        }
    }

// ------------------------------------------------------------------ row-per-lane unpack (Scalar4 destinations)
// Inverse of pack_rows_kernel for the restart path's common shape: a float4-like destination array fed by
// one or two dense chunks (position.xyz + the type id into position.w, velocity.xyz + mass).  Lane i loads
// row i of each chunk (12 bytes + 4 bytes: contiguous pieces, consecutive lanes consecutive rows), and
// stores ONE 16-byte row (two for a double4 destination restored from f32 chunks).  No LDS, no barrier;
// blockIdx.y = destination array; dense same-type arrays ride along as 16-byte copies.  Measured in the lab
// (the round-2 lab, profiles/r02_lab_unpack.jsonl): 100 us for position + id + velocity + mass of 10 M
// particles where the LDS-tiled unpack needs 113 us.  A first, fully generic version of this kernel (run-time
// chunk lists and widths) was no faster than the tiled one; the static hot path is what pays.
template<bool F64> __device__ __forceinline__ void unrows_store(uint32_t* drow, const uint32_t* w, uint32_t nw, uint32_t col0)
    {
    // nw source dwords (f32 / 32-bit integers) -> destination elements col0 .. col0+nw
    if constexpr (!F64)
        {
        uint32_t c[ROWS_MAX_WORDS] = {w[0], w[1], w[2], w[3], 0, 0, 0, 0};
        row_store<4>(drow + col0, c, nw);
        }
    else
        {
        uint32_t c[ROWS_MAX_WORDS];
#pragma unroll
        for (uint32_t e = 0; e < 4; e++)
            {
            const uint64_t bits = (uint64_t)__double_as_longlong((double)__uint_as_float(w[e]));
            c[2 * e] = (uint32_t)bits;
            c[2 * e + 1] = (uint32_t)(bits >> 32);
            }
        row_store<8>(drow + 2 * col0, c, 2 * nw);
        }
    }

__device__ __forceinline__ void unrows_load(const uint32_t* p, uint32_t nw, uint32_t (&w)[4])
    {
    RowRegs r;
    r.lo = u32x4 {0, 0, 0, 0};
    switch (nw)
        {
        case 1: row_load<1>(p, r); break;
        case 2: row_load<2>(p, r); break;
        case 3: row_load<3>(p, r); break;
        default: row_load<4>(p, r); break;
        }
    w[0] = r.lo.x, w[1] = r.lo.y, w[2] = r.lo.z, w[3] = r.lo.w;
    }

// One WHOLE destination row: column e takes v[e] (f32 / 32-bit bits from a chunk) unless bit e of fillmask is
// set, then the fill element.  Compile-time loops only (a run-time index into c[] would go to scratch).
template<bool F64>
__device__ __forceinline__ void unrows_store_whole(uint32_t* drow, const uint32_t (&v)[4], uint32_t fillmask, uint32_t fill_lo,
                                                   uint32_t fill_hi)
    {
    if constexpr (!F64)
        {
        uint32_t c[ROWS_MAX_WORDS] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (uint32_t e = 0; e < 4; e++)
            c[e] = ((fillmask >> e) & 1u) ? fill_lo : v[e];
        row_store<4>(drow, c, 4);
        }
    else
        {
        uint32_t c[ROWS_MAX_WORDS];
#pragma unroll
        for (uint32_t e = 0; e < 4; e++)
            {
            const uint64_t bits = (uint64_t)__double_as_longlong((double)__uint_as_float(v[e]));
            const bool f = (fillmask >> e) & 1u;
            c[2 * e] = f ? fill_lo : (uint32_t)bits;
            c[2 * e + 1] = f ? fill_hi : (uint32_t)(bits >> 32);
            }
        row_store<8>(drow, c, 8);
        }
    }

// v[col0 + s] = w[s] for s < nw, without run-time indexing
__device__ __forceinline__ void unrows_place(uint32_t (&v)[4], const uint32_t (&w)[4], uint32_t nw, uint32_t col0)
    {
#pragma unroll
    for (uint32_t e = 0; e < 4; e++)
#pragma unroll
        for (uint32_t s = 0; s < 4; s++)
            if (s < nw && col0 + s == e)
                v[e] = w[s];
    }

template<int T, int U, bool F64> __global__ __launch_bounds__(T) void unpack_rows_kernel(const UnrowsArgs args)
    {
    const UnrowsGroup& g = args.g[blockIdx.y];
    if (g.copy_vecs != 0 || g.copy_tail != 0)
        {
        // dense same-type array: the chunk IS the array
        const u32x4* src = (const u32x4*)g.a;
        u32x4* dst = (u32x4*)g.dst;
        const uint64_t nvec = g.copy_vecs;
        const uint64_t per = (nvec + gridDim.x - 1) / gridDim.x;
        const uint64_t first = (uint64_t)blockIdx.x * per;
        const uint64_t last = first + per < nvec ? first + per : nvec;
        for (uint64_t v = first + threadIdx.x; v < last; v += T)
            __builtin_nontemporal_store(__builtin_nontemporal_load(src + v), dst + v);
        if (blockIdx.x == 0 && threadIdx.x < g.copy_tail)
            ((char*)dst)[nvec * 16 + threadIdx.x] = ((const char*)src)[nvec * 16 + threadIdx.x];
        return;
        }
    const uint64_t N = args.N;
    const uint64_t base = (uint64_t)blockIdx.x * (uint64_t)(T * U) + threadIdx.x;
    constexpr uint32_t DW = F64 ? 8 : 4;
    uint32_t* dst = (uint32_t*)g.dst;
    const uint32_t* a = (const uint32_t*)g.a;
    const uint32_t* b = (const uint32_t*)g.b;
    if (g.a_nw == 3 && g.a_col0 == 0 && b == nullptr && g.fill_on)
        {
        // xyz from a chunk, w = a constant (velocity without a mass chunk -> {vx, vy, vz, 1.0f}): whole rows out
        u32x3 xyz[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * T;
            if (i < N)
                xyz[k] = __builtin_nontemporal_load((const u32x3_a4*)(a + i * 3));
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * T;
            if (i < N)
                {
                const uint32_t c[4] = {xyz[k].x, xyz[k].y, xyz[k].z, 0};
                unrows_store_whole<F64>(dst + i * DW, c, 8u, g.fill_lo, g.fill_hi);
                }
            }
        return;
        }
    if (g.fill_on)
        {
        // any other one- or two-chunk shape with a fill: compose the whole row, one store
        uint32_t mask = 15u;
#pragma unroll
        for (uint32_t s = 0; s < 4; s++)
            {
            if (s < g.a_nw)
                mask &= ~(1u << (g.a_col0 + s));
            if (b != nullptr && s < g.b_nw)
                mask &= ~(1u << (g.b_col0 + s));
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * T;
            if (i >= N)
                continue;
            uint32_t wa[4], wb[4] = {0, 0, 0, 0}, v[4] = {0, 0, 0, 0};
            unrows_load(a + i * g.a_nw, g.a_nw, wa);
            unrows_place(v, wa, g.a_nw, g.a_col0);
            if (b != nullptr)
                {
                unrows_load(b + i * g.b_nw, g.b_nw, wb);
                unrows_place(v, wb, g.b_nw, g.b_col0);
                }
            unrows_store_whole<F64>(dst + i * DW, v, mask, g.fill_lo, g.fill_hi);
            }
        return;
        }
    if (g.a_nw == 3 && g.a_col0 == 0 && b != nullptr && g.b_nw == 1 && g.b_col0 == 3)
        {
        // the hot shape: xyz from one chunk, w from another, whole rows out
        u32x3 xyz[U];
        uint32_t w[U];
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * T;
            if (i < N)
                {
                xyz[k] = __builtin_nontemporal_load((const u32x3_a4*)(a + i * 3));
                w[k] = __builtin_nontemporal_load(b + i);
                }
            }
#pragma unroll
        for (int k = 0; k < U; k++)
            {
            const uint64_t i = base + (uint64_t)k * T;
            if (i < N)
                {
                const uint32_t c[4] = {xyz[k].x, xyz[k].y, xyz[k].z, w[k]};
                unrows_store<F64>(dst + i * DW, c, 4, 0);
                }
            }
        return;
        }
    // any other one- or two-chunk shape: each chunk's elements go to their columns
#pragma unroll
    for (int k = 0; k < U; k++)
        {
        const uint64_t i = base + (uint64_t)k * T;
        if (i >= N)
            continue;
        uint32_t wa[4], wb[4] = {0, 0, 0, 0};
        unrows_load(a + i * g.a_nw, g.a_nw, wa);
        if (b != nullptr)
            unrows_load(b + i * g.b_nw, g.b_nw, wb);
        unrows_store<F64>(dst + i * DW, wa, g.a_nw, g.a_col0);
        if (b != nullptr)
            unrows_store<F64>(dst + i * DW, wb, g.b_nw, g.b_col0);
        }
    }

// ------------------------------------------------------------------ fill of untouched columns (generic paths)
// pgsd_field_dst.fill_rest where the launch does not assemble whole rows (scatter index, narrow or wide
// elements, more than two chunks per array): the columns in colmask of every destination row receive the fill
// element before the chunks' kernels run.  Element per lane: a fallback, not a hot path.
__global__ __launch_bounds__(256) void fill_cols_kernel(const FillArgs a)
    {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t row = t / a.stride;
    const uint32_t col = (uint32_t)(t % a.stride);
    if (row >= a.N || col >= 32 || !((a.colmask >> col) & 1u))
        return;
    const uint64_t r = a.order ? (uint64_t)a.order[row] : row;
    char* p = (char*)a.dst + (r * a.stride + col) * a.dsz;
    switch (a.dsz)
        {
        case 1: *(uint8_t*)p = (uint8_t)a.bits; break;
        case 2: *(uint16_t*)p = (uint16_t)a.bits; break;
        case 4: *(uint32_t*)p = (uint32_t)a.bits; break;
        default: *(uint64_t*)p = a.bits; break;
        }
    }

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

__global__ __launch_bounds__(SEL_THREADS) void select_count_kernel(const uint8_t* flags, uint64_t N,
                                                                   uint32_t* block_counts)
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

// exclusive scan of the block counts by ONE workgroup; also writes the total
__global__ __launch_bounds__(SEL_THREADS) void select_scan_kernel(uint32_t* block_counts, uint32_t n_blocks,
                                                                  uint64_t* block_offsets,
                                                                  uint64_t* out_count)
    {
    __shared__ uint64_t carry;
    __shared__ uint32_t wave_sums[SEL_THREADS / 64];
    if (threadIdx.x == 0)
        carry = 0;
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
            block_offsets[i] = carry + wave_off + inc - c;
        __syncthreads();
        if (threadIdx.x == SEL_THREADS - 1)
            carry += (uint64_t)wave_off + inc;
        __syncthreads();
        }
    if (threadIdx.x == 0)
        *out_count = carry;
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

// ------------------------------------------------------------------ host side

static uint32_t conv_kind(uint32_t src_type, uint32_t dst_type, uint32_t bitcast)
    {
    const bool s_int = src_type <= PGSD_TYPE_INT64, d_int = dst_type <= PGSD_TYPE_INT64;
    const size_t ssz = sizeof_type(src_type), dsz = sizeof_type(dst_type);
    if (bitcast || src_type == dst_type)
        return PACK_BITS;
    if (s_int && d_int)
        {
        const bool s_signed = src_type >= PGSD_TYPE_INT8;
        return (dsz > ssz && s_signed) ? PACK_SEXT : PACK_BITS;
        }
    if (!s_int && !d_int)
        return PACK_F2F;
    // integer -> float (checked by the caller: source <= 32 bit)
    return src_type >= PGSD_TYPE_INT8 ? PACK_S2F : PACK_U2F;
    }

static bool job_valid(const pgsd_pack_job& j)
    {
    const size_t ssz = sizeof_type(j.src.src_type), dsz = sizeof_type(j.dst_type);
    if (!j.dst || !j.src.src || ssz == 0 || dsz == 0 || j.M == 0 || j.src.src_col0 + j.M > j.src.src_stride)
        return false;
    const bool s_int = j.src.src_type <= PGSD_TYPE_INT64, d_int = j.dst_type <= PGSD_TYPE_INT64;
    if (j.src.bitcast)
        return dsz <= ssz;
    if (!s_int && d_int)
        return false;
    if (s_int && !d_int && ssz == 8)
        return false;
    return true;
    }

uint64_t pack_algorithmic_bytes_in(const pgsd_pack_job& j, uint64_t N)
    {
    return N * (uint64_t)j.M * sizeof_type(j.src.src_type) + (j.src.order ? N * 4 : 0);
    }

uint64_t pack_bytes_out(const pgsd_pack_job& j, uint64_t N)
    {
    return N * (uint64_t)j.M * sizeof_type(j.dst_type);
    }

static int g_num_cus = 0;

static int num_cus()
    {
    if (g_num_cus == 0)
        {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
            g_num_cus = prop.multiProcessorCount;
        if (g_num_cus <= 0)
            g_num_cus = 256;
        }
    return g_num_cus;
    }

// the LDS-tiled kernels: plain or software-pipelined, one instantiation per conversion class
static void launch_tiles(bool prefetch, int mode, unsigned blocks, size_t lds_bytes, hipStream_t stream,
                         const PackArgs& args, hipEvent_t ev_start, hipEvent_t ev_stop)
    {
    // hipExtLaunchKernelGGL stamps the events with the dispatch's own begin / end times, so a
    // profiled launch measures the kernel and nothing else (what rocprofv3 reports)
#define TILES_LAUNCH(KERNEL, MODE)                                                                              \
    hipExtLaunchKernelGGL((KERNEL<MODE>), dim3(blocks), dim3(PACK_THREADS), (uint32_t)lds_bytes, stream, ev_start, \
                          ev_stop, 0, args)
    if (prefetch)
        {
        if (mode == PACK_MODE_W32)
            TILES_LAUNCH(pack_tiles_prefetch_kernel, PACK_MODE_W32);
        else if (mode == PACK_MODE_F64_F32)
            TILES_LAUNCH(pack_tiles_prefetch_kernel, PACK_MODE_F64_F32);
        else
            TILES_LAUNCH(pack_tiles_prefetch_kernel, PACK_MODE_GENERIC);
        }
    else
        {
        if (mode == PACK_MODE_W32)
            TILES_LAUNCH(pack_tiles_kernel, PACK_MODE_W32);
        else if (mode == PACK_MODE_F64_F32)
            TILES_LAUNCH(pack_tiles_kernel, PACK_MODE_F64_F32);
        else
            TILES_LAUNCH(pack_tiles_kernel, PACK_MODE_GENERIC);
        }
#undef TILES_LAUNCH
    }

// ---- tuning knobs: the PGSD_PACK_* / PGSD_UNPACK_* variables of the sweeps in tools/ (pack_bench.py, unpack_bench.py).
// Read ONCE, when the first launch needs them; pgsd_reload_tuning() (pgsd_private.h) reads them again for tools
// that A/B variants inside one process.  Defaults come from measurements on MI355X (profiles/).
struct PackTuning
    {
    int rows_t = 0, rows_u = 0;           // PGSD_PACK_ROWS_CFG "<threads>x<rows per lane>" (0: by size)
    bool pack_tiles = false;              // PGSD_PACK_KERNEL=tiles: the LDS-tiled kernel for everything (A/B, tests)
    uint64_t per_cu = 4;                  // PGSD_PACK_BLOCKS_PER_CU
    uint32_t tile_cap = 1024;             // PGSD_PACK_TILE
    size_t lds_budget = PACK_LDS_BYTES;   // PGSD_PACK_LDS_KB
    int prefetch = -1;                    // PGSD_PACK_PREFETCH (-1: by size)
    int unrows_t = 0, unrows_u = 0;       // PGSD_UNPACK_ROWS_CFG (0: the default 64x2)
    uint32_t unpack_tile_cap = 0;         // PGSD_UNPACK_TILE (0: by size)
    uint64_t unpack_per_cu = 8;           // PGSD_UNPACK_BLOCKS_PER_CU
    bool unpack_tiles = false;            // PGSD_UNPACK_KERNEL=tiles
    };

static std::mutex g_tuning_lock;
static bool g_tuning_loaded = false;
static PackTuning g_tuning;

static PackTuning read_tuning()
    {
    PackTuning t;
    if (const char* e = getenv("PGSD_PACK_ROWS_CFG"))
        {
        // only the instantiated pairs (launch_rows below): the grid is sized from T x U, so a pair the
        // dispatcher does not know would pack too few rows per block and leave a part of every chunk unwritten
        int a = 0, u = 0;
        const bool known = sscanf(e, "%dx%d", &a, &u) == 2
                           && ((a == 64 && u == 2) || (a == 128 && (u == 1 || u == 2))
                               || (a == 256 && (u == 1 || u == 2 || u == 4 || u == 8)) || (a == 512 && (u == 2 || u == 4)));
        if (known)
            t.rows_t = a, t.rows_u = u;
        else
            fprintf(stderr, "pgsd_amd: PGSD_PACK_ROWS_CFG=%s is not one of 64x2 128x1 128x2 256x1 256x2 256x4 256x8 512x2 512x4: ignored\n", e);
        }
    if (const char* e = getenv("PGSD_PACK_KERNEL"))
        t.pack_tiles = strcmp(e, "tiles") == 0;
    if (const char* e = getenv("PGSD_PACK_BLOCKS_PER_CU"))
        t.per_cu = (uint64_t)atoi(e) > 0 ? (uint64_t)atoi(e) : t.per_cu;
    if (const char* e = getenv("PGSD_PACK_TILE"))
        t.tile_cap = (uint32_t)atoi(e) >= 16 ? (uint32_t)atoi(e) : t.tile_cap;
    if (const char* e = getenv("PGSD_PACK_LDS_KB"))
        t.lds_budget = (size_t)atoi(e) > 0 ? (size_t)atoi(e) << 10 : t.lds_budget;
    if (const char* e = getenv("PGSD_PACK_PREFETCH"))
        t.prefetch = atoi(e);
    if (const char* e = getenv("PGSD_UNPACK_ROWS_CFG"))
        {
        int a = 0, u = 0;
        if (sscanf(e, "%dx%d", &a, &u) == 2
            && ((a == 64 && u == 2) || (a == 128 && (u == 1 || u == 2)) || (a == 256 && (u == 1 || u == 2))))
            t.unrows_t = a, t.unrows_u = u;
        else
            fprintf(stderr, "pgsd_amd: PGSD_UNPACK_ROWS_CFG=%s is not one of 64x2 128x1 128x2 256x1 256x2: ignored\n", e);
        }
    if (const char* e = getenv("PGSD_UNPACK_TILE"))
        t.unpack_tile_cap = (uint32_t)atoi(e) >= 16 ? (uint32_t)atoi(e) : 0;
    if (const char* e = getenv("PGSD_UNPACK_BLOCKS_PER_CU"))
        t.unpack_per_cu = (uint64_t)atoi(e) > 0 ? (uint64_t)atoi(e) : t.unpack_per_cu;
    if (const char* e = getenv("PGSD_UNPACK_KERNEL"))
        t.unpack_tiles = strcmp(e, "tiles") == 0;
    return t;
    }

static PackTuning tuning()
    {
    std::lock_guard<std::mutex> guard(g_tuning_lock);
    if (!g_tuning_loaded)
        {
        g_tuning = read_tuning();
        g_tuning_loaded = true;
        }
    return g_tuning;
    }

void reload_pack_tuning()
    {
    std::lock_guard<std::mutex> guard(g_tuning_lock);
    g_tuning_loaded = false;
    }

// ---- row-per-lane launches
struct RowsCfg
    {
    int T, U;
    };

static RowsCfg rows_config(uint64_t N, uint32_t n_groups)
    {
    // measured, interleaved launch by launch in one process (profiles/r02_pack_ab.jsonl): 256 threads x 2
    // rows per lane is the best or within 1-2 % of the best for every workload from 1 M rows up
    // (10 M particles, HOOMD layout: 94.8 us against 106.5 us for the LDS-tiled kernel); a launch of one
    // or two arrays below 2 M rows does better with half as many, fatter workgroups
    RowsCfg c = (N < (2u << 20) && n_groups <= 2) ? RowsCfg {256, 4} : RowsCfg {256, 2};
    const PackTuning t = tuning();
    if (t.rows_t)
        c = RowsCfg {t.rows_t, t.rows_u};
    return c;
    }

template<int T, int U>
static void launch_rows_tu(const RowsArgs& a, bool copy, uint32_t rw, bool bits_only, hipStream_t stream, hipEvent_t e0,
                           hipEvent_t e1)
    {
    const dim3 grid((unsigned)a.n_blocks, a.n_groups);
    if (copy)
        {
        hipExtLaunchKernelGGL((pack_copy_kernel<T, U>), grid, dim3(T), 0, stream, e0, e1, 0, a);
        return;
        }
#define ROWS_LAUNCH(RW)                                                                                       \
    if (bits_only)                                                                                            \
        hipExtLaunchKernelGGL((pack_rows_kernel<T, U, RW, true>), grid, dim3(T), 0, stream, e0, e1, 0, a);    \
    else                                                                                                      \
        hipExtLaunchKernelGGL((pack_rows_kernel<T, U, RW, false>), grid, dim3(T), 0, stream, e0, e1, 0, a)
    switch (rw)
        {
        case 1: ROWS_LAUNCH(1); break;
        case 2: ROWS_LAUNCH(2); break;
        case 3: ROWS_LAUNCH(3); break;
        case 4: ROWS_LAUNCH(4); break;
        case 6: ROWS_LAUNCH(6); break;
        default: ROWS_LAUNCH(8); break;
        }
#undef ROWS_LAUNCH
    }

static void launch_rows(const RowsCfg& c, const RowsArgs& a, bool copy, uint32_t rw, bool bits_only, hipStream_t stream,
                        hipEvent_t e0, hipEvent_t e1)
    {
    if (c.T == 256 && c.U == 4)
        launch_rows_tu<256, 4>(a, copy, rw, bits_only, stream, e0, e1);
    else if (c.T == 256 && c.U == 2)
        launch_rows_tu<256, 2>(a, copy, rw, bits_only, stream, e0, e1);
    else if (c.T == 256 && c.U == 1)
        launch_rows_tu<256, 1>(a, copy, rw, bits_only, stream, e0, e1);
    else if (c.T == 128 && c.U == 2)
        launch_rows_tu<128, 2>(a, copy, rw, bits_only, stream, e0, e1);
    else if (c.T == 64 && c.U == 2)
        launch_rows_tu<64, 2>(a, copy, rw, bits_only, stream, e0, e1);
    else if (c.T == 256 && c.U == 8)
        launch_rows_tu<256, 8>(a, copy, rw, bits_only, stream, e0, e1);
    else if (c.T == 512 && c.U == 4)
        launch_rows_tu<512, 4>(a, copy, rw, bits_only, stream, e0, e1);
    else if (c.T == 512 && c.U == 2)
        launch_rows_tu<512, 2>(a, copy, rw, bits_only, stream, e0, e1);
    else
        launch_rows_tu<128, 1>(a, copy, rw, bits_only, stream, e0, e1);
    }

// Can the row-per-lane kernel take this job?  4- and 8-byte elements moved unchanged or converted
// between f32 and f64, source rows and chunk rows of at most 8 dwords, 16-byte aligned arrays.
static bool rows_eligible(const pgsd_pack_job& j, uint64_t N, uint32_t* kind_out)
    {
    const uint32_t ssz = (uint32_t)sizeof_type(j.src.src_type), dsz = (uint32_t)sizeof_type(j.dst_type);
    if ((ssz != 4 && ssz != 8) || (dsz != 4 && dsz != 8))
        return false;
    const uint32_t kind = conv_kind(j.src.src_type, j.dst_type, j.src.bitcast);
    uint32_t rk;
    if (kind == PACK_BITS && ssz == dsz)
        rk = ROWS_BITS;
    else if (kind == PACK_F2F && ssz == 8 && dsz == 4)
        rk = ROWS_F64_F32;
    else if (kind == PACK_F2F && ssz == 4 && dsz == 8)
        rk = ROWS_F32_F64;
    else
        return false;
    const uint64_t rw = (uint64_t)j.src.src_stride * ssz / 4, nw = (uint64_t)j.M * dsz / 4;
    const bool dense = rk == ROWS_BITS && j.src.src_col0 == 0 && j.M == j.src.src_stride;
    if (!dense && ((rw != 1 && rw != 2 && rw != 3 && rw != 4 && rw != 6 && rw != 8) || nw > ROWS_MAX_WORDS))
        return false;
    if ((((uintptr_t)j.dst | (uintptr_t)j.src.src) & 15) != 0)
        return false;
    // gathers (tag order through a permutation) stay with the LDS-tiled kernel: one random 16-byte row per
    // lane costs a whole memory request either way, and its four-deep row fetches measured faster than a
    // row-per-lane gather (417 vs 480-510 us for two float4 arrays of 10 M rows, profiles/r02_pack_ab.jsonl)
    if (j.src.order != nullptr)
        return false;
    if (N >= (1ull << 31)) // one lane per row (or per U rows): keeps the grid's x extent below 2^32 threads
        return false;
    *kind_out = rk;
    return true;
    }

int launch_pack(uint32_t n_jobs, const pgsd_pack_job* jobs, uint64_t N, hipStream_t stream, std::string* err,
                hipEvent_t ev_start, hipEvent_t ev_stop)
    {
    if (n_jobs == 0 || N == 0)
        return PGSD_SUCCESS;
    for (uint32_t i = 0; i < n_jobs; i++)
        if (!job_valid(jobs[i]))
            {
            if (err)
                *err = "invalid pack job (types, columns or pointers)";
            return PGSD_ERROR_INVALID_ARGUMENT;
            }
    // every kernel of the call, in issue order; the first one stamps ev_start, the last one ev_stop
    // (hipExtLaunchKernelGGL: the dispatch's own begin / end times, what rocprofv3 reports)
    std::vector<std::function<void(hipEvent_t, hipEvent_t)>> launches;
    std::vector<bool> done(n_jobs, false);

    // kernel choice (defaults from measurements on MI355X, profiles/; env overrides are for the
    // tuning sweeps of tools/pack_bench.py)
    enum
        {
        K_ROWS,
        K_TILES
        } kernel
        = K_ROWS;
    const PackTuning tune = tuning();
    if (tune.pack_tiles)
        kernel = K_TILES;

    // 1. the row-per-lane kernel takes every job it can: one launch per class of source-row width (a
    //    compile-time parameter), dense same-type arrays ride along in any launch; up to ROWS_MAX_GROUPS
    //    arrays each.  The headline layouts are ONE launch.
    if (kernel == K_ROWS)
        {
        while (true)
            {
            RowsArgs a;
            memset(&a, 0, sizeof(a));
            a.N = N;
            bool bits_only = true;
            uint32_t cls_rw = 0; // 0: no row-mode array in this launch yet
            // row-mode arrays first decide the class, then the dense ones fill the launch up
            for (int pass = 0; pass < 2; pass++)
                for (uint32_t i = 0; i < n_jobs; i++)
                    {
                    if (done[i])
                        continue;
                    const pgsd_pack_job& j = jobs[i];
                    uint32_t rk = 0;
                    if (!rows_eligible(j, N, &rk))
                        continue;
                    const uint32_t ssz = (uint32_t)sizeof_type(j.src.src_type), dsz = (uint32_t)sizeof_type(j.dst_type);
                    const uint32_t rw = j.src.src_stride * ssz / 4;
                    const bool dense = rk == ROWS_BITS && j.src.src_col0 == 0 && j.M == j.src.src_stride;
                    if (dense != (pass == 1))
                        continue;
                    if (!dense && cls_rw != 0 && rw != cls_rw)
                        continue; // another class: a later launch
                    int gi = -1;
                    if (!dense)
                        for (uint32_t k = 0; k < a.n_groups; k++)
                            if (a.g[k].copy_vecs == 0 && a.g[k].copy_tail == 0 && a.g[k].src == j.src.src
                                && a.g[k].n_out < PACK_MAX_OUT)
                                gi = (int)k;
                    if (gi < 0)
                        {
                        if (a.n_groups == ROWS_MAX_GROUPS)
                            continue; // next launch
                        gi = (int)a.n_groups++;
                        RowsGroup& g = a.g[gi];
                        g.src = j.src.src;
                        g.row_words = rw;
                        if (dense)
                            {
                            const uint64_t bytes = N * (uint64_t)j.src.src_stride * ssz;
                            g.copy_vecs = bytes >> 4;
                            g.copy_tail = (uint32_t)(bytes & 15);
                            }
                        else
                            cls_rw = rw;
                        }
                    RowsGroup& g = a.g[gi];
                    RowsOut& o = g.out[g.n_out++];
                    o.dst = j.dst;
                    o.col0 = j.src.src_col0 * ssz / 4; // first source dword
                    o.M = j.M;
                    o.kind = rk;
                    o.nw_out = j.M * dsz / 4;
                    bits_only = bits_only && rk == ROWS_BITS;
                    done[i] = true;
                    }
            if (a.n_groups == 0)
                break;
            const bool copy_only = cls_rw == 0;
            const RowsCfg cfg = rows_config(N, a.n_groups);
            const uint64_t per_block = (uint64_t)cfg.T * cfg.U;
            uint64_t blocks = 1;
            for (uint32_t k = 0; k < a.n_groups; k++)
                {
                const uint64_t units = (a.g[k].copy_vecs || a.g[k].copy_tail) ? a.g[k].copy_vecs : N;
                blocks = std::max(blocks, (units + per_block - 1) / per_block);
                }
            a.n_blocks = blocks; // N < 2^31 rows (rows_eligible) keeps the grid's x extent below 2^32 threads
            launches.push_back([=](hipEvent_t e0, hipEvent_t e1)
                               { launch_rows(cfg, a, copy_only, cls_rw, bits_only, stream, e0, e1); });
            }
        }

    // 2. jobs the tiled kernel cannot take either go through the generic kernel
    for (uint32_t i = 0; i < n_jobs; i++)
        {
        if (done[i])
            continue;
        const pgsd_pack_job& j = jobs[i];
        const size_t ssz = sizeof_type(j.src.src_type);
        const uint64_t rowbytes = (uint64_t)j.src.src_stride * ssz;
        const bool aligned = (((uintptr_t)j.dst | (uintptr_t)j.src.src) & 15) == 0
                             && (j.src.order == nullptr || ((uintptr_t)j.src.order & 3) == 0);
        if (!aligned || rowbytes > PACK_MAX_ROWBYTES || j.M > PACK_MAX_M || N * rowbytes >= (1ull << 62))
            {
            PackGenericArgs a;
            a.dst = j.dst;
            a.src = j.src.src;
            a.order = j.src.order;
            a.N = N;
            a.M = j.M;
            a.stride = j.src.src_stride;
            a.col0 = j.src.src_col0;
            a.ssz = (uint32_t)ssz;
            a.dsz = (uint32_t)sizeof_type(j.dst_type);
            a.kind = conv_kind(j.src.src_type, j.dst_type, j.src.bitcast);
            uint64_t total = N * j.M;
            uint64_t blocks = (total + PACK_THREADS - 1) / PACK_THREADS;
            uint64_t cap = (uint64_t)num_cus() * 8;
            if (blocks > cap)
                blocks = cap;
            launches.push_back(
                [a, blocks, stream](hipEvent_t e0, hipEvent_t e1)
                {
                    hipExtLaunchKernelGGL(pack_generic_kernel, dim3((unsigned)blocks), dim3(PACK_THREADS), 0, stream, e0, e1,
                                          0, a);
                });
            done[i] = true;
            }
        }

    // 3. the LDS-tiled kernel: group the remaining jobs by source array, in batches that fit PackArgs
    uint32_t next = 0;
    while (next < n_jobs && done[next])
        next++;
    while (next < n_jobs)
        {
        PackArgs args;
        memset(&args, 0, sizeof(args));
        args.N = N;
        uint32_t max_rowbytes = 0;
        bool any = false;
        int mode = -1; // a launch holds jobs of one specialisation only
        for (uint32_t i = next; i < n_jobs; i++)
            {
            if (done[i])
                continue;
            const pgsd_pack_job& j = jobs[i];
            const uint32_t ssz = (uint32_t)sizeof_type(j.src.src_type);
            const uint32_t jdsz = (uint32_t)sizeof_type(j.dst_type);
            const uint32_t jkind = conv_kind(j.src.src_type, j.dst_type, j.src.bitcast);
            const int jmode = (ssz == 4 && jdsz == 4 && jkind == PACK_BITS)  ? PACK_MODE_W32
                              : (ssz == 8 && jdsz == 4 && jkind == PACK_F2F) ? PACK_MODE_F64_F32
                                                                             : PACK_MODE_GENERIC;
            if (mode < 0)
                mode = jmode;
            else if (mode != jmode)
                continue; // next batch
            // find a group with the same source
            int gi = -1;
            for (uint32_t k = 0; k < args.n_groups; k++)
                if (args.g[k].src == j.src.src && args.g[k].order == j.src.order && args.g[k].ssz == ssz
                    && args.g[k].stride == j.src.src_stride && args.g[k].n_out < PACK_MAX_OUT)
                    gi = (int)k;
            if (gi < 0)
                {
                if (args.n_groups == PACK_MAX_GROUPS)
                    continue; // next batch
                gi = (int)args.n_groups++;
                PackGroup& g = args.g[gi];
                g.src = j.src.src;
                g.order = j.src.order;
                g.ssz = ssz;
                g.stride = j.src.src_stride;
                g.rowbytes = j.src.src_stride * ssz;
                g.n_out = 0;
                max_rowbytes = std::max(max_rowbytes, g.rowbytes);
                }
            PackGroup& g = args.g[gi];
            PackOut& o = g.out[g.n_out++];
            o.dst = j.dst;
            o.M = j.M;
            o.col0 = j.src.src_col0;
            o.dsz = (uint32_t)sizeof_type(j.dst_type);
            o.kind = conv_kind(j.src.src_type, j.dst_type, j.src.bitcast);
            o.magic = j.M == 1 ? 0u : (uint32_t)(((1ull << 32) + j.M - 1) / j.M);
            done[i] = true;
            any = true;
            }
        if (!any)
            break;
        const uint64_t per_cu = tune.per_cu;
        const uint32_t tile_cap = tune.tile_cap;
        const size_t lds_budget = tune.lds_budget;
        // tile: as many rows as the widest source row allows (power of two in [16, tile_cap]);
        // consecutive source arrays then share a batch while their tiles fit the budget
        uint32_t tile = 16;
        while (tile * 2 <= tile_cap && (uint64_t)tile * 2 * max_rowbytes <= lds_budget)
            tile <<= 1;
        args.tile_rows = tile;
        args.n_tiles = (N + tile - 1) / tile;
        size_t lds_bytes = 0, used = 0;
        args.n_batches = 0;
        args.batch_start[0] = 0;
        for (uint32_t k = 0; k < args.n_groups; k++)
            {
            size_t lin = (size_t)tile * args.g[k].rowbytes;
            size_t need = (lin + ((lin >> 7) << 4) + 31) & ~(size_t)15; // see lds_skew
            if (used != 0 && used + need > lds_budget)
                {
                args.batch_start[++args.n_batches] = (uint8_t)k;
                used = 0;
                }
            args.g[k].lds_off = (uint32_t)used;
            used += need;
            lds_bytes = std::max(lds_bytes, used);
            }
        args.batch_start[++args.n_batches] = (uint8_t)args.n_groups;
        // never ask for more workgroups per CU than the 160 KiB of LDS admit: the surplus
        // would queue behind the resident ones and run as a ragged second wave
        uint64_t resident = lds_bytes ? (160u * 1024u) / lds_bytes : 8;
        if (resident < 1)
            resident = 1;
        uint64_t blocks = args.n_tiles;
        uint64_t cap = (uint64_t)num_cus() * std::min<uint64_t>(per_cu, resident);
        if (blocks > cap)
            blocks = cap;
        // the software-pipelined kernel takes launches it has registers for: one batch, linear
        // sources, tiles of at most PF_VECS x 256 vectors
        bool prefetch = args.n_batches == 1 && args.n_groups <= PF_GROUPS;
        for (uint32_t k = 0; k < args.n_groups; k++)
            prefetch = prefetch && args.g[k].order == nullptr
                       && (size_t)tile * args.g[k].rowbytes <= (size_t)PF_VECS * PACK_THREADS * 16;
        const int want = tune.prefetch; // -1: by size
        if (want == 0 || (want < 0 && args.n_tiles > 2 * blocks))
            prefetch = false;
        launches.push_back(
            [=](hipEvent_t e0, hipEvent_t e1)
            { launch_tiles(prefetch, mode, (unsigned)blocks, lds_bytes, stream, args, e0, e1); });
        while (next < n_jobs && done[next])
            next++;
        }
    for (size_t i = 0; i < launches.size(); i++)
        launches[i](i == 0 ? ev_start : nullptr, i + 1 == launches.size() ? ev_stop : nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        {
        if (err)
            *err = std::string("pack kernel launch failed: ") + hipGetErrorString(e);
        return PGSD_ERROR_DEVICE;
        }
    return PGSD_SUCCESS;
    }
// ---- row-per-lane unpack: which destination arrays it takes and how it is launched
struct UnrowsPlan
    {
    UnrowsArgs args;
    bool f64 = false;
    };

static void launch_unrows(const UnrowsPlan& p, uint64_t N, hipStream_t stream)
    {
    // measured (profiles/r02_lab_unpack.jsonl, r02_unpack_rows_final.jsonl): thin workgroups; 64 x 2 rows
    // 102.4-103.0 us, 128 x 1 104.6-105.0 us, 256 x 2 105.9-106.0 us (10 M particles, stream events)
    int T = 64, U = 2;
    const PackTuning tune = tuning();
    if (tune.unrows_t)
        T = tune.unrows_t, U = tune.unrows_u;
    UnrowsArgs a = p.args;
    a.n_blocks = (N + (uint64_t)T * U - 1) / ((uint64_t)T * U);
    const dim3 grid((unsigned)a.n_blocks, a.n_groups);
#define UNROWS_LAUNCH(TT, UU)                                                                                  \
    if (T == TT && U == UU)                                                                                    \
        {                                                                                                      \
        if (p.f64)                                                                                             \
            hipLaunchKernelGGL((unpack_rows_kernel<TT, UU, true>), grid, dim3(TT), 0, stream, a);              \
        else                                                                                                   \
            hipLaunchKernelGGL((unpack_rows_kernel<TT, UU, false>), grid, dim3(TT), 0, stream, a);             \
        return;                                                                                                \
        }
    UNROWS_LAUNCH(128, 1)
    UNROWS_LAUNCH(256, 1)
    UNROWS_LAUNCH(256, 2)
    UNROWS_LAUNCH(128, 2)
    a.n_blocks = (N + 127) / 128;
    const dim3 grid1((unsigned)a.n_blocks, a.n_groups);
    if (p.f64)
        hipLaunchKernelGGL((unpack_rows_kernel<64, 2, true>), grid1, dim3(64), 0, stream, a);
    else
        hipLaunchKernelGGL((unpack_rows_kernel<64, 2, false>), grid1, dim3(64), 0, stream, a);
#undef UNROWS_LAUNCH
    }

// One batch of <= UNPACK_MAX_JOBS validated chunks -> one launch.
static void launch_unpack_batch(const std::vector<UnpackJob>& jobs, uint64_t N, hipStream_t stream)
    {
    UnpackArgs args;
    memset(&args, 0, sizeof(args));
    args.N = N;
    args.n_jobs = (uint32_t)jobs.size();
    uint32_t sum_rowbytes = 0;
    bool w32 = true;
    for (uint32_t i = 0; i < args.n_jobs; i++)
        {
        args.j[i] = jobs[i];
        sum_rowbytes += jobs[i].rowbytes;
        w32 = w32 && jobs[i].ssz == 4 && jobs[i].dsz == 4 && jobs[i].kind == PACK_BITS;
        }
    // destination arrays whose rows this batch restores completely: dst row of 16, 32 or 64 bytes,
    // 4- or 8-byte elements, every column written by exactly one chunk
    for (uint32_t i = 0; i < args.n_jobs && args.n_groups < UNPACK_MAX_GROUPS; i++)
        {
        UnpackJob& a = args.j[i];
        if (a.in_group)
            continue;
        const uint32_t rowbytes = a.dst_stride * a.dsz;
        if ((a.dsz != 4 && a.dsz != 8) || (rowbytes != 16 && rowbytes != 32 && rowbytes != 64)
            || (((uintptr_t)a.dst) & 15) != 0)
            continue;
        UnpackGroup g;
        memset(&g, 0, sizeof(g));
        uint8_t covered[UNPACK_MAX_ROW_COLS] = {0};
        uint32_t n_cov = 0;
        bool clean = true;
        for (uint32_t k = i; k < args.n_jobs; k++)
            {
            const UnpackJob& b = args.j[k];
            if (b.dst != a.dst || b.in_group)
                continue;
            if (b.order != a.order || b.dst_stride != a.dst_stride || b.dsz != a.dsz)
                {
                clean = false; // same array seen through different shapes: leave it to the element path
                break;
                }
            for (uint32_t c = 0; c < b.M; c++)
                {
                if (covered[b.dst_col0 + c])
                    clean = false;
                covered[b.dst_col0 + c] = 1;
                g.col_job[b.dst_col0 + c] = (uint8_t)k;
                g.col_off[b.dst_col0 + c] = (uint8_t)c;
                n_cov++;
                }
            }
        if (!clean || n_cov != a.dst_stride)
            continue;
        g.dst = a.dst;
        g.order = a.order;
        g.stride = a.dst_stride;
        g.dsz = a.dsz;
        g.vec_shift = rowbytes == 16 ? 0u : (rowbytes == 32 ? 1u : 2u);
        for (uint32_t k = i; k < args.n_jobs; k++)
            if (args.j[k].dst == a.dst)
                args.j[k].in_group = 1;
        args.g[args.n_groups++] = g;
        }
    // measured (profiles/r01_unpack_sweep.log): the unpack wants more resident workgroups than the
    // pack -- 512-row tiles x 8 workgroups per CU beat 1024 x 4 by 8 % at 10 M rows; launches too small
    // to fill the chip twice keep the larger tile
    uint32_t tile = 16, tile_cap = N > (1ull << 21) ? 512 : 1024;
    uint64_t per_cu = 8;
    const PackTuning tune = tuning(); // tuning sweeps (tools/unpack_bench.py)
    if (tune.unpack_tile_cap)
        tile_cap = tune.unpack_tile_cap;
    per_cu = tune.unpack_per_cu;
    while (tile * 2 <= tile_cap && (uint64_t)tile * 2 * sum_rowbytes <= UNPACK_LDS_BYTES)
        tile <<= 1;
    args.tile_rows = tile;
    args.n_tiles = (N + tile - 1) / tile;
    size_t lds_bytes = UNPACK_TABLE_BYTES; // the column table of the kernel sits in front
    for (uint32_t i = 0; i < args.n_jobs; i++)
        {
        args.j[i].lds_off = (uint32_t)lds_bytes;
        lds_bytes += (size_t)tile * args.j[i].rowbytes; // tile is a multiple of 16: stays 16-byte aligned
        }
    uint64_t resident = lds_bytes ? (160u * 1024u) / lds_bytes : 8;
    uint64_t blocks = args.n_tiles;
    uint64_t cap = (uint64_t)num_cus() * std::max<uint64_t>(1, std::min<uint64_t>(per_cu, resident));
    if (blocks > cap)
        blocks = cap;
    if (w32)
        hipLaunchKernelGGL(unpack_tiles_kernel<true>, dim3((unsigned)blocks), dim3(PACK_THREADS), lds_bytes, stream, args);
    else
        hipLaunchKernelGGL(unpack_tiles_kernel<false>, dim3((unsigned)blocks), dim3(PACK_THREADS), lds_bytes, stream, args);
    }

int launch_unpack(uint32_t n_jobs, const pgsd_unpack_job* jobs, uint64_t N, hipStream_t stream, std::string* err)
    {
    if (n_jobs == 0 || N == 0)
        return PGSD_SUCCESS;
    std::vector<UnpackJob> all;
    all.reserve(n_jobs);
    for (uint32_t i = 0; i < n_jobs; i++)
        {
        const pgsd_unpack_job& q = jobs[i];
        const uint32_t ssz = (uint32_t)sizeof_type(q.src_type), dsz = (uint32_t)sizeof_type(q.dst.dst_type);
        const bool s_int = q.src_type <= PGSD_TYPE_INT64, d_int = q.dst.dst_type <= PGSD_TYPE_INT64;
        bool ok = q.src && q.dst.dst && ssz && dsz && q.M && q.M <= PACK_MAX_M
                  && q.dst.dst_col0 + q.M <= q.dst.dst_stride && (((uintptr_t)q.src) & 15) == 0
                  && (((uintptr_t)q.dst.dst) & (dsz - 1)) == 0 && (uint64_t)q.M * ssz <= PACK_MAX_ROWBYTES;
        if (q.dst.bitcast)
            ok = ok && ssz == dsz;
        else
            ok = ok && !(!s_int && d_int) && !(s_int && !d_int && ssz == 8);
        ok = ok && !(q.dst.fill_rest && q.dst.dst_stride > 32); // the fill addresses columns with a 32-bit mask
        if (!ok)
            {
            if (err)
                *err = "invalid unpack job (types, columns, alignment or pointers)";
            return PGSD_ERROR_INVALID_ARGUMENT;
            }
        UnpackJob j;
        memset(&j, 0, sizeof(j));
        j.src = q.src;
        j.dst = q.dst.dst;
        j.order = q.dst.order;
        j.M = q.M;
        j.ssz = ssz;
        j.dsz = dsz;
        j.kind = conv_kind(q.src_type, q.dst.dst_type, q.dst.bitcast);
        j.dst_stride = q.dst.dst_stride;
        j.dst_col0 = q.dst.dst_col0;
        j.magic = q.M == 1 ? 0u : (uint32_t)(((1ull << 32) + q.M - 1) / q.M);
        j.rowbytes = q.M * ssz;
        j.fill_rest = q.dst.fill_rest ? 1u : 0u;
        j.fill_bits = q.dst.fill_bits;
        all.push_back(j);
        }
    // chunks of one destination array next to each other (their relative order is kept)
    std::stable_sort(all.begin(), all.end(), [](const UnpackJob& a, const UnpackJob& b) { return (uintptr_t)a.dst < (uintptr_t)b.dst; });
    // 1. destination arrays the row-per-lane kernel takes: rows of four 4-byte elements (or four doubles
    //    restored from f32 chunks) fed by one or two chunks of 4-byte elements on disjoint columns, no
    //    scatter index; plus dense same-type arrays (the chunk IS the array).  One launch per conversion
    //    class; everything else goes to the LDS-tiled kernel below.
    if (N < (1ull << 31) && !tuning().unpack_tiles)
        {
        std::vector<bool> taken(all.size(), false);
        for (int f64 = 0; f64 < 2; f64++)
            {
            while (true)
                {
                UnrowsPlan plan;
                memset(&plan.args, 0, sizeof(plan.args));
                plan.args.N = N;
                plan.f64 = f64 != 0;
                for (size_t i = 0; i < all.size();)
                    {
                    size_t e = i; // [i, e) = the chunks of one destination array (sorted by dst, order kept)
                    while (e < all.size() && all[e].dst == all[i].dst)
                        e++;
                    const UnpackJob& j0 = all[i];
                    const size_t n = e - i;
                    bool ok = !taken[i] && n <= 2 && plan.args.n_groups < ROWS_MAX_GROUPS && (((uintptr_t)j0.dst) & 15) == 0;
                    for (size_t k = i; k < e && ok; k++)
                        {
                        const UnpackJob& j = all[k];
                        ok = j.order == nullptr && j.dst_stride == j0.dst_stride && j.dsz == j0.dsz
                             && (((uintptr_t)j.src) & 15) == 0;
                        }
                    // a dense array of the chunk's own type (any element size, any row width): a plain copy
                    const bool dense = ok && n == 1 && j0.kind == PACK_BITS && j0.ssz == j0.dsz && j0.dst_col0 == 0
                                       && j0.M == j0.dst_stride;
                    if (dense)
                        ok = f64 == 0; // rides along in the launch of the first pass
                    else if (ok)
                        {
                        ok = j0.dst_stride == 4 && (f64 ? j0.dsz == 8 : j0.dsz == 4);
                        for (size_t k = i; k < e && ok; k++)
                            ok = all[k].ssz == 4 && all[k].kind == (uint32_t)(f64 ? PACK_F2F : PACK_BITS) && all[k].M <= 4;
                        if (ok && n == 2) // disjoint columns: no "later chunk wins" question inside a row
                            ok = all[i].dst_col0 + all[i].M <= all[i + 1].dst_col0
                                 || all[i + 1].dst_col0 + all[i + 1].M <= all[i].dst_col0;
                        }
                    if (ok)
                        {
                        UnrowsGroup& g = plan.args.g[plan.args.n_groups++];
                        g.dst = j0.dst;
                        g.a = j0.src;
                        for (size_t k = i; k < e && !dense; k++)
                            if (all[k].fill_rest && !g.fill_on)
                                {
                                g.fill_on = 1;
                                g.fill_lo = (uint32_t)all[k].fill_bits;
                                g.fill_hi = (uint32_t)(all[k].fill_bits >> 32);
                                }
                        if (dense)
                            {
                            const uint64_t bytes = N * (uint64_t)j0.M * j0.ssz;
                            g.copy_vecs = bytes >> 4;
                            g.copy_tail = (uint32_t)(bytes & 15);
                            }
                        else
                            {
                            // `a` = the chunk of the lower columns (xyz before w: the kernel's static hot shape)
                            const UnpackJob& lo = (n == 2 && all[i + 1].dst_col0 < j0.dst_col0) ? all[i + 1] : j0;
                            g.a = lo.src;
                            g.a_nw = lo.M;
                            g.a_col0 = lo.dst_col0;
                            if (n == 2)
                                {
                                const UnpackJob& hi = (&lo == &j0) ? all[i + 1] : j0;
                                g.b = hi.src;
                                g.b_nw = hi.M;
                                g.b_col0 = hi.dst_col0;
                                }
                            }
                        for (size_t k = i; k < e; k++)
                            taken[k] = true;
                        }
                    i = e;
                    }
                if (plan.args.n_groups == 0)
                    break;
                launch_unrows(plan, N, stream);
                }
            }
        std::vector<UnpackJob> rest;
        for (size_t i = 0; i < all.size(); i++)
            if (!taken[i])
                rest.push_back(all[i]);
        all.swap(rest);
        }
    // fills the remaining (tiled / generic) chunks asked for: one pass per destination array over the columns
    // none of ITS chunks writes, ahead of the chunks on the stream
    for (size_t i = 0; i < all.size();)
        {
        size_t e = i;
        while (e < all.size() && all[e].dst == all[i].dst)
            e++;
        uint32_t covered = 0;
        const UnpackJob* want = nullptr;
        for (size_t k = i; k < e; k++)
            {
            for (uint32_t c = 0; c < all[k].M && all[k].dst_col0 + c < 32; c++)
                covered |= 1u << (all[k].dst_col0 + c);
            if (all[k].fill_rest && !want)
                want = &all[k];
            }
        if (want && want->dst_stride <= 32)
            {
            FillArgs fa;
            memset(&fa, 0, sizeof(fa));
            fa.dst = want->dst;
            fa.order = want->order;
            fa.N = N;
            fa.bits = want->fill_bits;
            fa.stride = want->dst_stride;
            fa.dsz = want->dsz;
            fa.colmask = ~covered & (want->dst_stride >= 32 ? 0xffffffffu : ((1u << want->dst_stride) - 1u));
            if (fa.colmask)
                {
                const uint64_t lanes = N * (uint64_t)fa.stride;
                hipLaunchKernelGGL(fill_cols_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, stream, fa);
                }
            }
        i = e;
        }
    std::vector<UnpackJob> batch;
    uint32_t sum_rowbytes = 0;
    for (size_t i = 0; i < all.size(); i++)
        {
        const UnpackJob& j = all[i];
        // a chunk that rewrites columns an earlier chunk of the batch wrote goes to the next launch:
        // "the later chunk wins" then holds by stream order
        bool overlap = false;
        for (const UnpackJob& b : batch)
            if (b.dst == j.dst && j.dst_col0 < b.dst_col0 + b.M && b.dst_col0 < j.dst_col0 + j.M)
                overlap = true;
        if (!batch.empty()
            && (overlap || batch.size() == UNPACK_MAX_JOBS || sum_rowbytes + j.rowbytes > UNPACK_MAX_SUM_ROWBYTES))
            {
            launch_unpack_batch(batch, N, stream);
            batch.clear();
            sum_rowbytes = 0;
            }
        batch.push_back(j);
        sum_rowbytes += j.rowbytes;
        }
    if (!batch.empty())
        launch_unpack_batch(batch, N, stream);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        {
        if (err)
            *err = std::string("unpack kernel launch failed: ") + hipGetErrorString(e);
        return PGSD_ERROR_DEVICE;
        }
    return PGSD_SUCCESS;
    }
    } // namespace pgsd_amd

using namespace pgsd_amd;

extern "C" int pgsd_unpack_fields(uint32_t n_jobs, const struct pgsd_unpack_job* jobs, uint64_t N, void* stream)
    try
    {
    if (n_jobs > 0 && !jobs)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (!pgsd_device_available())
        {
        set_last_error("pgsd_unpack_fields: no HIP device visible (the HIP path has no CPU fallback)");
        return PGSD_ERROR_NO_DEVICE;
        }
    std::string err;
    int rc = launch_unpack(n_jobs, jobs, N, (hipStream_t)stream, &err);
    if (rc != PGSD_SUCCESS)
        set_last_error(err);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_device_available(void)
    try
    {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess)
        {
        (void)hipGetLastError();
        return 0;
        }
    return n > 0 ? 1 : 0;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_pack_fields(uint32_t n_jobs, const struct pgsd_pack_job* jobs, uint64_t N, void* stream, float* kernel_ms)
    try
    {
    if (n_jobs > 0 && !jobs)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (kernel_ms)
        *kernel_ms = 0.f;
    if (!pgsd_device_available())
        {
        set_last_error("pgsd_pack_fields: no HIP device visible (the HIP path has no CPU fallback)");
        return PGSD_ERROR_NO_DEVICE;
        }
    std::string err;
    if (!kernel_ms)
        {
        int rc = launch_pack(n_jobs, jobs, N, (hipStream_t)stream, &err);
        if (rc != PGSD_SUCCESS)
            set_last_error(err);
        return rc;
        }
    // timed: the dispatches' own begin / end stamps (what rocprofv3 reports per kernel; no launch latency)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)
        return PGSD_ERROR_DEVICE;
    int rc = launch_pack(n_jobs, jobs, N, (hipStream_t)stream, &err, e0, e1);
    if (rc != PGSD_SUCCESS)
        set_last_error(err);
    else if (n_jobs > 0 && N > 0)
        {
        if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(kernel_ms, e0, e1) != hipSuccess)
            rc = PGSD_ERROR_DEVICE;
        }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return rc;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

// scratch space of pgsd_select_rows: the library's, one per device, grown on demand (a call holds the lock: it ends
// with a stream synchronisation anyway)
namespace
    {
struct SelectScratch
    {
    void* dev = nullptr;
    size_t cap = 0;
    uint64_t* host_count = nullptr; // pinned
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
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess)
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
                set_last_error("pgsd_select_rows: cannot allocate the scratch space");
                return PGSD_ERROR_MEMORY_ALLOCATION_FAILED;
                }
            sc.cap = cap;
            }
        if (!sc.host_count && hipHostMalloc((void**)&sc.host_count, sizeof(uint64_t), hipHostMallocDefault) != hipSuccess)
            {
            set_last_error("pgsd_select_rows: cannot allocate pinned memory");
            return PGSD_ERROR_MEMORY_ALLOCATION_FAILED;
            }
        }
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
                           (uint32_t)n_blocks, block_offsets, out_count);
        hipLaunchKernelGGL(select_scatter_kernel, dim3((unsigned)n_blocks), dim3(SEL_THREADS), 0, stream, flags,
                           N, block_offsets, out_index);
        }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        {
        set_last_error(std::string("select kernel launch failed: ") + hipGetErrorString(e));
        return PGSD_ERROR_DEVICE;
        }
    e = hipMemcpyAsync(sc.host_count, out_count, sizeof(uint64_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess)
        e = hipStreamSynchronize(stream);
    if (e != hipSuccess)
        {
        set_last_error(std::string("pgsd_select_rows: ") + hipGetErrorString(e));
        return PGSD_ERROR_DEVICE;
        }
    *out_count_host = *sc.host_count;
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" void pgsd_reload_tuning(void)
    try
    {
    pgsd_amd::reload_pack_tuning();
    }
catch (...)
    {
        pgsd_amd::abi_guard();
    }

extern "C" uint32_t pgsd_abi_version(void)
    {
    return PGSD_ABI_VERSION;
    }

// Device memory owned by the library (include/pgsd.h): what pgsd.fl / pgsd.hoomd keep elision references, device
// reads and index lists in, so that the Python device path needs no tensor library.
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
