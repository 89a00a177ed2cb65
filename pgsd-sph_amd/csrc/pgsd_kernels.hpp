// pgsd_kernels.hpp -- what the three kernel translation units of libpgsd_amd.so share (round 5: pgsd_pack.hip was one
// 2 600-line file): pgsd_pack.hip (pack: LDS-tiled, row-per-lane, copy and generic kernels + their launcher),
// pgsd_unpack.hip (the read path's inverse kernels + launcher) and pgsd_select.hip (chunk comparison, stream compaction,
// library-owned device memory).  Device helpers are header-only; the few host helpers are defined in pgsd_pack.hip.
#ifndef PGSD_KERNELS_HPP
#define PGSD_KERNELS_HPP

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

// ------------------------------------------------------------------ rows in registers (row-per-lane kernels, both directions)
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

// ------------------------------------------------------------------ host side, shared (definitions: pgsd_pack.hip)
// element conversion class PACK_* of (source type, chunk type, bitcast)
uint32_t conv_kind(uint32_t src_type, uint32_t dst_type, uint32_t bitcast);
int num_cus();

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

PackTuning tuning();
    } // namespace pgsd_amd

#endif
