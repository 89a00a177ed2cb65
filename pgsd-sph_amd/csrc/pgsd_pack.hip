// pgsd_pack.hip -- gfx950 (CDNA4 / MI355X) kernels of the snapshot PACK path and their launcher.
//
// pack:    chunk[i][c] = convert(src[(order ? order[i] : i) * stride + col0 + c])
//          for every field of a frame, from HBM-resident particle arrays (HOOMD-style
//          float4 / double4 / scalar arrays) into dense GSD chunk buffers.
// (unpack, the inverse for restart reads: pgsd_unpack.hip; chunk comparison, stream compaction and library-owned device
//  memory: pgsd_select.hip; shared device helpers: pgsd_kernels.hpp.)
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
#include "pgsd_kernels.hpp"

namespace pgsd_amd
    {
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

// ------------------------------------------------------------------ host side

uint32_t conv_kind(uint32_t src_type, uint32_t dst_type, uint32_t bitcast)
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

int num_cus()
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

// ---- tuning knobs (struct PackTuning: pgsd_kernels.hpp)
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

PackTuning tuning()
    {
    std::lock_guard<std::mutex> guard(g_tuning_lock);
    if (!g_tuning_loaded)
        {
        g_tuning = read_tuning();
        g_tuning_loaded = true;
        }
    return g_tuning;
    }

void warm_pack_kernels()
    {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, (const void*)pack_generic_kernel);
    (void)hipGetLastError();
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
    // whatever an earlier call of this thread left in the runtime's last-error slot (a failed hipMalloc, the caller's own
    // calls) is not this launch's: the slot is read again right behind the launches
    (void)hipGetLastError();
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
    } // namespace pgsd_amd

using namespace pgsd_amd;

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
