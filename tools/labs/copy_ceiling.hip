// tools/labs/copy_ceiling.hip -- the streaming ceiling the pack kernels are judged against.
//
// A bare register float4 copy (no LDS, no barrier, U independent 16-byte loads in flight per lane) moving as many
// bytes as one pack launch moves: 600 MB for the headline (10 M particles: 320 MB read + 280 MB written) and 62.9 MB
// for BASELINE config 2 (2^20 particles).  Buffer pairs are rotated so that nothing is re-read from the 256 MiB
// Infinity Cache; every launch is timed by the dispatch's own begin / end stamps (hipExtLaunchKernelGGL events),
// i.e. what rocprofv3 --kernel-trace reports per kernel.  MI355X_MICROARCH.md quotes 6.29 TB/s for this shape.
//
//   hipcc -O3 --offload-arch=gfx950 tools/labs/copy_ceiling.hip -o gpurun_out/copy_ceiling
//   ./copy_ceiling [reps=40]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CK(x)                                                                                 \
    do                                                                                        \
        {                                                                                     \
        hipError_t e_ = (x);                                                                  \
        if (e_ != hipSuccess)                                                                 \
            {                                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(2);                                                                          \
            }                                                                                 \
        } while (0)

template<int U, bool NT> __global__ __launch_bounds__(256) void copy_kernel(const u32x4* __restrict__ a, u32x4* __restrict__ b,
                                                                            uint64_t nvec)
    {
    const uint64_t piece = (uint64_t)U * 256;
    for (uint64_t base = (uint64_t)blockIdx.x * piece; base < nvec; base += (uint64_t)gridDim.x * piece)
        {
        u32x4 r[U];
#pragma unroll
        for (int u = 0; u < U; u++)
            {
            const uint64_t v = base + (uint64_t)u * 256 + threadIdx.x;
            if (v < nvec)
                r[u] = NT ? __builtin_nontemporal_load(a + v) : a[v];
            }
#pragma unroll
        for (int u = 0; u < U; u++)
            {
            const uint64_t v = base + (uint64_t)u * 256 + threadIdx.x;
            if (v < nvec)
                {
                if (NT)
                    __builtin_nontemporal_store(r[u], b + v);
                else
                    b[v] = r[u];
                }
            }
        }
    }

// grid_blocks > 0: a persistent-style launch of that many workgroups walking the array with a grid stride (reads and
// writes interleave in steady state instead of "every workgroup loads, then every workgroup stores")
template<int U, bool NT> static void run(const char* what, uint64_t moved_bytes, int reps, hipStream_t s, unsigned grid_blocks = 0)
    {
    const uint64_t nvec = moved_bytes / 32; // each vector is read once and written once
    const int n_sets = (int)std::max<uint64_t>(2, (600ull << 20) / moved_bytes + 2);
    std::vector<u32x4*> a(n_sets), b(n_sets);
    for (int i = 0; i < n_sets; i++)
        {
        CK(hipMalloc(&a[i], nvec * 16));
        CK(hipMalloc(&b[i], nvec * 16));
        CK(hipMemsetAsync(a[i], 0x3c + i, nvec * 16, s));
        }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned blocks = grid_blocks ? grid_blocks : (unsigned)((nvec + U * 256 - 1) / (U * 256));
    std::vector<float> us;
    for (int r = 0; r < reps + 5; r++)
        {
        const int k = r % n_sets;
        hipExtLaunchKernelGGL((copy_kernel<U, NT>), dim3(blocks), dim3(256), 0, s, e0, e1, 0, a[k], b[k], nvec);
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (r >= 5)
            us.push_back(ms * 1e3f);
        }
    std::sort(us.begin(), us.end());
    const float med = us[us.size() / 2];
    printf("{\"lab\": \"copy_ceiling\", \"what\": \"%s\", \"moved_bytes\": %llu, \"rows_per_lane\": %d, \"nontemporal\": %s, "
           "\"blocks\": %u, \"buffer_sets\": %d, \"median_us\": %.2f, \"min_us\": %.2f, \"TBps\": %.3f, \"frac_of_8TBps\": %.3f}\n",
           what, (unsigned long long)moved_bytes, U, NT ? "true" : "false", blocks, n_sets, med, us[0], moved_bytes / med / 1e6,
           moved_bytes / med / 1e6 / 8.0);
    fflush(stdout);
    for (int i = 0; i < n_sets; i++)
        {
        CK(hipFree(a[i]));
        CK(hipFree(b[i]));
        }
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
    }

int main(int argc, char** argv)
    {
    const int reps = argc > 1 ? atoi(argv[1]) : 40;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    // headline: 10 M particles x (32 B read + 28 B written)
    run<2, true>("headline 10M particles (600 MB)", 600000000ull, reps, s);
    run<4, true>("headline 10M particles (600 MB)", 600000000ull, reps, s);
    run<4, false>("headline 10M particles (600 MB)", 600000000ull, reps, s);
    // BASELINE config 2: 2^20 particles x 60 B
    run<2, true>("config 2: 2^20 particles (62.9 MB)", 62914560ull, reps * 5, s);
    run<4, true>("config 2: 2^20 particles (62.9 MB)", 62914560ull, reps * 5, s);
    run<4, false>("config 2: 2^20 particles (62.9 MB)", 62914560ull, reps * 5, s);
    // does ANY launch shape move config 2's bytes faster?  persistent grids (1-8 workgroups per CU), 1-8 vectors per lane
    for (unsigned per_cu : {1u, 2u, 4u, 8u})
        {
        run<1, true>("config 2, persistent grid", 62914560ull, reps * 5, s, 256 * per_cu);
        run<2, true>("config 2, persistent grid", 62914560ull, reps * 5, s, 256 * per_cu);
        run<4, true>("config 2, persistent grid", 62914560ull, reps * 5, s, 256 * per_cu);
        }
    run<1, true>("config 2: 2^20 particles (62.9 MB)", 62914560ull, reps * 5, s);
    run<8, true>("config 2: 2^20 particles (62.9 MB)", 62914560ull, reps * 5, s);
    return 0;
    }
