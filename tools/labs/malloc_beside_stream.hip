// tools/labs/malloc_beside_stream.hip -- does a hipMalloc of a 256 MiB staging block, made by ANOTHER thread, stall
// a stream of short kernels?  (The device pipeline allocates such a block when asynchronously sealed frames pile up;
// made on the caller's thread it costs the simulation 1-5 ms, profiles/r05_dump_writer.jsonl.)
// The main thread launches 14-us kernels back to back and records the largest gap between the completion stamps of
// consecutive launches while a helper thread allocates (and touches nothing); then the same with the allocation made
// on the launching thread itself.
//   hipcc -O2 --offload-arch=gfx950 tools/labs/malloc_beside_stream.hip -o tools/labs/build/malloc_beside_stream -lpthread
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void busy(float* p, size_t n)
    {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        p[i] = p[i] * 1.0001f + 0.5f;
    }

static double now_us()
    {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

int main()
    {
    const size_t n = 1u << 20;
    float* p;
    hipMalloc((void**)&p, n * 4 * 4);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    // beside it, as in the device pipeline: device->pinned copies of 16 MiB pieces on a stream of their own, all the time
    std::atomic<int> copies_stop {0};
    std::thread copier(
        [&]
        {
            hipSetDevice(0);
            hipStream_t cs;
            hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
            char* dsrc;
            hipMalloc((void**)&dsrc, (size_t)256 << 20);
            char* slab[4];
            for (auto& h : slab)
                hipHostMalloc((void**)&h, (size_t)16 << 20, hipHostMallocDefault);
            for (size_t i = 0; !copies_stop.load(); i++)
                {
                hipMemcpyAsync(slab[i % 4], dsrc + ((i % 16) << 24), (size_t)16 << 20, hipMemcpyDeviceToHost, cs);
                if (i % 4 == 3)
                    hipStreamSynchronize(cs);
                }
            hipStreamSynchronize(cs);
        });
    std::this_thread::sleep_for(std::chrono::milliseconds(50));
    const int K = 4000;
    std::vector<hipEvent_t> ev(K);
    for (auto& e : ev)
        hipEventCreate(&e);
    for (int mode = 0; mode < 5; mode++) // 0: nobody allocates, 1: a helper thread does, 2: the launching thread does,
                                         // 3 / 4: the same with hipHostMalloc of a 16 MiB pinned slab
        {
        std::atomic<int> stop {0};
        std::vector<double> alloc_ms;
        std::thread helper;
        if (mode == 1 || mode == 3)
            helper = std::thread(
                [&]
                {
                    hipSetDevice(0);
                    std::vector<void*> got;
                    while (!stop.load())
                        {
                        std::this_thread::sleep_for(std::chrono::milliseconds(3));
                        void* q = nullptr;
                        const double t0 = now_us();
                        if ((mode == 1 ? hipMalloc(&q, (size_t)256 << 20) : hipHostMalloc(&q, (size_t)16 << 20, hipHostMallocDefault))
                            == hipSuccess)
                            got.push_back(q);
                        alloc_ms.push_back((now_us() - t0) / 1e3);
                        }
                    for (void* q : got)
                        mode == 1 ? hipFree(q) : hipHostFree(q);
                });
        std::vector<void*> mine;
        double launch_max = 0;
        for (int k = 0; k < K; k++)
            {
            const double t0 = now_us();
            if ((mode == 2 || mode == 4) && k % 400 == 200)
                {
                void* q = nullptr;
                const double a0 = now_us();
                if ((mode == 2 ? hipMalloc(&q, (size_t)256 << 20) : hipHostMalloc(&q, (size_t)16 << 20, hipHostMallocDefault)) == hipSuccess)
                    mine.push_back(q);
                alloc_ms.push_back((now_us() - a0) / 1e3);
                }
            hipLaunchKernelGGL(busy, dim3((unsigned)(n * 4 / 256)), dim3(256), 0, s, p, n * 4);
            hipEventRecord(ev[k], s);
            launch_max = std::max(launch_max, now_us() - t0);
            if (k % 64 == 63) // keep the queue short: the stream must never run dry because of the host
                hipEventSynchronize(ev[k - 32]);
            }
        hipStreamSynchronize(s);
        stop.store(1);
        if (helper.joinable())
            helper.join();
        std::vector<float> gaps;
        for (int k = 1; k < K; k++)
            {
            float ms = 0;
            hipEventElapsedTime(&ms, ev[k - 1], ev[k]);
            gaps.push_back(ms * 1e3f);
            }
        std::sort(gaps.begin(), gaps.end());
        double amax = 0, asum = 0;
        for (double a : alloc_ms)
            {
            amax = std::max(amax, a);
            asum += a;
            }
        printf("{\"lab\": \"malloc_beside_stream\", \"who_allocates\": \"%s\", \"allocations\": %zu, \"alloc_ms_mean\": %.2f, "
               "\"alloc_ms_max\": %.2f, \"kernel_period_us_median\": %.1f, \"p99\": %.1f, \"max\": %.1f, \"host_launch_us_max\": %.1f}\n",
               mode == 0   ? "nobody"
               : mode == 1 ? "a helper thread, hipMalloc 256 MiB"
               : mode == 2 ? "the launching thread, hipMalloc 256 MiB"
               : mode == 3 ? "a helper thread, hipHostMalloc 16 MiB"
                           : "the launching thread, hipHostMalloc 16 MiB", alloc_ms.size(),
               alloc_ms.empty() ? 0.0 : asum / alloc_ms.size(), amax, gaps[gaps.size() / 2], gaps[gaps.size() * 99 / 100], gaps.back(),
               launch_max);
        for (void* q : mine)
            mode == 2 ? hipFree(q) : hipHostFree(q);
        }
    copies_stop.store(1);
    copier.join();
    return 0;
    }
