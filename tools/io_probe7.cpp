// io_probe7.cpp -- can a long-lived MAP_SHARED window beat pwrite into fresh tmpfs pages?
//
// Diagnostic tool (not part of the product):  g++ -O2 -pthread -o io_probe7 io_probe7.cpp
//
// Facts from the earlier probes (profiles/r01_io_probe*.log): fallocate builds tmpfs pages at ~19 GB/s on one
// thread, pwrite into existing pages runs at ~8.8 GB/s and into fresh ones at 5.5-7 GB/s; both take the inode lock,
// so they do not overlap.  A memcpy through a mapping takes no inode lock: its minor faults map pages that already
// exist.  This probe keeps ONE mapping for the whole run (no munmap, no populate, no registration), lets one thread
// fallocate the extent of frame i+1 while T threads memcpy frame i into the window, and compares with the single
// pwrite thread the product uses.
//
//   io_probe7 [frame_MiB=267] [frames=20] [dir=/dev/shm]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>

static double now()
    {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }

static void copy_parallel(char* dst, const char* src, size_t bytes, int T, size_t block)
    {
    // blocks of `block` bytes are dealt round-robin: every thread streams through its own 2 MiB page-table pages
    std::atomic<size_t> next(0);
    const size_t nblocks = (bytes + block - 1) / block;
    auto work = [&]
    {
        for (;;)
            {
            size_t b = next.fetch_add(1);
            if (b >= nblocks)
                return;
            size_t off = b * block, n = std::min(block, bytes - off);
            memcpy(dst + off, src + off, n);
            }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; t++)
        th.emplace_back(work);
    work();
    for (auto& x : th)
        x.join();
    }

int main(int argc, char** argv)
    {
    const size_t frame = (size_t)(argc > 1 ? atol(argv[1]) : 267) << 20;
    const int frames = argc > 2 ? atoi(argv[2]) : 20;
    const std::string dir = argc > 3 ? argv[3] : "/dev/shm";
    const std::string path = dir + "/pgsd_io_probe7_" + std::to_string(getpid());
    char* src = (char*)malloc(frame);
    for (size_t i = 0; i < frame; i += 8)
        *(uint64_t*)(src + i) = i * 0x9E3779B97F4A7C15ull;
    const size_t total = frame * (size_t)frames;

    // ---- baseline: one pwrite thread, 16 MiB calls (the product's writer)
    for (int rep = 0; rep < 2; rep++)
        {
        int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
        double t0 = now();
        for (int f = 0; f < frames; f++)
            for (size_t o = 0; o < frame; o += 16u << 20)
                {
                size_t n = std::min((size_t)16 << 20, frame - o);
                if (pwrite(fd, src + o, n, (off_t)((size_t)f * frame + o)) != (ssize_t)n)
                    {
                    perror("pwrite");
                    return 1;
                    }
                }
        double dt = now() - t0;
        printf("pwrite, one thread, fresh pages          %6.2f GB/s\n", total / dt / 1e9);
        close(fd);
        unlink(path.c_str());
        }

    // ---- mapping window, with and without fallocate running ahead
    for (int ahead = 1; ahead >= 0; ahead--)
        for (int T : {1, 2, 4, 8, 16})
            for (size_t block : {(size_t)2 << 20, (size_t)16 << 20})
                {
                int fd = open(path.c_str(), O_CREAT | O_TRUNC | O_RDWR, 0644);
                if (!ahead && ftruncate(fd, (off_t)total) != 0)
                    {
                    perror("ftruncate");
                    return 1;
                    }
                char* win = (char*)mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
                if (win == MAP_FAILED)
                    {
                    perror("mmap");
                    return 1;
                    }
                double t0 = now(), t_falloc = 0;
                std::atomic<int> ready(0);
                std::thread alloc;
                if (ahead)
                    alloc = std::thread(
                        [&]
                        {
                            for (int f = 0; f < frames; f++)
                                {
                                double a = now();
                                if (posix_fallocate(fd, (off_t)((size_t)f * frame), (off_t)frame) != 0)
                                    {
                                    perror("fallocate");
                                    exit(1);
                                    }
                                t_falloc += now() - a;
                                ready.store(f + 1, std::memory_order_release);
                                }
                        });
                for (int f = 0; f < frames; f++)
                    {
                    while (ahead && ready.load(std::memory_order_acquire) <= f)
                        std::this_thread::yield();
                    copy_parallel(win + (size_t)f * frame, src, frame, T, block);
                    }
                double dt = now() - t0;
                if (ahead)
                    alloc.join();
                double t1 = now();
                munmap(win, total);
                double t_unmap = now() - t1;
                // the bytes are in the file, not only in the window
                char probe[64];
                bool ok = pread(fd, probe, 64, (off_t)(total - frame + 4096)) == 64 && memcmp(probe, src + 4096, 64) == 0;
                printf("window %s T=%2d block %2zu MiB  %6.2f GB/s   (fallocate busy %4.0f %%, final munmap %.0f ms, check %s)\n",
                       ahead ? "+ fallocate ahead" : "faulting alloc   ", T, block >> 20, total / dt / 1e9,
                       100.0 * t_falloc / dt, t_unmap * 1e3, ok ? "ok" : "BAD");
                close(fd);
                unlink(path.c_str());
                }
    free(src);
    return 0;
    }
