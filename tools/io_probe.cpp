// io_probe.cpp -- measure ways of landing 280 MB of pinned-host / device bytes in a tmpfs file.
// Diagnostic tool (not part of the product): hipcc -O2 -o io_probe io_probe.cpp -pthread
#include <hip/hip_runtime.h>
#include <functional>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
#include <linux/falloc.h>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par(int T, size_t bytes, const std::function<void(size_t, size_t)>& fn) {
    std::vector<std::thread> th;
    size_t piece = (bytes / T + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; t++) {
        size_t off = (size_t)t * piece;
        if (off >= bytes) break;
        size_t n = std::min(piece, bytes - off);
        th.emplace_back([=, &fn] { fn(off, n); });
    }
    for (auto& x : th) x.join();
}

int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/dev/shm";
    size_t bytes = (size_t)280 << 20;
    char path[512];
    snprintf(path, sizeof path, "%s/io_probe_%d.bin", dir, (int)getpid());
    char* host;
    hipHostMalloc((void**)&host, bytes, hipHostMallocDefault);
    memset(host, 0x5a, bytes);
    char* dev;
    hipMalloc((void**)&dev, bytes);
    hipMemset(dev, 0x3c, bytes);
    hipDeviceSynchronize();
    system("cat /sys/kernel/mm/transparent_hugepage/shmem_enabled; grep -E 'shm|tmpfs' /proc/mounts | head -5; nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null");

    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    long long off = 0;
    int Ts[] = {1, 2, 4, 8, 16, 32, 64};
    // 1. pwrite into fresh pages
    for (int T : Ts) {
        double t0 = now();
        par(T, bytes, [&](size_t o, size_t n) { size_t d = 0; while (d < n) { ssize_t w = pwrite(fd, host + o + d, n - d, off + o + d); if (w <= 0) break; d += w; } });
        double dt = now() - t0;
        printf("pwrite fresh   T=%2d  %.2f GB/s\n", T, bytes / dt / 1e9);
        off += bytes;
    }
    // 2. fallocate, then pwrite into allocated pages
    {
        double t0 = now();
        fallocate(fd, 0, off, bytes);
        double dt = now() - t0;
        printf("fallocate 1 thread      %.2f GB/s\n", bytes / dt / 1e9);
        for (int T : {1, 4, 8, 16, 32}) {
            t0 = now();
            par(T, bytes, [&](size_t o, size_t n) { size_t d = 0; while (d < n) { ssize_t w = pwrite(fd, host + o + d, n - d, off + o + d); if (w <= 0) break; d += w; } });
            dt = now() - t0;
            printf("pwrite prealloc T=%2d  %.2f GB/s\n", T, bytes / dt / 1e9);
        }
        off += bytes;
        for (int T : {4, 16}) {
            t0 = now();
            par(T, bytes, [&](size_t o, size_t n) { fallocate(fd, 0, off + o, n); });
            dt = now() - t0;
            printf("fallocate T=%2d          %.2f GB/s\n", T, bytes / dt / 1e9);
            off += bytes;
        }
    }
    // 3. mmap + memcpy into fresh pages (with and without MADV_HUGEPAGE)
    for (int huge = 0; huge < 2; huge++)
        for (int T : {1, 8, 16, 32}) {
            ftruncate(fd, off + bytes);
            double t0 = now();
            char* m = (char*)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, off);
            if (huge) madvise(m, bytes, MADV_HUGEPAGE);
            par(T, bytes, [&](size_t o, size_t n) { memcpy(m + o, host + o, n); });
            munmap(m, bytes);
            double dt = now() - t0;
            printf("mmap memcpy huge=%d T=%2d  %.2f GB/s\n", huge, T, bytes / dt / 1e9);
            off += bytes;
        }
    // 4. hipHostRegister an mmap'd fresh region and copy device -> file pages directly
    for (int pre = 0; pre < 2; pre++) {
        ftruncate(fd, off + bytes);
        if (pre) fallocate(fd, 0, off, bytes);
        double t0 = now();
        char* m = (char*)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, off);
        hipError_t e = hipHostRegister(m, bytes, hipHostRegisterDefault);
        double t1 = now();
        if (e != hipSuccess) { printf("hipHostRegister(mmap tmpfs) failed: %s\n", hipGetErrorString(e)); munmap(m, bytes); (void)hipGetLastError(); continue; }
        hipMemcpy(m, dev, bytes, hipMemcpyDeviceToHost);
        double t2 = now();
        hipHostUnregister(m);
        munmap(m, bytes);
        double t3 = now();
        printf("register(prealloc=%d) %.1f ms, D2H direct %.1f ms (%.1f GB/s), unregister %.1f ms; total %.2f GB/s\n", pre, (t1 - t0) * 1e3,
               (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3, bytes / (t3 - t0) / 1e9);
        off += bytes;
    }
    // 5. plain D2H into pinned slab (reference speed), and pageable
    {
        double t0 = now();
        hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost);
        double dt = now() - t0;
        printf("D2H pinned %.1f GB/s\n", bytes / dt / 1e9);
        t0 = now();
        for (int i = 0; i < 4; i++) hipMemcpyAsync(host + i * (bytes / 4), dev + i * (bytes / 4), bytes / 4, hipMemcpyDeviceToHost, 0);
        hipDeviceSynchronize();
        dt = now() - t0;
        printf("D2H pinned 4 pieces %.1f GB/s\n", bytes / dt / 1e9);
    }
    // 6. memcpy host->host bandwidth with T threads (upper bound for the page-cache copy)
    {
        char* dst = (char*)malloc(bytes);
        memset(dst, 1, bytes);
        for (int T : {1, 8, 16, 32}) {
            double t0 = now();
            par(T, bytes, [&](size_t o, size_t n) { memcpy(dst + o, host + o, n); });
            double dt = now() - t0;
            printf("memcpy pinned->heap T=%2d %.2f GB/s\n", T, bytes / dt / 1e9);
        }
        free(dst);
    }
    close(fd);
    unlink(path);
    return 0;
}
