// io_probe2.cpp -- cost of the mmap-window write path on tmpfs: fallocate -> mmap/populate ->
// parallel memcpy (or direct DMA into a registered window). Diagnostic tool.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <functional>
#include <sys/mman.h>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void par(int T, size_t bytes, const std::function<void(size_t, size_t)>& fn) {
    std::vector<std::thread> th;
    size_t piece = (bytes / T + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; t++) { size_t off = (size_t)t * piece; if (off >= bytes) break; size_t n = std::min(piece, bytes - off); th.emplace_back([=, &fn] { fn(off, n); }); }
    for (auto& x : th) x.join();
}
int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/dev/shm";
    size_t bytes = (size_t)280 << 20;
    char path[512]; snprintf(path, sizeof path, "%s/io_probe2_%d.bin", dir, (int)getpid());
    char* host; (void)hipHostMalloc((void**)&host, bytes, hipHostMallocDefault); memset(host, 0x5a, bytes);
    char* dev; (void)hipMalloc((void**)&dev, bytes); (void)hipMemset(dev, 0x3c, bytes); (void)hipDeviceSynchronize();
    int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
    long long off = 0;
    for (int rep = 0; rep < 2; rep++)
    for (int T : {1, 4, 8, 16}) {
        double t0 = now(); fallocate(fd, 0, off, bytes); double t1 = now();
        char* m = (char*)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, off); double t2 = now();
        par(T, bytes, [&](size_t o, size_t n) { madvise(m + o, n, MADV_POPULATE_WRITE); }); double t3 = now();
        par(T, bytes, [&](size_t o, size_t n) { memcpy(m + o, host + o, n); }); double t4 = now();
        munmap(m, bytes); double t5 = now();
        printf("T=%2d fallocate %.1f ms | mmap %.2f ms | populate %.1f ms (%.1f GB/s) | memcpy %.1f ms (%.1f GB/s) | munmap %.1f ms | total %.2f GB/s\n", T,
               (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, bytes / (t3 - t2) / 1e9, (t4 - t3) * 1e3, bytes / (t4 - t3) / 1e9, (t5 - t4) * 1e3, bytes / (t5 - t0) / 1e9);
        off += bytes;
    }
    // no explicit populate: memcpy faults in the preallocated pages
    for (int T : {1, 8, 16}) {
        fallocate(fd, 0, off, bytes);
        char* m = (char*)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, off);
        double t0 = now(); par(T, bytes, [&](size_t o, size_t n) { memcpy(m + o, host + o, n); }); double t1 = now();
        munmap(m, bytes);
        printf("T=%2d memcpy into prealloc (faulting) %.1f ms (%.1f GB/s)\n", T, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        off += bytes;
    }
    // populate without fallocate (allocation inside the fault path), T threads
    for (int T : {1, 8, 16}) {
        ftruncate(fd, off + bytes);
        char* m = (char*)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, off);
        double t0 = now(); par(T, bytes, [&](size_t o, size_t n) { madvise(m + o, n, MADV_POPULATE_WRITE); }); double t1 = now();
        munmap(m, bytes);
        printf("T=%2d populate-alloc (no fallocate) %.1f ms (%.1f GB/s)\n", T, (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        off += bytes;
    }
    // registered window: fallocate + mmap + populate + hipHostRegister, then DMA
    for (int T : {1, 8}) {
        fallocate(fd, 0, off, bytes);
        char* m = (char*)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, off);
        par(T, bytes, [&](size_t o, size_t n) { madvise(m + o, n, MADV_POPULATE_WRITE); });
        double t0 = now(); hipError_t e = hipHostRegister(m, bytes, hipHostRegisterDefault); double t1 = now();
        if (e != hipSuccess) { printf("register failed %s\n", hipGetErrorString(e)); (void)hipGetLastError(); munmap(m, bytes); continue; }
        (void)hipMemcpy(m, dev, bytes, hipMemcpyDeviceToHost); double t2 = now();
        (void)hipHostUnregister(m); double t3 = now(); munmap(m, bytes);
        printf("register(populated) %.1f ms (%.1f GB/s) | DMA %.1f ms (%.1f GB/s) | unregister %.1f ms\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9, (t2 - t1) * 1e3, bytes / (t2 - t1) / 1e9, (t3 - t2) * 1e3);
        // chunked registration in parallel threads
        off += bytes;
    }
    {
        fallocate(fd, 0, off, bytes);
        char* m = (char*)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, off);
        par(8, bytes, [&](size_t o, size_t n) { madvise(m + o, n, MADV_POPULATE_WRITE); });
        double t0 = now();
        par(8, bytes, [&](size_t o, size_t n) { (void)hipHostRegister(m + o, n, hipHostRegisterDefault); });
        double t1 = now();
        printf("register 8 threads x 35MB: %.1f ms (%.1f GB/s)\n", (t1 - t0) * 1e3, bytes / (t1 - t0) / 1e9);
        (void)hipMemcpy(m, dev, bytes, hipMemcpyDeviceToHost);
        double t2 = now();
        printf("DMA into 8 registered pieces with one memcpy: %.1f ms\n", (t2 - t1) * 1e3);
    }
    // two processes' worth of contention is out of scope here
    close(fd); unlink(path);
    return 0;
}
