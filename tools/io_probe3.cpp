// io_probe3.cpp -- do P processes writing disjoint ranges of ONE tmpfs file scale?
// Compares pwrite (inode lock) with mmap + populate + memcpy (fault path). Diagnostic tool.
// g++ -O2 -o io_probe3 io_probe3.cpp -pthread
#include <chrono>
#include <initializer_list>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/dev/shm";
    size_t bytes = (size_t)280 << 20;
    int frames = 4;
    char path[512]; snprintf(path, sizeof path, "%s/io_probe3_%d.bin", dir, (int)getpid());
    for (int mode = 0; mode < 3; mode++)
    for (int P : {1, 2, 4, 8}) {
        int fd0 = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644); close(fd0);
        double t0 = now();
        for (int r = 0; r < P; r++) if (fork() == 0) {
            int fd = open(path, O_RDWR);
            char* src = (char*)malloc(bytes); memset(src, r + 1, bytes);
            for (int f = 0; f < frames; f++) {
                long long off = ((long long)f * P + r) * (long long)bytes;
                if (mode == 0) { size_t d = 0; while (d < bytes) { ssize_t w = pwrite(fd, src + d, bytes - d, off + d); if (w <= 0) break; d += w; } }
                else {
                    if (mode == 2) fallocate(fd, 0, off, bytes);
                    else { struct stat st; fstat(fd, &st); if (st.st_size < off + (long long)bytes) fallocate(fd, 0, off + bytes - 4096, 4096); }
                    char* m = (char*)mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, off);
                    madvise(m, bytes, MADV_POPULATE_WRITE);
                    memcpy(m, src, bytes);
                    munmap(m, bytes);
                }
            }
            _exit(0);
        }
        for (int r = 0; r < P; r++) wait(NULL);
        double dt = now() - t0;
        // subtract nothing: includes the malloc+memset of the source (same for all modes)
        printf("%s P=%d: %.2f GB/s aggregate (%.0f ms for %d frames)\n", mode == 0 ? "pwrite          " : mode == 1 ? "mmap fault-alloc" : "fallocate+mmap  ", P,
               (double)P * frames * bytes / dt / 1e9, dt * 1e3, frames);
        unlink(path);
    }
    return 0;
}
