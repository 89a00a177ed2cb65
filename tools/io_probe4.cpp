// io_probe4.cpp -- can a threaded fallocate -> mmap/populate -> memcpy -> munmap pipeline beat one
// pwrite thread on a tmpfs file?  Pieces of 16 MiB flow through the stages. Diagnostic tool.
// g++ -O2 -o io_probe4 io_probe4.cpp -pthread
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fcntl.h>
#include <mutex>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
template <class T> struct Chan {
    std::deque<T> q; std::mutex m; std::condition_variable cv; bool closed = false;
    void push(T v) { { std::lock_guard<std::mutex> g(m); q.push_back(v); } cv.notify_one(); }
    void close() { { std::lock_guard<std::mutex> g(m); closed = true; } cv.notify_all(); }
    bool pop(T& v) { std::unique_lock<std::mutex> lk(m); cv.wait(lk, [&] { return closed || !q.empty(); }); if (q.empty()) return false; v = q.front(); q.pop_front(); return true; }
};
struct Piece { long long off; size_t n; char* map; };
int main(int argc, char** argv) {
    const char* dir = argc > 1 ? argv[1] : "/dev/shm";
    const size_t piece = (size_t)16 << 20, frame = (size_t)280 << 20;
    const int frames = 6;
    char path[512]; snprintf(path, sizeof path, "%s/io_probe4_%d.bin", dir, (int)getpid());
    char* src = (char*)aligned_alloc(4096, frame); memset(src, 0x5a, frame);
    for (int mode = 0; mode < 4; mode++) {
        int fd = open(path, O_RDWR | O_CREAT | O_TRUNC, 0644);
        double t0 = now();
        if (mode == 0) {                       // baseline: one pwrite thread
            for (int f = 0; f < frames; f++) for (size_t o = 0; o < frame; o += piece) { size_t n = std::min(piece, frame - o); pwrite(fd, src + o, n, (long long)f * frame + o); }
        } else {
            // mode 1: falloc | populate | memcpy+munmap      (3 threads)
            // mode 2: falloc | populate x2 | memcpy | munmap  (5 threads)
            // mode 3: falloc | memcpy (faulting, no populate) x2 | munmap
            Chan<Piece> c1, c2, c3;
            int npop = mode == 2 ? 2 : (mode == 3 ? 0 : 1);
            std::thread tf([&] { for (int f = 0; f < frames; f++) for (size_t o = 0; o < frame; o += piece) { size_t n = std::min(piece, frame - o); long long off = (long long)f * frame + o; fallocate(fd, 0, off, n); c1.push({off, n, nullptr}); } c1.close(); });
            std::vector<std::thread> tp; std::atomic<int> live(npop);
            for (int i = 0; i < npop; i++) tp.emplace_back([&] { Piece p; while (c1.pop(p)) { p.map = (char*)mmap(NULL, p.n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, p.off); madvise(p.map, p.n, MADV_POPULATE_WRITE); c2.push(p); } if (--live == 0) c2.close(); });
            int ncopy = mode == 3 ? 2 : 1; std::atomic<int> livec(ncopy);
            std::vector<std::thread> tc;
            for (int i = 0; i < ncopy; i++) tc.emplace_back([&] { Piece p; Chan<Piece>& in = (mode == 3) ? c1 : c2; while (in.pop(p)) { if (!p.map) p.map = (char*)mmap(NULL, p.n, PROT_READ | PROT_WRITE, MAP_SHARED, fd, p.off); memcpy(p.map, src + (p.off % frame), p.n); if (mode == 1) munmap(p.map, p.n); else c3.push(p); } if (--livec == 0) c3.close(); });
            std::thread tu([&] { if (mode == 1) return; Piece p; while (c3.pop(p)) munmap(p.map, p.n); });
            tf.join(); for (auto& t : tp) t.join(); for (auto& t : tc) t.join(); tu.join();
        }
        double dt = now() - t0;
        printf("mode %d: %.2f GB/s\n", mode, (double)frames * frame / dt / 1e9);
        close(fd); unlink(path);
    }
    return 0;
}
