// pgsd_io.cpp -- positional file IO at the offsets of the reference's MPI-IO calls
// (MPI_File_write_at pgsd.c:2229/1154/2032, MPI_File_read_at pgsd.c:651/1559/2534).
// Each rank writes its own byte range at the offset the reference computes, through one of two back ends behind the
// io_* functions below: POSIX pwrite / pread (the default), or -- PGSD_IO=mpiio -- the reference's own
// MPI_File_write_at / MPI_File_read_at, reached through the optional plugin libpgsd_amd_mpiio.so
// (pgsd_mpiio_plugin.h; the library itself does not link MPI).  Same bytes at the same offsets either way.
// The small thread pool here carries the device pipeline's writer and reader threads.  For
// WRITES one thread per file is the measured optimum on the target boxes (buffered writes
// serialise on the file's inode lock: profiles/r01_io_probe*.log); page-cache READS take no
// exclusive lock and scale with threads.
#include "pgsd_internal.hpp"
#include "pgsd_mpiio_plugin.h"

#include <atomic>
#include <cctype>
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <pthread.h>
#include <sched.h>
#include <string>
#include <cstdlib>
#include <sys/file.h>
#include <sys/uio.h>
#include <condition_variable>
#include <deque>
#include <dlfcn.h>
#include <fcntl.h>
#include <map>
#include <sys/stat.h>
#include <functional>
#include <mutex>
#include <thread>
#include <unistd.h>

namespace pgsd_amd
    {
// ------------------------------------------------------------------ file back ends
namespace
    {
struct MpiFiles
    {
    std::mutex lock; // serialises every call into the plugin: MPI_THREAD_SERIALIZED is all the caller must provide
    int mode = -1;   // -1: PGSD_IO not looked at yet, 0: POSIX, 1: MPI-IO
    bool loaded = false;
    pgsd_mpiio_api api {};
    std::map<int, void*> files; // descriptor -> the plugin's file handle
    };
MpiFiles g_mpi;
std::atomic<int> g_mpi_open {0}; // number of MPI-IO files: the POSIX back end never takes the lock

// the plugin lies next to this library unless PGSD_MPIIO_LIBRARY names another build (another MPI)
bool load_mpiio_plugin()
    {
    if (g_mpi.loaded)
        return true;
    std::string path;
    if (const char* e = getenv("PGSD_MPIIO_LIBRARY"))
        path = e;
    else
        {
        Dl_info info;
        if (dladdr((void*)&load_mpiio_plugin, &info) && info.dli_fname)
            {
            path = info.dli_fname;
            const size_t slash = path.rfind('/');
            path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + "libpgsd_amd_mpiio.so";
            }
        else
            path = "libpgsd_amd_mpiio.so";
        }
    void* lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!lib)
        {
        set_last_error(std::string("PGSD_IO=mpiio: cannot load the MPI-IO back end (build it with `make -C pgsd-sph_amd/csrc mpiio`): ")
                       + dlerror());
        return false;
        }
    auto entry = (int (*)(pgsd_mpiio_api*))dlsym(lib, "pgsd_mpiio_plugin");
    if (!entry)
        {
        set_last_error("PGSD_IO=mpiio: " + path + " has no pgsd_mpiio_plugin");
        return false;
        }
    if (entry(&g_mpi.api) != 0)
        {
        set_last_error(std::string("PGSD_IO=mpiio: ") + (g_mpi.api.last_error ? g_mpi.api.last_error() : "the back end refused"));
        return false;
        }
    g_mpi.loaded = true;
    return true;
    }

// run f(plugin handle of fd) under the plugin lock when fd is an MPI-IO file; false: fd is a POSIX file
template<class F> bool with_mpi_file(int fd, F f)
    {
    if (g_mpi_open.load(std::memory_order_acquire) == 0)
        return false;
    std::lock_guard<std::mutex> guard(g_mpi.lock);
    auto it = g_mpi.files.find(fd);
    if (it == g_mpi.files.end())
        return false;
    f(it->second);
    return true;
    }
    } // namespace

const char* io_backend_name()
    {
    std::lock_guard<std::mutex> guard(g_mpi.lock);
    if (g_mpi.mode < 0)
        {
        const char* e = getenv("PGSD_IO");
        g_mpi.mode = (e && strcmp(e, "mpiio") == 0) ? 1 : 0;
        if (e && *e && strcmp(e, "mpiio") != 0 && strcmp(e, "posix") != 0)
            fprintf(stderr, "pgsd_amd: PGSD_IO=%s is neither posix nor mpiio: posix\n", e);
        }
    return g_mpi.mode == 1 ? "mpiio" : "posix";
    }

// open(2); with PGSD_IO=mpiio the file is ALSO opened through MPI-IO and every later io_* call on the descriptor
// goes there (the descriptor stays the file's identity and what the advisory write lock is taken on)
int io_open(const char* path, int oflags, int mode)
    {
    const int fd = open(path, oflags, mode);
    if (fd < 0 || strcmp(io_backend_name(), "mpiio") != 0)
        return fd;
    std::lock_guard<std::mutex> guard(g_mpi.lock);
    void* fh = nullptr;
    if (!load_mpiio_plugin() || g_mpi.api.open(path, (oflags & O_ACCMODE) == O_RDONLY, &fh) != 0)
        {
        if (g_mpi.loaded)
            set_last_error(std::string("PGSD_IO=mpiio: ") + g_mpi.api.last_error());
        close(fd);
        errno = EIO;
        return -1;
        }
    g_mpi.files[fd] = fh;
    g_mpi_open.fetch_add(1, std::memory_order_release);
    return fd;
    }

int io_close(int fd)
    {
    int rc = 0;
    bool mpi = false;
        {
        std::lock_guard<std::mutex> guard(g_mpi.lock);
        auto it = g_mpi.files.find(fd);
        if (it != g_mpi.files.end())
            {
            mpi = true;
            rc = g_mpi.api.close(it->second);
            g_mpi.files.erase(it);
            g_mpi_open.fetch_sub(1, std::memory_order_release);
            }
        }
    const int crc = close(fd);
    if (mpi && rc != 0)
        {
        errno = EIO;
        return -1;
        }
    return crc;
    }

int io_truncate(int fd, long long size)
    {
    int rc = 0;
    if (with_mpi_file(fd, [&](void* fh) { rc = g_mpi.api.set_size(fh, size); }))
        {
        if (rc != 0)
            errno = EIO;
        return rc;
        }
    return ftruncate(fd, (off_t)size);
    }

long long io_file_size(int fd)
    {
    long long size = -1;
    if (with_mpi_file(fd, [&](void* fh) { size = g_mpi.api.get_size(fh); }))
        {
        if (size < 0)
            errno = EIO;
        return size;
        }
    struct stat st;
    if (fstat(fd, &st) != 0)
        return -1;
    return (long long)st.st_size;
    }

// pwrite(2) / pread(2) semantics (bytes moved, -1 + errno) on either back end
ssize_t io_pwrite(int fd, const void* buf, size_t bytes, long long offset)
    {
    long long n = -1;
    if (with_mpi_file(fd, [&](void* fh) { n = g_mpi.api.write_at(fh, offset, buf, (long long)bytes); }))
        {
        if (n < 0)
            {
            set_last_error(std::string("PGSD_IO=mpiio: ") + g_mpi.api.last_error());
            errno = EIO;
            }
        return (ssize_t)n;
        }
    return pwrite(fd, buf, bytes, (off_t)offset);
    }

ssize_t io_pread(int fd, void* buf, size_t bytes, long long offset)
    {
    long long n = -1;
    if (with_mpi_file(fd, [&](void* fh) { n = g_mpi.api.read_at(fh, offset, buf, (long long)bytes); }))
        {
        if (n < 0)
            errno = EIO;
        return (ssize_t)n;
        }
    return pread(fd, buf, bytes, (off_t)offset);
    }

int pwrite_full(int fd, const void* buf, size_t bytes, long long offset)
    {
    const char* p = (const char*)buf;
    while (bytes > 0)
        {
        ssize_t w = io_pwrite(fd, p, bytes, offset);
        if (w < 0)
            {
            if (errno == EINTR)
                continue;
            return -errno;
            }
        p += w;
        offset += w;
        bytes -= (size_t)w;
        }
    return 0;
    }

// Buffered writes to one file serialise on its inode lock inside the kernel; when several
// PROCESSES (ranks) write the same file at once the lock convoy makes the aggregate rate drop
// well below a single writer's (profiles/r01_io_probe3.log, bench rehearsal: 7.0 -> 4.1 -> 3.6
// GB/s for 1/2/4 ranks on tmpfs).  Taking an advisory flock around each piece lets the ranks
// queue asleep instead, which keeps the file at the single-writer rate.
static int g_write_lock = -1;

int pwrite_locked(int fd, const void* buf, size_t bytes, long long offset, bool shared_file)
    {
    if (g_write_lock < 0)
        {
        const char* e = getenv("PGSD_WRITE_LOCK");
        g_write_lock = (e && atoi(e) == 0) ? 0 : 1;
        }
    if (!shared_file || !g_write_lock)
        return pwrite_full(fd, buf, bytes, offset);
    while (flock(fd, LOCK_EX) != 0)
        if (errno != EINTR)
            return pwrite_full(fd, buf, bytes, offset); // no lock support: write anyway
    int rc = pwrite_full(fd, buf, bytes, offset);
    flock(fd, LOCK_UN);
    return rc;
    }

// Several buffers, one contiguous file range, one system call (the chunks of a small frame follow each other in
// the file but not necessarily in memory).  `iov` is consumed.
int pwritev_locked(int fd, struct iovec* iov, int n, long long offset, bool shared_file)
    {
    if (g_write_lock < 0)
        {
        const char* e = getenv("PGSD_WRITE_LOCK");
        g_write_lock = (e && atoi(e) == 0) ? 0 : 1;
        }
    bool locked = false;
    if (shared_file && g_write_lock)
        {
        locked = true;
        while (flock(fd, LOCK_EX) != 0)
            if (errno != EINTR)
                {
                locked = false; // no lock support: write anyway
                break;
                }
        }
    int rc = 0;
    if (g_mpi_open.load(std::memory_order_acquire) != 0 && with_mpi_file(fd, [](void*) { }))
        {
        // MPI-IO has no gather write: one MPI_File_write_at per buffer, in file order
        for (; n > 0 && rc == 0; iov++, n--)
            {
            rc = pwrite_full(fd, iov->iov_base, iov->iov_len, offset);
            offset += (long long)iov->iov_len;
            }
        n = 0;
        }
    while (n > 0)
        {
        ssize_t w = pwritev(fd, iov, n > 1024 ? 1024 : n, (off_t)offset);
        if (w < 0)
            {
            if (errno == EINTR)
                continue;
            rc = -errno;
            break;
            }
        offset += w;
        size_t left = (size_t)w;
        while (n > 0 && left >= iov->iov_len)
            {
            left -= iov->iov_len;
            iov++;
            n--;
            }
        if (n > 0 && left > 0)
            {
            iov->iov_base = (char*)iov->iov_base + left;
            iov->iov_len -= left;
            }
        else if (n > 0 && w == 0 && iov->iov_len > 0)
            {
            rc = -EIO; // no progress on a non-empty buffer
            break;
            }
        }
    if (locked)
        flock(fd, LOCK_UN);
    return rc;
    }

void pread_some(int fd, void* buf, size_t bytes, long long offset)
    {
    char* p = (char*)buf;
    while (bytes > 0)
        {
        ssize_t r = io_pread(fd, p, bytes, offset);
        if (r <= 0)
            {
            if (r < 0 && errno == EINTR)
                continue;
            return; // short read: the tail keeps its previous contents
            }
        p += r;
        offset += r;
        bytes -= (size_t)r;
        }
    }

// Large host reads (restart files read on the CPU side) are split over a few short-lived
// threads: page-cache reads take no exclusive lock and scale (profiles/r01_read_sweep.log).
void pread_parallel(int fd, void* buf, size_t bytes, long long offset)
    {
    const size_t min_piece = (size_t)32 << 20;
    unsigned hw = std::thread::hardware_concurrency();
    size_t n = bytes / min_piece;
    size_t cap = 8;
    if (const char* e = getenv("PGSD_HOST_READ_THREADS"))
        cap = (size_t)(atoi(e) > 0 ? atoi(e) : 1);
    if (n > cap)
        n = cap;
    if (hw && n > hw)
        n = hw;
    if (n < 2)
        {
        pread_some(fd, buf, bytes, offset);
        return;
        }
    const size_t piece = ((bytes + n - 1) / n + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    for (size_t off = piece; off < bytes; off += piece)
        {
        const size_t len = bytes - off < piece ? bytes - off : piece;
        th.emplace_back([=] { pread_some(fd, (char*)buf + off, len, offset + (long long)off); });
        }
    pread_some(fd, buf, piece < bytes ? piece : bytes, offset);
    for (auto& t : th)
        t.join();
    }

class WriterPool
    {
    public:
    explicit WriterPool(unsigned n, const cpu_set_t* cpus = nullptr) : m_stop(false)
        {
        if (n == 0)
            n = 1;
        for (unsigned i = 0; i < n; i++)
            {
            m_threads.emplace_back([this] { run(); });
            if (cpus) // keep the copy loops next to the memory they touch
                (void)pthread_setaffinity_np(m_threads.back().native_handle(), sizeof(cpu_set_t), cpus);
            }
        }

    ~WriterPool()
        {
            {
            std::lock_guard<std::mutex> g(m_mutex);
            m_stop = true;
            }
        m_cv.notify_all();
        for (auto& t : m_threads)
            t.join();
        }

    void submit(std::function<void()> fn)
        {
            {
            std::lock_guard<std::mutex> g(m_mutex);
            m_jobs.push_back(std::move(fn));
            }
        m_cv.notify_one();
        }

    unsigned size() const
        {
        return (unsigned)m_threads.size();
        }

    private:
    void run()
        {
        for (;;)
            {
            std::function<void()> fn;
                {
                std::unique_lock<std::mutex> lk(m_mutex);
                m_cv.wait(lk, [this] { return m_stop || !m_jobs.empty(); });
                if (m_jobs.empty())
                    return;
                fn = std::move(m_jobs.front());
                m_jobs.pop_front();
                }
            fn();
            }
        }

    std::vector<std::thread> m_threads;
    std::deque<std::function<void()>> m_jobs;
    std::mutex m_mutex;
    std::condition_variable m_cv;
    bool m_stop;
    };

WriterPool* writer_pool_create(unsigned n_threads, const cpu_set_t* cpus)
    {
    return new WriterPool(n_threads, cpus);
    }

// "0-7,64-71" -> the CPUs of the list that are also in `allowed`; returns how many
static int parse_cpulist(const char* buf, const cpu_set_t* allowed, cpu_set_t* out)
    {
    CPU_ZERO(out);
    int count = 0;
    for (const char* p = buf; *p;)
        {
        char* end;
        long a = strtol(p, &end, 10);
        if (end == p)
            break;
        long b = a;
        if (*end == '-')
            b = strtol(end + 1, &end, 10);
        for (long c = a; c <= b && c < CPU_SETSIZE; c++)
            if (CPU_ISSET((int)c, allowed))
                {
                CPU_SET((int)c, out);
                count++;
                }
        if (*end != ',')
            break;
        p = end + 1;
        }
    return count;
    }

// CPUs of the NUMA node a PCI device hangs off, intersected with what this process may run on.
// false when the node is unknown (-1), the machine has one node, or PGSD_NUMA=0.
bool numa_cpus_of_pci_device(const char* pci_bus_id, cpu_set_t* out)
    {
    if (const char* e = getenv("PGSD_NUMA"))
        if (atoi(e) == 0)
            return false;
    char path[256], buf[4096];
    std::string bdf(pci_bus_id);
    for (char& c : bdf)
        c = (char)tolower((unsigned char)c);
    snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bdf.c_str());
    FILE* f = fopen(path, "r");
    if (!f)
        return false;
    int node = -1;
    if (fscanf(f, "%d", &node) != 1)
        node = -1;
    fclose(f);
    if (node < 0)
        return false;
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    f = fopen(path, "r");
    if (!f)
        return false;
    size_t n = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[n] = 0;
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0)
        return false;
    const int count = parse_cpulist(buf, &allowed, out);
    int total = CPU_COUNT(&allowed);
    return count > 0 && count < total; // a single node (or all CPUs) needs no pinning
    }

void writer_pool_destroy(WriterPool* p)
    {
    delete p;
    }

void writer_pool_submit(WriterPool* p, std::function<void()> fn)
    {
    p->submit(std::move(fn));
    }

int writer_pool_pwrite_sync(WriterPool* pool, int fd, const void* buf, size_t bytes, long long offset,
                            bool shared_file)
    {
    const size_t piece = (size_t)8 << 20;
    if (!pool || bytes <= piece || shared_file)
        {
        // pieces keep the lock hold time bounded so that ranks interleave
        int rc = 0;
        for (size_t off = 0; off < bytes && rc == 0; off += piece)
            rc = pwrite_locked(fd, (const char*)buf + off, bytes - off < piece ? bytes - off : piece,
                               offset + (long long)off, shared_file);
        return rc;
        }
    struct Latch
        {
        std::mutex m;
        std::condition_variable cv;
        size_t pending;
        int err;
        } latch;
    size_t n_pieces = (bytes + piece - 1) / piece;
    latch.pending = n_pieces;
    latch.err = 0;
    for (size_t i = 0; i < n_pieces; i++)
        {
        size_t off = i * piece;
        size_t n = bytes - off < piece ? bytes - off : piece;
        const char* p = (const char*)buf + off;
        pool->submit(
            [&latch, fd, p, n, offset, off]
            {
                int e = pwrite_full(fd, p, n, offset + (long long)off);
                std::lock_guard<std::mutex> g(latch.m);
                if (e != 0 && latch.err == 0)
                    latch.err = e;
                if (--latch.pending == 0)
                    latch.cv.notify_all();
            });
        }
    std::unique_lock<std::mutex> lk(latch.m);
    latch.cv.wait(lk, [&latch] { return latch.pending == 0; });
    return latch.err;
    }
    } // namespace pgsd_amd
