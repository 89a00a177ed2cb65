// pgsd_comm.cpp -- process-wide default communicator and its host back ends.
//
// Replaces the reference's hard-coded MPI_COMM_WORLD (pgsd.c:106-202, 1748).  The only
// primitive the file layer needs is a small allgather; barrier and broadcast are built
// from it.  Back ends here: "self" (one rank) and "shm" (ranks of one node meeting in a
// /dev/shm segment: process-shared pthread barrier + one 4 KiB slot per rank).  The RCCL
// back end lives in pgsd_comm_rccl.cpp, host-callback back ends are installed with
// pgsd_comm_set_default().
#include "pgsd_internal.hpp"

#include <new>
#include <signal.h>
#include <stdexcept>

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

namespace pgsd_amd
    {
static thread_local std::string g_last_error;
static thread_local uint64_t g_last_error_serial = 0;

void set_last_error(const std::string& s)
    {
    g_last_error = s;
    g_last_error_serial++;
    }

uint64_t last_error_serial()
    {
    return g_last_error_serial;
    }

const char* last_error()
    {
    return g_last_error.c_str();
    }

int abi_guard() noexcept
    {
    try
        {
        throw; // the exception the entry point caught
        }
    catch (const std::bad_alloc&)
        {
        }
    catch (const std::length_error&)
        {
        }
    catch (const std::exception& e)
        {
        try
            {
            set_last_error(std::string("internal error: ") + e.what());
            }
        catch (...)
            {
            }
        return PGSD_ERROR_INVALID_ARGUMENT;
        }
    catch (...)
        {
        return PGSD_ERROR_INVALID_ARGUMENT;
        }
    // an allocation the request (or a damaged file) asked for cannot be made: the reference's
    // malloc-failure code (pgsd.c:1520)
    try
        {
        set_last_error("memory allocation failed");
        }
    catch (...)
        {
        }
    return PGSD_ERROR_MEMORY_ALLOCATION_FAILED;
    }

// ---------------------------------------------------------------- phase timeline (roctx)
namespace
    {
struct Roctx
    {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    bool on = false;
    Roctx()
        {
        const char* e = getenv("PGSD_TRACE");
        if (!e || !*e || atoi(e) == 0)
            return;
        void* lib = nullptr;
        for (const char* n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so",
                              "/opt/rocm/lib/librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"})
            if ((lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL)))
                break;
        if (!lib)
            return;
        push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
        pop = (int (*)())dlsym(lib, "roctxRangePop");
        on = push && pop;
        }
    };

const Roctx& roctx()
    {
    static const Roctx r;
    return r;
    }
    } // namespace

bool trace_on()
    {
    return roctx().on;
    }

void trace_push(const char* name)
    {
    (void)roctx().push(name);
    }

void trace_pop()
    {
    (void)roctx().pop();
    }

TraceRange::TraceRange(const char* fmt, unsigned long long a, unsigned long long b) : on(trace_on())
    {
    if (on)
        {
        char buf[160];
        snprintf(buf, sizeof(buf), fmt, a, b);
        trace_push(buf);
        }
    }

// ---------------------------------------------------------------- self
static int self_allgather(void*, const void* send, void* recv, size_t bytes)
    {
    if (send != recv)
        memcpy(recv, send, bytes);
    return 0;
    }

static pgsd_comm make_self()
    {
    pgsd_comm c;
    memset(&c, 0, sizeof(c));
    c.rank = 0;
    c.size = 1;
    c.allgather = self_allgather;
    return c;
    }

static std::shared_ptr<CommBox>& comm_slot()
    {
    // Leaked on purpose: a process that exits without pgsd_comm_finalize must not run a communicator's
    // destroy hook from static destruction -- librccl (dlopen'ed later than this library) has torn itself
    // down by then, ncclCommDestroy is not safe when ranks exit at different times, and the shm back end's
    // destroy hook is a barrier.  Teardown happens in pgsd_comm_finalize / pgsd_comm_init_* or when the last
    // handle that outlived its communicator is closed; at exit the OS takes the rest.
    static std::shared_ptr<CommBox>* box = new std::shared_ptr<CommBox>(std::make_shared<CommBox>(make_self()));
    return *box;
    }

std::shared_ptr<CommBox> default_comm_box()
    {
    return comm_slot();
    }

pgsd_comm default_comm()
    {
    return comm_slot()->c;
    }

// ---------------------------------------------------------------- shm
enum
    {
    SHM_SLOT_BYTES = 4096,
    SHM_MAGIC = 0x50475344 // "PGSD"
    };

enum
    {
    SHM_MAX_RANKS = 1024
    };

struct ShmSegment
    {
    volatile uint32_t ready;
    uint32_t size;
    volatile int32_t creator_pid; // rank 0's process: a segment whose creator is gone is a crashed run's
    // sense-reversing barrier on plain atomics: waiters poll (spin, then sleep), which lets them
    // notice a peer that died instead of waiting for it forever as a futex barrier (or MPI) would
    uint32_t arrived;
    uint32_t generation;
    uint32_t broken;
    int32_t pids[SHM_MAX_RANKS];
    uint64_t want[SHM_MAX_RANKS]; // bytes per rank of the allgather each rank is in: ranks out of step are told so
    char pad[64];
    // followed by size * SHM_SLOT_BYTES slot bytes
    };

struct ShmCtx
    {
    ShmSegment* seg;
    char* slots;
    size_t map_bytes;
    std::string name;
    int rank, size;
    bool told = false; // this rank has been given the reason why the communicator is broken
    };

// no such process, or a zombie nobody has reaped yet
static bool process_gone(int32_t pid)
    {
    if (kill((pid_t)pid, 0) != 0)
        return errno == ESRCH;
    char path[64], buf[512];
    snprintf(path, sizeof(path), "/proc/%d/stat", (int)pid);
    FILE* f = fopen(path, "r");
    if (!f)
        return false;
    size_t n = fread(buf, 1, sizeof(buf) - 1, f);
    fclose(f);
    buf[n] = 0;
    const char* close_paren = strrchr(buf, ')'); // "pid (comm) S ..."
    return close_paren && close_paren[1] == ' ' && (close_paren[2] == 'Z' || close_paren[2] == 'X');
    }

static bool shm_peer_gone(ShmCtx* c)
    {
    for (int r = 0; r < c->size; r++)
        {
        int32_t pid = __atomic_load_n(&c->seg->pids[r], __ATOMIC_ACQUIRE);
        if (pid > 0 && r != c->rank && process_gone(pid))
            return true;
        }
    return false;
    }

static int shm_barrier(void* p)
    {
    ShmCtx* c = (ShmCtx*)p;
    ShmSegment* seg = c->seg;
    if (__atomic_load_n(&seg->broken, __ATOMIC_ACQUIRE))
        {
        if (!c->told)
            set_last_error("the shm communicator is broken: a collective failed on another rank (ranks in different "
                           "collectives, or a rank gone)");
        c->told = true;
        return -1;
        }
    const uint32_t gen = __atomic_load_n(&seg->generation, __ATOMIC_ACQUIRE);
    if (__atomic_add_fetch(&seg->arrived, 1, __ATOMIC_ACQ_REL) == (uint32_t)c->size)
        {
        __atomic_store_n(&seg->arrived, 0, __ATOMIC_RELAXED);
        __atomic_add_fetch(&seg->generation, 1, __ATOMIC_RELEASE);
        return 0;
        }
    uint32_t spins = 0;
    long sleep_ns = 20000;
    double slept = 0, next_check = 0.25;
    while (__atomic_load_n(&seg->generation, __ATOMIC_ACQUIRE) == gen)
        {
        if (++spins < 2000)
            {
            __builtin_ia32_pause();
            continue;
            }
        struct timespec ts = {0, sleep_ns};
        nanosleep(&ts, NULL);
        slept += sleep_ns * 1e-9;
        if (sleep_ns < 1000000)
            sleep_ns *= 2;
        if (slept >= next_check)
            {
            next_check = slept + 0.25;
            if (__atomic_load_n(&seg->broken, __ATOMIC_ACQUIRE) || shm_peer_gone(c))
                {
                __atomic_store_n(&seg->broken, 1, __ATOMIC_RELEASE);
                set_last_error("a rank of the shm communicator is gone (process exited without pgsd_comm_finalize)");
                c->told = true;
                return -1;
                }
            }
        }
    return 0;
    }

static int shm_allgather(void* p, const void* send, void* recv, size_t bytes)
    {
    ShmCtx* c = (ShmCtx*)p;
    const char* s = (const char*)send;
    char* r = (char*)recv;
    // messages larger than a slot go in rounds
    size_t done = 0;
    __atomic_store_n(&c->seg->want[c->rank], (uint64_t)bytes, __ATOMIC_RELAXED);
    do
        {
        size_t n = bytes - done < (size_t)SHM_SLOT_BYTES ? bytes - done : (size_t)SHM_SLOT_BYTES;
        memcpy(c->slots + (size_t)c->rank * SHM_SLOT_BYTES, s + done, n);
        if (shm_barrier(p) != 0)
            return -1;
        if (done == 0)
            {
            // Every rank must be in the SAME collective: a rank that made a call the others did not (a collective
            // read on one rank only, say) would otherwise pair its message with a different exchange of theirs and
            // every rank would go on with the other's bytes.  All ranks see the same vector, so all fail alike.
            // (a rank that has found the mismatch already may be in its next call, its size overwritten: `broken`)
            for (int j = 0; j < c->size; j++)
                if (__atomic_load_n(&c->seg->want[j], __ATOMIC_RELAXED) != (uint64_t)bytes
                    || __atomic_load_n(&c->seg->broken, __ATOMIC_ACQUIRE))
                    {
                    c->told = true;
                    set_last_error("the ranks are in different collectives (message sizes differ): a collective call "
                                   "was made by some ranks only");
                    __atomic_store_n(&c->seg->broken, 1, __ATOMIC_RELEASE);
                    return -1;
                    }
            }
        for (int j = 0; j < c->size; j++)
            memcpy(r + (size_t)j * bytes + done, c->slots + (size_t)j * SHM_SLOT_BYTES, n);
        if (shm_barrier(p) != 0)
            return -1;
        done += n;
        } while (done < bytes);
    return 0;
    }

static void shm_destroy(void* p)
    {
    ShmCtx* c = (ShmCtx*)p;
    // Rank 0 removes the NAME first (the mapping outlives it), then everyone meets once more: a rank
    // that comes out of this barrier and re-initialises under the same name can no longer open the
    // old segment.
    if (c->rank == 0)
        shm_unlink(c->name.c_str());
    (void)shm_barrier(p);
    __atomic_store_n(&c->seg->pids[c->rank], 0, __ATOMIC_RELEASE); // leaving in good order
    munmap((void*)c->seg, c->map_bytes);
    delete c;
    }

static int comm_install(const pgsd_comm& c)
    {
    // the previous communicator is destroyed here unless open handles still hold it
    comm_slot() = std::make_shared<CommBox>(c);
    return PGSD_SUCCESS;
    }
    } // namespace pgsd_amd

using namespace pgsd_amd;

extern "C" const char* pgsd_last_error_string(void)
    try
    {
    return last_error();
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return nullptr;
    }

extern "C" int pgsd_comm_set_default(const struct pgsd_comm* comm)
    try
    {
    if (!comm || !comm->allgather || comm->size < 1 || comm->rank < 0 || comm->rank >= comm->size)
        return PGSD_ERROR_INVALID_ARGUMENT;
    return comm_install(*comm);
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

enum
    {
    SHM_RETRY = 1
    };

static int shm_attach(const char* name, int rank, int size, int attempt, pgsd_comm* out)
    {
    std::string nm = name[0] == '/' ? name : std::string("/") + name;
    size_t bytes = sizeof(ShmSegment) + (size_t)size * SHM_SLOT_BYTES;
    int fd = -1;
    if (rank == 0)
        {
        // a segment of that name left by a crashed run is removed; one whose creator is another LIVE
        // process belongs to a running job (two jobs of one user picked the same name) and is left alone
        int old = shm_open(nm.c_str(), O_RDWR, 0600);
        if (old >= 0)
            {
            struct stat st;
            int32_t creator = 0;
            if (fstat(old, &st) == 0 && (size_t)st.st_size >= sizeof(ShmSegment))
                {
                void* om = mmap(NULL, sizeof(ShmSegment), PROT_READ, MAP_SHARED, old, 0);
                if (om != MAP_FAILED)
                    {
                    const ShmSegment* os = (const ShmSegment*)om;
                    if (os->ready == SHM_MAGIC)
                        creator = os->creator_pid;
                    munmap(om, sizeof(ShmSegment));
                    }
                }
            close(old);
            if (creator > 0 && creator != (int32_t)getpid() && !process_gone(creator))
                {
                set_last_error("shm segment " + nm + " is in use by live process " + std::to_string(creator)
                               + ": choose another PGSD_SHM_NAME");
                return PGSD_ERROR_COMM;
                }
            shm_unlink(nm.c_str());
            }
        fd = shm_open(nm.c_str(), O_RDWR | O_CREAT | O_EXCL, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0)
            {
            set_last_error("shm_open/ftruncate failed for " + nm + ": " + strerror(errno));
            if (fd >= 0)
                close(fd);
            return PGSD_ERROR_COMM;
            }
        }
    else
        {
        // wait (up to ~60 s) for rank 0 to create and size the segment
        struct timespec ts = {0, 2000000};
        for (int tries = 0; tries < 30000; tries++)
            {
            fd = shm_open(nm.c_str(), O_RDWR, 0600);
            if (fd >= 0)
                {
                struct stat st;
                if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes)
                    break;
                close(fd);
                fd = -1;
                }
            nanosleep(&ts, NULL);
            }
        if (fd < 0)
            {
            set_last_error("timed out waiting for shm segment " + nm);
            return PGSD_ERROR_COMM;
            }
        }
    void* m = mmap(NULL, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED)
        {
        set_last_error(std::string("mmap of shm segment failed: ") + strerror(errno));
        return PGSD_ERROR_COMM;
        }
    ShmSegment* seg = (ShmSegment*)m;
    if (rank == 0)
        {
        seg->arrived = 0;
        seg->generation = 0;
        seg->broken = 0;
        memset((void*)seg->pids, 0, sizeof(seg->pids));
        seg->size = (uint32_t)size;
        seg->creator_pid = (int32_t)getpid();
        __sync_synchronize();
        seg->ready = SHM_MAGIC;
        }
    else
        {
        struct timespec ts = {0, 1000000};
        int tries = 0;
        while (seg->ready != SHM_MAGIC && tries++ < 60000)
            nanosleep(&ts, NULL);
        __sync_synchronize();
        if (seg->ready == SHM_MAGIC && seg->creator_pid > 0 && process_gone(seg->creator_pid))
            {
            // the segment of a crashed run under the same name, opened before this run's rank 0
            // replaced it: drop it and look again
            munmap(m, bytes);
            if (attempt < 6000) // about a minute, like the wait for the segment itself
                return SHM_RETRY;
            set_last_error("shm segment " + nm + " belongs to a process that no longer exists");
            return PGSD_ERROR_COMM;
            }
        if (seg->ready != SHM_MAGIC || seg->size != (uint32_t)size)
            {
            munmap(m, bytes);
            set_last_error("shm segment " + nm + " not initialised or size mismatch");
            return PGSD_ERROR_COMM;
            }
        }
    __atomic_store_n(&seg->pids[rank], (int32_t)getpid(), __ATOMIC_RELEASE);
    ShmCtx* c = new ShmCtx;
    c->seg = seg;
    c->slots = (char*)m + sizeof(ShmSegment);
    c->map_bytes = bytes;
    c->name = nm;
    c->rank = rank;
    c->size = size;
    pgsd_comm pc;
    memset(&pc, 0, sizeof(pc));
    pc.ctx = c;
    pc.rank = rank;
    pc.size = size;
    pc.allgather = shm_allgather;
    pc.barrier = shm_barrier;
    pc.destroy = shm_destroy;
    if (out)
        {
        *out = pc;
        return PGSD_SUCCESS;
        }
    return comm_install(pc);
    }

static int shm_open_comm(const char* name, int rank, int size, pgsd_comm* out)
    {
    if (!name || size < 1 || rank < 0 || rank >= size)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (size > SHM_MAX_RANKS)
        {
        set_last_error("the shm communicator takes at most 1024 ranks");
        return PGSD_ERROR_INVALID_ARGUMENT;
        }
    for (int attempt = 0;; attempt++)
        {
        int rc = shm_attach(name, rank, size, attempt, out);
        if (rc != SHM_RETRY)
            return rc;
        struct timespec nap = {0, 10000000};
        nanosleep(&nap, NULL);
        }
    }

extern "C" int pgsd_comm_create_shm(const char* name, int rank, int size, struct pgsd_comm* out)
    try
    {
    if (!out)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (size == 1 && name && rank == 0)
        {
        *out = make_self();
        return PGSD_SUCCESS;
        }
    return shm_open_comm(name, rank, size, out);
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" void pgsd_comm_release(struct pgsd_comm* comm)
    try
    {
    if (!comm)
        return;
    if (comm->destroy)
        comm->destroy(comm->ctx);
    memset(comm, 0, sizeof(*comm));
    }
catch (...)
    {
        pgsd_amd::abi_guard();
    }

// ranks of one node through a /dev/shm segment, installed as the process default (pgsd_comm_init_from_env)
static int install_shm(const char* name, int rank, int size)
    {
    if (!name || size < 1 || rank < 0 || rank >= size)
        return PGSD_ERROR_INVALID_ARGUMENT;
    if (size == 1)
        return comm_install(make_self());
    return shm_open_comm(name, rank, size, nullptr);
    }

extern "C" int pgsd_comm_init_from_env(void)
    try
    {
    const char* r = getenv("PGSD_RANK");
    const char* n = getenv("PGSD_NRANKS");
    const char* nm = getenv("PGSD_SHM_NAME");
    if (r && n)
        {
        // ranks started by one launcher share their parent: a name no other job of this user has
        std::string name = nm ? nm : std::string("pgsd_amd_ppid_") + std::to_string((long)getppid());
        return install_shm(name.c_str(), atoi(r), atoi(n));
        }
    r = getenv("RANK");
    n = getenv("WORLD_SIZE");
    if (r && n && atoi(n) > 1)
        {
        const char* port = getenv("MASTER_PORT");
        std::string name = std::string("pgsd_amd_") + (port ? port : "0");
        return install_shm(name.c_str(), atoi(r), atoi(n));
        }
    return comm_install(make_self());
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_comm_finalize(void)
    try
    {
    return comm_install(make_self());
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_comm_rank(void)
    try
    {
    return default_comm().rank;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_comm_size(void)
    try
    {
    return default_comm().size;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_comm_allgather(const void* send, void* recv, size_t bytes)
    try
    {
    pgsd_comm c = default_comm();
    return c.allgather(c.ctx, send, recv, bytes) == 0 ? PGSD_SUCCESS : PGSD_ERROR_COMM;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_comm_barrier(void)
    try
    {
    pgsd_comm c = default_comm();
    return comm_barrier(c);
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

extern "C" int pgsd_partition_rows(uint64_t n_local, uint64_t* row0, uint64_t* n_global, uint64_t* counts)
    try
    {
    pgsd_comm c = default_comm();
    std::vector<uint64_t> all((size_t)c.size);
    if (c.allgather(c.ctx, &n_local, all.data(), sizeof(uint64_t)) != 0)
        return PGSD_ERROR_COMM;
    uint64_t before = 0, total = 0;
    for (int j = 0; j < c.size; j++)
        {
        if (j < c.rank)
            before += all[(size_t)j];
        total += all[(size_t)j];
        if (counts)
            counts[j] = all[(size_t)j];
        }
    if (row0)
        *row0 = before;
    if (n_global)
        *n_global = total;
    return PGSD_SUCCESS;
    }
catch (...)
    {
        return pgsd_amd::abi_guard();
    }

// reference pgsd.h:735 / pgsd.c:152-172: broadcast of one index entry from rank 0
extern "C" void pgsd_bcast_index_entry(struct pgsd_index_entry* e)
    try
    {
    if (!e)
        return;
    pgsd_comm c = default_comm();
    std::vector<pgsd_index_entry> all((size_t)c.size);
    if (c.allgather(c.ctx, e, all.data(), sizeof(*e)) == 0)
        *e = all[0];
    }
catch (...)
    {
        pgsd_amd::abi_guard();
    }
