// pgsd_device.cpp -- the HBM -> file pipeline behind pgsd_write_chunk_device().
//
//   pack stream   fused pack kernel(s) write dense chunk bytes into a device staging arena
//                 (one kernel per submit; the arena holds a whole frame: 288 GB of HBM make
//                 a 0.3-1.6 GB snapshot copy cheap, and the simulation only has to wait for
//                 the ~100 us pack, not for the file write)
//   copy stream   waits for the pack event, then streams the staged bytes piece by piece
//                 into a ring of pinned host slabs with hipMemcpyAsync, so the copy of piece
//                 k+1 overlaps the file write of piece k and the pack of the next chunk.
//                 With the HIP runtime PyTorch ships (bench.py, pgsd.hoomd under torch) these copies run as
//                 SHADER BLITS, not on an SDMA engine (a HIP-only process on ROCm 7.2's own runtime shows no
//                 copy kernels: examples/dump_writer.hip): they appear in the kernel trace as
//                 __amd_rocclr_copyBuffer (240 dispatches of ~256 us for 12 frames of 10 M
//                 particles, profiles/r03_kernel_stats_10M.csv) and share the CUs with whatever
//                 the simulation runs.  They are PCIe-bound (50-55 GB/s) and need few CUs:
//                 a queue of HBM-bound kernels runs 0.7 % slower while snapshots drain, an
//                 fp32 GEMM queue 0.25 % (tools/overlap_probe.py, profiles/r04_overlap_probe.jsonl;
//                 bench.py reports it as snapshot_overlap_slowdown_pct) -- which is why no SDMA
//                 route was built
//   writer pool   each piece is pwrite()n at the file offset the reference's
//                 MPI_File_write_at would use (pgsd.c:2225-2229) as soon as its copy event
//                 has fired; the slab then returns to the ring
//
// pgsd_end_frame()/pgsd_flush() call drain(): the frame is in the file when it returns.
//
// Small frames take a shorter road (the "direct" path): when the chunks of one fused launch hold at most
// PGSD_DIRECT_MAX_KIB (default 2048 KiB) the kernel packs them straight into a pinned, device-mapped host
// arena -- the stores cross PCIe themselves -- and the bytes are pwrite()n by the thread that calls drain()
// after ONE stream wait: no HBM staging, no device->host copy, no dispatcher / writer hand-over.  For a snapshot of a
// few thousand particles those fixed costs were several times the frame itself (round 2: 176 us against 67 us
// for the same frame from host arrays).  HIP events come from a pool instead of being created per launch.
#include "pgsd_internal.hpp"
#include "pgsd_pack.hpp"

#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <sys/uio.h>
#include <unistd.h>
#include <cerrno>
#include <mutex>
#include <thread>

namespace pgsd_amd
    {
#define HIP_TRY(expr)                                                                      \
    do                                                                                     \
        {                                                                                  \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            {                                                                              \
            fail(std::string(#expr) + ": " + hipGetErrorString(e_));                       \
            (void)hipGetLastError(); /* reported; not left for a later launch check to find */ \
            return PGSD_ERROR_DEVICE;                                                      \
            }                                                                              \
        } while (0)

// Reader threads and their pinned ring are shared by every handle that reads on a device: ten
// trajectories open for reading cost 16 threads and 128 MiB of pinned memory, not 160 and 1.3 GiB.
// Created by the first reading pipeline, destroyed with the last.
struct ReadEngine
    {
    struct Slab
        {
        char* host = nullptr;
        hipEvent_t copied = nullptr;
        };
    int device = 0;
    int refs = 0;
    size_t piece = (size_t)4 << 20;
    WriterPool* pool = nullptr;
    std::vector<Slab> slabs;
    std::deque<uint32_t> free_slabs;
    std::mutex m;
    std::condition_variable cv;

    int get_slab()
        {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this] { return !free_slabs.empty(); });
        int si = (int)free_slabs.front();
        free_slabs.pop_front();
        return si;
        }

    void put_slab(int si)
        {
            {
            std::lock_guard<std::mutex> g(m);
            free_slabs.push_back((uint32_t)si);
            }
        cv.notify_one();
        }

    static std::mutex& registry_mutex()
        {
        static std::mutex mu;
        return mu;
        }

    static std::vector<ReadEngine*>& registry()
        {
        static std::vector<ReadEngine*> r;
        return r;
        }

    static ReadEngine* acquire(int device, const cpu_set_t* cpus, hipError_t* err)
        {
        std::lock_guard<std::mutex> g(registry_mutex());
        for (ReadEngine* e : registry())
            if (e->device == device)
                {
                e->refs++;
                return e;
                }
        ReadEngine* e = new ReadEngine;
        e->device = device;
        // Reads of the page cache take no exclusive lock and scale with threads; pieces smaller
        // than the write slabs keep all of them busy on one chunk and start the H2D copies
        // earlier (profiles/r01_read_sweep.log: 16 readers x 4 MiB pieces beat 8 x 16 MiB by 20-40 %).
        unsigned n = 16;
        if (const char* v = getenv("PGSD_READERS"))
            n = (unsigned)atoi(v) > 0 ? (unsigned)atoi(v) : n;
        if (const char* v = getenv("PGSD_READ_PIECE_MIB"))
            e->piece = (size_t)(atoi(v) > 0 ? atoi(v) : 4) << 20;
        e->slabs.resize((size_t)n * 2);
        hipError_t alloc_err = hipSuccess;
        std::thread allocator(
            [&]
            {
                if (cpus) // first touch on the GPU's node, like the write ring
                    (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), cpus);
                (void)hipSetDevice(device);
                for (auto& s : e->slabs)
                    {
                    hipError_t rc = hipHostMalloc((void**)&s.host, e->piece, hipHostMallocDefault);
                    if (rc == hipSuccess)
                        rc = hipEventCreateWithFlags(&s.copied, hipEventDisableTiming);
                    if (rc != hipSuccess && alloc_err == hipSuccess)
                        alloc_err = rc;
                    }
            });
        allocator.join();
        if (alloc_err != hipSuccess)
            {
            e->free_all();
            delete e;
            *err = alloc_err;
            return nullptr;
            }
        for (uint32_t i = 0; i < e->slabs.size(); i++)
            e->free_slabs.push_back(i);
        e->pool = writer_pool_create(n, cpus);
        e->refs = 1;
        registry().push_back(e);
        return e;
        }

    static void release(ReadEngine* e)
        {
            {
            std::lock_guard<std::mutex> g(registry_mutex());
            if (--e->refs > 0)
                return;
            auto& r = registry();
            for (size_t i = 0; i < r.size(); i++)
                if (r[i] == e)
                    r.erase(r.begin() + (long)i);
            }
        if (e->pool)
            writer_pool_destroy(e->pool); // joins the readers (no pipeline has work queued any more)
        (void)hipSetDevice(e->device);
        e->free_all();
        delete e;
        }

    void free_all()
        {
        for (auto& s : slabs)
            {
            if (s.copied)
                (void)hipEventDestroy(s.copied);
            if (s.host)
                (void)hipHostFree(s.host);
            }
        slabs.clear();
        }
    };

// What a pipeline needs from the HIP runtime is expensive to make and to give back: two streams (ms each),
// pinned slabs (2-3 ms per 16 MiB hipHostMalloc, more to free), the pinned arena of the direct path, HBM
// arenas, events.  A trajectory writer that opens one file per snapshot (or a benchmark that re-creates its
// file) paid 20-30 ms per open/close for them.  A closed pipeline of the default geometry therefore parks its
// resources here, at most two sets per process, and the next pipeline on the same device adopts them.
// Never freed at exit: no HIP calls from static destruction.
struct ParkedResources
    {
    int device = -1;
    uint64_t slab_bytes = 0;
    hipStream_t pack_stream = nullptr, copy_stream = nullptr;
    std::vector<std::pair<char*, hipEvent_t>> slabs; // pinned slab + its copy event
    char* dhost = nullptr;
    char* ddev = nullptr;
    size_t dcap = 0;
    std::vector<char*> arenas; // HBM staging arenas of the default size (256 MiB each)
    std::vector<hipEvent_t> ev_plain, ev_timing;
    };

static std::mutex& parked_mutex()
    {
    static std::mutex* m = new std::mutex;
    return *m;
    }

static std::vector<ParkedResources>& parked()
    {
    static std::vector<ParkedResources>* v = new std::vector<ParkedResources>;
    return *v;
    }

// Everything a parked set holds goes back to the runtime (pgsd_device_release_parked).
static void free_parked(ParkedResources& r)
    {
    (void)hipSetDevice(r.device);
    for (auto& sl : r.slabs)
        {
        if (sl.second)
            (void)hipEventDestroy(sl.second);
        if (sl.first)
            (void)hipHostFree(sl.first);
        }
    if (r.dhost)
        (void)hipHostFree(r.dhost);
    for (char* a : r.arenas)
        (void)hipFree(a);
    for (hipEvent_t e : r.ev_plain)
        (void)hipEventDestroy(e);
    for (hipEvent_t e : r.ev_timing)
        (void)hipEventDestroy(e);
    if (r.pack_stream)
        (void)hipStreamDestroy(r.pack_stream);
    if (r.copy_stream)
        (void)hipStreamDestroy(r.copy_stream);
    }

int release_parked_sets()
    {
    std::vector<ParkedResources> sets;
        {
        std::lock_guard<std::mutex> g(parked_mutex());
        sets.swap(parked());
        }
    int device = 0;
    const bool have = hipGetDevice(&device) == hipSuccess;
    for (auto& r : sets)
        free_parked(r);
    if (have && !sets.empty())
        (void)hipSetDevice(device);
    return (int)sets.size();
    }

class DevicePipeline
    {
    public:
    DevicePipeline(const pgsd_device_config& cfg, int fd, bool shared_file) : m_cfg(cfg), m_fd(fd), m_shared(shared_file) { }

    int init()
        {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
            {
            (void)hipGetLastError();
            fail("no HIP device visible: the device path has no CPU fallback");
            return PGSD_ERROR_NO_DEVICE;
            }
        if (m_cfg.device < 0)
            HIP_TRY(hipGetDevice(&m_cfg.device));
        HIP_TRY(hipSetDevice(m_cfg.device));
        if (m_cfg.slab_bytes == 0)
            m_cfg.slab_bytes = (uint64_t)16 << 20;
        if (m_cfg.n_slabs == 0)
            m_cfg.n_slabs = 16;
        // Buffered writes into ONE file serialise on its inode lock: a second writer thread
        // halves the rate on tmpfs (profiles/r01_io_probe*.log), so one writer is the default.
        if (m_cfg.n_writers == 0)
            m_cfg.n_writers = 1;
        m_direct_max = (size_t)2048 << 10;
        if (const char* v = getenv("PGSD_DIRECT_MAX_KIB"))
            m_direct_max = (size_t)(atoll(v) > 0 ? atoll(v) : 0) << 10;
        if (const char* v = getenv("PGSD_DIRECT_COALESCE")) // 0: one pwrite per chunk (for A/B measurements)
            m_coalesce = atoi(v) != 0;
        if (const char* v = getenv("PGSD_STAGING_CAP_MIB")) // how much HBM asynchronously sealed frames may hold before a call waits
            m_soft_cap = (size_t)(atoll(v) > 0 ? atoll(v) : 6144) << 20;
        ParkedResources adopted;
        bool have_parked = false;
            {
            std::lock_guard<std::mutex> g(parked_mutex());
            auto& v = parked();
            for (size_t i = 0; i < v.size(); i++)
                if (v[i].device == m_cfg.device && v[i].slab_bytes == m_cfg.slab_bytes)
                    {
                    adopted = std::move(v[i]);
                    v.erase(v.begin() + (long)i);
                    have_parked = true;
                    break;
                    }
            }
        if (have_parked)
            {
            m_pack_stream = adopted.pack_stream;
            m_copy_stream = adopted.copy_stream;
            m_dhost = adopted.dhost, m_ddev = adopted.ddev, m_dcap = adopted.dcap;
            for (char* a : adopted.arenas)
                m_arenas.push_back({a, (size_t)256 << 20, 0});
            m_pool_plain = std::move(adopted.ev_plain);
            m_pool_timing = std::move(adopted.ev_timing);
            }
        else
            {
            HIP_TRY(hipStreamCreateWithFlags(&m_pack_stream, hipStreamNonBlocking));
            HIP_TRY(hipStreamCreateWithFlags(&m_copy_stream, hipStreamNonBlocking));
            }
        // The pinned slabs, the thread that copies them into the page cache and the pages it
        // allocates there all belong on the NUMA node the GPU hangs off (two-socket hosts: the
        // other node costs ~10 % of the write rate, profiles/r01_numa.log).
        char bdf[32] = {0};
        m_numa = hipDeviceGetPCIBusId(bdf, (int)sizeof(bdf), m_cfg.device) == hipSuccess
                 && numa_cpus_of_pci_device(bdf, &m_numa_cpus);
        m_slabs.resize(m_cfg.n_slabs);
        // parked slabs first (already pinned on this device's NUMA node); what the ring cannot take is freed
        for (auto& ps : adopted.slabs)
            {
            if (m_slabs_ready < m_cfg.n_slabs)
                {
                m_slabs[m_slabs_ready].host = ps.first;
                m_slabs[m_slabs_ready].copied = ps.second;
                m_slabs_ready++;
                }
            else
                {
                (void)hipEventDestroy(ps.second);
                (void)hipHostFree(ps.first);
                }
            }
        // prealloc_mib: the whole ring and that much staging now, nothing in the middle of the run
        const uint32_t first_slabs = m_cfg.prealloc_mib ? m_cfg.n_slabs : std::min<uint32_t>(m_cfg.n_slabs, 2);
        if (m_slabs_ready < first_slabs)
            {
            hipError_t alloc_err = hipSuccess;
            std::thread allocator(
                [&]
                {
                    // first touch decides the node: allocate from a thread that runs there
                    if (m_numa)
                        (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), &m_numa_cpus);
                    (void)hipSetDevice(m_cfg.device);
                    // two slabs now; the dispatcher (pinned to the same node) adds the rest of the ring
                    // when a frame actually needs them, so a small file never pins 256 MiB
                    while (m_slabs_ready < first_slabs)
                        {
                        hipError_t e = alloc_slab(m_slabs[m_slabs_ready]);
                        if (e != hipSuccess)
                            {
                            alloc_err = e;
                            break;
                            }
                        m_slabs_ready++;
                        }
                });
            allocator.join();
            HIP_TRY(alloc_err);
            }
        for (uint32_t i = 0; i < m_slabs_ready; i++)
            m_free_slabs.push_back(i);
        size_t staged_cap = 0;
        for (auto& a : m_arenas)
            staged_cap += a.cap;
        if (m_cfg.prealloc_mib)
            {
            warm_pack_kernels();
            warm_unpack_kernels();
            warm_select_kernels();
            int brc = compare_buffers();
            if (brc != PGSD_SUCCESS)
                return brc;
            }
        while (staged_cap < ((size_t)m_cfg.prealloc_mib << 20))
            {
            Arena a;
            a.base = nullptr;
            a.cap = (size_t)256 << 20;
            a.used = 0;
            HIP_TRY(hipMalloc((void**)&a.base, a.cap));
            m_arenas.push_back(a);
            staged_cap += a.cap;
            }
        const cpu_set_t* pin = m_numa ? &m_numa_cpus : nullptr;
        m_pool = writer_pool_create(m_cfg.n_writers, pin);
        m_dispatcher = std::thread([this] { dispatch_loop(); });
        if (pin)
            (void)pthread_setaffinity_np(m_dispatcher.native_handle(), sizeof(cpu_set_t), pin);
        m_ok = true;
        return PGSD_SUCCESS;
        }

    ~DevicePipeline()
        {
        if (m_dispatcher.joinable())
            {
                {
                std::lock_guard<std::mutex> g(m_mutex);
                m_stop = true;
                }
            m_cv_jobs.notify_all();
            m_cv_slabs.notify_all();
            m_dispatcher.join();
            }
        if (m_reader)
            {
                {
                // the readers are shared: wait for this handle's pieces instead of joining them
                std::unique_lock<std::mutex> lk(m_mutex);
                m_cv_done.wait(lk, [this] { return m_reads_outstanding == 0; });
                }
            ReadEngine::release(m_reader);
            }
        if (m_pool)
            writer_pool_destroy(m_pool); // joins the writers
        (void)hipSetDevice(m_cfg.device);
        release_events();
        if (m_cmp_host)
            (void)hipHostFree(m_cmp_host);
        if (m_cmp_dev)
            (void)hipFree(m_cmp_dev);
        if (park())
            return;
        for (hipEvent_t e : m_pool_plain)
            (void)hipEventDestroy(e);
        for (hipEvent_t e : m_pool_timing)
            (void)hipEventDestroy(e);
        if (m_dhost)
            (void)hipHostFree(m_dhost);
        for (auto& s : m_slabs)
            {
            if (s.copied)
                (void)hipEventDestroy(s.copied);
            if (s.host)
                (void)hipHostFree(s.host);
            }
        for (auto& a : m_arenas)
            (void)hipFree(a.base);
        if (m_pack_stream)
            (void)hipStreamDestroy(m_pack_stream);
        if (m_copy_stream)
            (void)hipStreamDestroy(m_copy_stream);
        }

    // Hand streams, pinned memory, one HBM arena and the idle events to the next pipeline on this device
    // (see ParkedResources).  Only a healthy pipeline of the default slab size, only while fewer than two sets
    // are parked, and only what a default pipeline would hold: at most four slabs, four arenas of the default size.
    bool park()
        {
        const uint64_t default_slab = (uint64_t)16 << 20;
        if (!m_ok || !m_error.empty() || m_cfg.slab_bytes != default_slab || !m_pack_stream || !m_copy_stream)
            return false;
        if (getenv("PGSD_NO_PARKING"))
            return false;
        if (hipStreamSynchronize(m_pack_stream) != hipSuccess || hipStreamSynchronize(m_copy_stream) != hipSuccess)
            {
            (void)hipGetLastError();
            return false;
            }
        ParkedResources r;
        r.device = m_cfg.device;
        r.slab_bytes = m_cfg.slab_bytes;
        // room is checked and the set pushed under ONE lock (two pipelines closing at once could both see
        // "one set parked" and leave three)
        std::lock_guard<std::mutex> g(parked_mutex());
        if (parked().size() >= 2)
            return false;
        r.pack_stream = m_pack_stream;
        r.copy_stream = m_copy_stream;
        for (auto& sl : m_slabs)
            {
            if (!sl.host)
                continue;
            if (r.slabs.size() < 4)
                r.slabs.push_back({sl.host, sl.copied});
            else
                {
                (void)hipEventDestroy(sl.copied);
                (void)hipHostFree(sl.host);
                }
            }
        r.dhost = m_dhost, r.ddev = m_ddev, r.dcap = m_dcap;
        for (auto& a : m_arenas)
            {
            // up to 1 GiB of staging stays with the set (a run of asynchronously sealed frames grows into it;
            // giving 256 MiB blocks back and asking for them again cost 3-5 ms apiece)
            if (r.arenas.size() < 4 && a.cap == ((size_t)256 << 20))
                r.arenas.push_back(a.base);
            else
                (void)hipFree(a.base);
            }
        r.ev_plain = std::move(m_pool_plain);
        r.ev_timing = std::move(m_pool_timing);
        parked().push_back(std::move(r));
        return true;
        }

    // order the pack stream after whatever the caller enqueued on its source stream -- an event there and a wait
    // here, unless that stream has nothing in flight (one query instead of two calls: small frames)
    int order_after_source()
        {
        if (hipStreamQuery(m_source_stream) != hipSuccess)
            {
            (void)hipGetLastError(); // hipErrorNotReady is the answer, not an error
            hipEvent_t ready = get_event(false);
            if (!ready)
                return PGSD_ERROR_DEVICE;
            HIP_TRY(hipEventRecord(ready, m_source_stream));
            HIP_TRY(hipStreamWaitEvent(m_pack_stream, ready, 0));
            }
        return PGSD_SUCCESS;
        }

    // Pack now, place later.  stage() packs the chunks into the staging arena with one fused launch and
    // returns a ticket; commit() tells where chunk `index` of the ticket goes -- a file offset (asynchronous
    // copy + pwrite), a host buffer (synchronous copy: small replicated chunks headed for the write
    // buffer) or nowhere (both absent) -- and the ticket is closed by its last commit.  Between the two
    // calls lies the frame's size exchange between the ranks: the pack kernel does not need file
    // offsets, so it runs while the exchange is on its way.
    int stage(std::vector<DeviceChunk>& chunks, uint64_t N, int* ticket)
        {
        if (!m_ok)
            return PGSD_ERROR_NO_DEVICE;
        if (failed())
            return PGSD_ERROR_DEVICE;
        TraceRange tr("pgsd:stage+pack_launch rows=%llu chunks=%llu", N, chunks.size());
        HIP_TRY(hipSetDevice(m_cfg.device));
        int rrc = recycle_staging();
        if (rrc != PGSD_SUCCESS)
            return rrc;

        std::vector<pgsd_pack_job> jobs;
        uint64_t bytes_in = 0, bytes_out = 0;
        // a small launch packs straight into pinned host memory (see the head of this file)
        size_t padded = 0;
        for (auto& c : chunks)
            padded += ((size_t)(c.N * c.job.M * sizeof_type(c.job.dst_type)) + 255) & ~(size_t)255;
        const bool direct = padded > 0 && padded <= m_direct_max && direct_reserve(padded);
        for (auto& c : chunks)
            {
            size_t bytes = (size_t)(c.N * c.job.M * sizeof_type(c.job.dst_type));
            void* stage = nullptr;
            if (direct)
                {
                stage = m_ddev + m_dused;
                m_dused += (bytes + 255) & ~(size_t)255;
                }
            else
                {
                int rc = arena_alloc(bytes, &stage);
                if (rc != PGSD_SUCCESS)
                    return rc;
                }
            c.job.dst = stage;
            jobs.push_back(c.job);
            bytes_in += pack_algorithmic_bytes_in(c.job, c.N);
            bytes_out += bytes;
            }

        int orc = order_after_source();
        if (orc != PGSD_SUCCESS)
            return orc;

        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        if (m_cfg.profile)
            {
            ev0 = get_event(true, false);
            ev1 = get_event(true, false);
            if (!ev0 || !ev1)
                return PGSD_ERROR_DEVICE;
            }
        std::string err;
        // profiled: the events are stamped by the kernel dispatch itself (begin / end)
        int rc = launch_pack((uint32_t)jobs.size(), jobs.data(), N, m_pack_stream, &err, ev0, ev1);
        if (rc != PGSD_SUCCESS)
            {
            fail(err);
            return rc;
            }
        if (m_cfg.profile)
            {
            std::lock_guard<std::mutex> g(m_mutex);
            m_pack_events.push_back({ev0, ev1});
            }
        // the event behind the launch: what the copies of a staged chunk wait for.  A direct launch is waited for
        // with a stream synchronize by whoever writes it (drain, or the writer thread after an asynchronous
        // seal), so it needs none
        hipEvent_t packed = nullptr;
        if (!direct)
            {
            packed = get_event(false);
            if (!packed)
                return PGSD_ERROR_DEVICE;
            HIP_TRY(hipEventRecord(packed, m_pack_stream));
            }
        std::lock_guard<std::mutex> g(m_mutex);
        m_stats.pack_launches++;
        m_stats.pack_rows += N;
        m_stats.pack_bytes_in += bytes_in;
        m_stats.pack_bytes_out += bytes_out;
        Staged st;
        st.chunks = chunks;
        st.packed = packed;
        st.open = chunks.size();
        st.ramped = false;
        st.direct = direct;
        *ticket = m_next_ticket++;
        m_staged[*ticket] = st;
        return PGSD_SUCCESS;
        }

    int commit(int ticket, size_t index, long long file_offset, void* host_dst)
        {
        DeviceChunk c;
        hipEvent_t packed;
        bool ramp, direct;
            {
            std::lock_guard<std::mutex> g(m_mutex);
            auto it = m_staged.find(ticket);
            if (it == m_staged.end() || index >= it->second.chunks.size())
                return PGSD_ERROR_INVALID_ARGUMENT;
            c = it->second.chunks[index];
            packed = it->second.packed;
            direct = it->second.direct;
            ramp = !it->second.ramped && !host_dst && file_offset >= 0;
            if (ramp)
                it->second.ramped = true; // the first chunk of a launch starts with small pieces
            if (--it->second.open == 0)
                m_staged.erase(it);
            }
        const size_t bytes = (size_t)(c.N * c.job.M * sizeof_type(c.job.dst_type));
        if (bytes == 0 || (!host_dst && file_offset < 0))
            return PGSD_SUCCESS;
        if (failed())
            return PGSD_ERROR_DEVICE;
        if (direct)
            {
            // the kernel is packing (or has packed) these bytes into pinned host memory
            const char* host = m_dhost + ((const char*)c.job.dst - m_ddev);
            if (host_dst)
                {
                // small replicated chunk headed for the write buffer: wait for the launch, plain memcpy
                HIP_TRY(hipStreamSynchronize(m_pack_stream));
                memcpy(host_dst, host, bytes);
                return PGSD_SUCCESS;
                }
            std::lock_guard<std::mutex> g(m_mutex);
            m_direct.push_back({host, bytes, file_offset});
            return PGSD_SUCCESS;
            }
        if (host_dst)
            {
            // small replicated chunk headed for the write buffer: synchronous
            HIP_TRY(hipSetDevice(m_cfg.device));
            HIP_TRY(hipStreamSynchronize(m_pack_stream));
            HIP_TRY(hipMemcpy(host_dst, c.job.dst, bytes, hipMemcpyDeviceToHost));
            return PGSD_SUCCESS;
            }
        CopyJob j;
        j.dsrc = (const char*)c.job.dst;
        j.bytes = bytes;
        j.file_offset = file_offset;
        j.packed = packed;
        j.ramp = ramp;
        size_t pieces = 0;
        for (size_t off = 0; off < bytes; off += piece_len(off, bytes, j.ramp))
            pieces++;
        std::unique_lock<std::mutex> lk(m_mutex);
        m_outstanding += pieces;
        m_jobs.push_back(j);
        lk.unlock();
        m_cv_jobs.notify_one();
        return PGSD_SUCCESS;
        }

    // stage + commit at once: the caller already knows where the chunks go
    int submit(std::vector<DeviceChunk>& chunks, uint64_t N)
        {
        int ticket = -1;
        int rc = stage(chunks, N, &ticket);
        for (size_t i = 0; i < chunks.size() && rc == PGSD_SUCCESS; i++)
            rc = commit(ticket, i, chunks[i].host_dst ? -1 : chunks[i].file_offset, chunks[i].host_dst);
        return rc;
        }

    // Packed bytes of staged (not yet committed) chunks against reference bytes in device memory: one kernel behind
    // the pack on the pack stream, one stream wait, the answers in pinned words the kernel wrote across PCIe.
    int device() const
        {
        return m_cfg.device;
        }

    int compare(int ticket, size_t first, size_t count, const void* const* ref, const uint64_t* ref_bytes, uint8_t* equal)
        {
        if (!m_ok)
            return PGSD_ERROR_NO_DEVICE;
        if (failed())
            return PGSD_ERROR_DEVICE;
        HIP_TRY(hipSetDevice(m_cfg.device));
        std::vector<CompareJob> jobs;
        std::vector<size_t> who;
            {
            std::lock_guard<std::mutex> g(m_mutex);
            auto it = m_staged.find(ticket);
            if (it == m_staged.end() || first + count > it->second.chunks.size())
                return PGSD_ERROR_INVALID_ARGUMENT;
            for (size_t i = 0; i < count; i++)
                {
                const DeviceChunk& c = it->second.chunks[first + i];
                const uint64_t bytes = c.N * c.job.M * sizeof_type(c.job.dst_type);
                equal[i] = ref[i] != nullptr && bytes == 0 ? 1 : 0; // no rows here: nothing that could differ
                if (ref[i] != nullptr && bytes > 0)
                    {
                    CompareJob j;
                    memset(&j, 0, sizeof(j));
                    j.a = c.job.dst;
                    j.b = ref[i];
                    j.bytes = bytes;
                    j.mode = c.job.dst_type == PGSD_TYPE_FLOAT ? CMP_F32 : c.job.dst_type == PGSD_TYPE_DOUBLE ? CMP_F64 : CMP_BYTES;
                    if (ref_bytes && ref_bytes[i] < bytes)
                        {
                        // a reference shorter than the chunk repeats: whole 16-byte vectors, whole ROWS (a period that
                        // cuts a row would be compared out of phase from the second repetition on), and long enough for
                        // the kernel's one-step wrap (256 vectors)
                        const uint64_t p = ref_bytes[i];
                        if (p < 4096 || p % 16 != 0 || p % ((uint64_t)c.job.M * sizeof_type(c.job.dst_type)) != 0
                            || ((uintptr_t)ref[i] & 15) != 0)
                            {
                            set_last_error("a repeating reference must be 16-byte aligned, at least 4096 bytes and a multiple "
                                           "of 16 bytes and of whole rows");
                            return PGSD_ERROR_INVALID_ARGUMENT;
                            }
                        j.period = p;
                        }
                    jobs.push_back(j);
                    who.push_back(i);
                    }
                }
            }
        if (jobs.empty())
            return PGSD_SUCCESS;
        TraceRange tr("pgsd:compare chunks=%llu", (unsigned long long)jobs.size(), 0ull);
        // the reference bytes may have been produced on the caller's stream a moment ago
        int orc = order_after_source();
        if (orc != PGSD_SUCCESS)
            return orc;
        int brc = compare_buffers();
        if (brc != PGSD_SUCCESS)
            return brc;
        for (size_t at = 0; at < jobs.size(); at += CMP_MAX_JOBS)
            {
            const uint32_t n = (uint32_t)std::min<size_t>(CMP_MAX_JOBS, jobs.size() - at);
            // one sequence for the process: a later pipeline may be handed the memory an earlier one marked
            static std::atomic<uint32_t> next_gen {0};
            do
                m_cmp_gen = ++next_gen;
            while (m_cmp_gen == 0);
            std::string err;
            int rc = launch_compare(n, jobs.data() + at, m_cmp_gen, m_cmp_dev, m_cmp_host_dev, m_pack_stream, &err);
            if (rc != PGSD_SUCCESS)
                {
                fail(err);
                return rc;
                }
            hipError_t e = hipStreamSynchronize(m_pack_stream);
            if (e != hipSuccess)
                {
                fail(std::string("hipStreamSynchronize(compare): ") + hipGetErrorString(e));
                return PGSD_ERROR_DEVICE;
                }
            for (uint32_t k = 0; k < n; k++)
                equal[who[at + k]] = __atomic_load_n(&m_cmp_host[k], __ATOMIC_RELAXED) != m_cmp_gen ? 1 : 0;
            }
        return PGSD_SUCCESS;
        }

    // Packed bytes of staged chunks into caller-owned device memory (frame 0's rows kept for later comparisons):
    // copies on the pack stream behind the pack, asynchronous.
    int copy_staged(int ticket, size_t first, size_t count, void* const* dst)
        {
        if (!m_ok)
            return PGSD_ERROR_NO_DEVICE;
        if (failed())
            return PGSD_ERROR_DEVICE;
        HIP_TRY(hipSetDevice(m_cfg.device));
        std::vector<std::pair<const void*, size_t>> src(count);
            {
            std::lock_guard<std::mutex> g(m_mutex);
            auto it = m_staged.find(ticket);
            if (it == m_staged.end() || first + count > it->second.chunks.size())
                return PGSD_ERROR_INVALID_ARGUMENT;
            for (size_t i = 0; i < count; i++)
                {
                const DeviceChunk& c = it->second.chunks[first + i];
                src[i] = {c.job.dst, (size_t)(c.N * c.job.M * sizeof_type(c.job.dst_type))};
                }
            }
        // the destinations are the caller's memory (torch's caching allocator hands out blocks that work queued on
        // the caller's stream may still be using): the copies are ordered behind that stream, as stage() orders the
        // pack behind it and compare() the comparison
        int orc = order_after_source();
        if (orc != PGSD_SUCCESS)
            return orc;
        for (size_t i = 0; i < count; i++)
            if (dst[i] != nullptr && src[i].second > 0)
                HIP_TRY(hipMemcpyAsync(dst[i], src[i].first, src[i].second, hipMemcpyDefault, m_pack_stream));
        return PGSD_SUCCESS;
        }

    // Asynchronous seal (pgsd_end_frame_async): nobody will call drain() for this frame, so the direct
    // chunks committed so far are handed to the writer thread, which waits for their launch and writes them.
    void kick_direct()
        {
        std::vector<DirectWrite> list;
            {
            std::lock_guard<std::mutex> g(m_mutex);
            if (m_direct.empty())
                return;
            list.swap(m_direct);
            m_outstanding += 1;
            }
        writer_pool_submit(m_pool,
                           [this, list]
                           {
                               (void)hipSetDevice(m_cfg.device);
                               // everything launched on the pack stream so far, these chunks' kernel included
                               hipError_t e = hipStreamSynchronize(m_pack_stream);
                               if (e != hipSuccess)
                                   fail(std::string("hipStreamSynchronize(pack): ") + hipGetErrorString(e));
                               else
                                   write_direct(list);
                               piece_done(0, 0);
                           });
        }

    // host bytes (an asynchronously sealed frame's metadata) written by the writer thread, behind what is queued
    void write_host(const void* data, size_t bytes, long long file_offset)
        {
        if (bytes == 0)
            return;
        auto copy = std::make_shared<std::vector<char>>((const char*)data, (const char*)data + bytes);
            {
            std::lock_guard<std::mutex> g(m_mutex);
            m_outstanding += 1;
            }
        writer_pool_submit(m_pool,
                           [this, copy, file_offset]
                           {
                               int w = pwrite_locked(m_fd, copy->data(), copy->size(), file_offset, m_shared);
                               if (w != 0)
                                   fail(std::string("pwrite: ") + strerror(-w), true, -w);
                               piece_done(0, 0);
                           });
        }

    bool single_writer() const
        {
        return m_cfg.n_writers == 1;
        }

    bool staged_open()
        {
        std::lock_guard<std::mutex> g(m_mutex);
        return !m_staged.empty();
        }

    // ---- read side: file -> pinned slab (pread) -> HBM staging (H2D) -> unpack kernel ----
    int read_submit(long long file_offset, size_t bytes, pgsd_unpack_job job, uint64_t N)
        {
        if (!m_ok)
            return PGSD_ERROR_NO_DEVICE;
        if (failed())
            return PGSD_ERROR_DEVICE;
        HIP_TRY(hipSetDevice(m_cfg.device));
        const size_t padded = (bytes + 255) & ~(size_t)255;
        if (bytes > 0 && bytes <= m_direct_max && direct_reserve(padded))
            {
            // Small read, the short road (twin of the direct write path): THIS thread preads the rows straight
            // into the pinned, device-mapped arena and the unpack kernel fetches them from there over PCIe --
            // no reader-thread hand-over, no host->device copy, no staging in HBM.  For a frame of a few
            // thousand particles those fixed costs were ten times the read itself.
            char* host = m_dhost + m_dused;
            job.src = m_ddev + m_dused;
            m_dused += padded;
                {
                TraceRange tr("pgsd:pread_direct file_off=%llu bytes=%llu", (unsigned long long)file_offset, bytes);
                size_t got = 0;
                while (got < bytes)
                    {
                    ssize_t r = io_pread(m_fd, host + got, bytes - got, file_offset + (long long)got);
                    if (r < 0 && errno == EINTR)
                        continue;
                    if (r <= 0)
                        break;
                    got += (size_t)r;
                    }
                if (got != bytes)
                    {
                    fail("pread returned fewer bytes than the chunk holds", true);
                    return PGSD_SUCCESS; // reported by pgsd_device_wait_read, like the threaded path
                    }
                }
            auto direct_req = std::make_shared<ReadReq>();
            direct_req->job = job;
            direct_req->N = N;
            direct_req->pieces_left = 0;
            direct_req->all_copied = nullptr; // nothing to wait for: the bytes are there
            std::lock_guard<std::mutex> g(m_copy_mutex);
            m_unpack_pending.push_back(direct_req);
            return PGSD_SUCCESS;
            }
        void* stage = nullptr;
        int rc = arena_alloc(bytes, &stage);
        if (rc != PGSD_SUCCESS)
            return rc;
        job.src = stage;
        if (!m_reader)
            {
            hipError_t rerr = hipSuccess;
            m_reader = ReadEngine::acquire(m_cfg.device, m_numa ? &m_numa_cpus : nullptr, &rerr);
            if (!m_reader)
                HIP_TRY(rerr);
            }
        const size_t piece = m_reader->piece;
        auto req = std::make_shared<ReadReq>();
        req->job = job;
        req->N = N;
        req->pieces_left = (bytes + piece - 1) / piece;
        req->all_copied = get_event(false);
        if (!req->all_copied)
            return PGSD_ERROR_DEVICE;
        std::unique_lock<std::mutex> lk(m_mutex);
        m_reads_outstanding += req->pieces_left; // counted per piece: see read_piece()
        lk.unlock();
        for (size_t off = 0; off < bytes; off += piece)
            {
            size_t n = std::min(piece, bytes - off);
            char* dst = (char*)stage + off;
            long long foff = file_offset + (long long)off;
            writer_pool_submit(m_reader->pool, [this, req, dst, n, foff] { read_piece(req, dst, n, foff); });
            }
        return PGSD_SUCCESS;
        }

    int wait_read()
        {
        if (!m_ok)
            return PGSD_SUCCESS;
        std::unique_lock<std::mutex> lk(m_mutex);
        m_cv_done.wait(lk, [this] { return m_reads_outstanding == 0; });
        lk.unlock();
        (void)hipSetDevice(m_cfg.device);
        launch_pending_unpacks();
        hipError_t e = m_copy_used.exchange(false) ? hipStreamSynchronize(m_copy_stream) : hipSuccess;
        if (e == hipSuccess)
            e = hipStreamSynchronize(m_pack_stream);
        if (e != hipSuccess)
            fail(std::string("stream synchronize: ") + hipGetErrorString(e));
        bool writes_idle;
            {
            std::lock_guard<std::mutex> g(m_mutex);
            writes_idle = m_outstanding == 0;
            }
        if (writes_idle && !staged_open() && !direct_pending())
            {
            release_events();
            for (auto& a : m_arenas)
                a.used = 0;
            m_dused = 0;
            }
        if (failed())
            {
            if (m_io_error && m_io_errno)
                errno = m_io_errno;
            return m_io_error ? PGSD_ERROR_IO : PGSD_ERROR_DEVICE;
            }
        return PGSD_SUCCESS;
        }

    void set_source_stream(void* stream)
        {
        m_source_stream = (hipStream_t)stream;
        }

    int wait_packed()
        {
        if (!m_ok)
            return PGSD_SUCCESS;
        HIP_TRY(hipSetDevice(m_cfg.device));
        HIP_TRY(hipStreamSynchronize(m_pack_stream));
        return failed() ? PGSD_ERROR_DEVICE : PGSD_SUCCESS;
        }

    // every submitted byte is in the file (or an error is reported); staging is recycled
    int drain()
        {
        if (!m_ok)
            {
            return PGSD_SUCCESS;
            }
        TraceRange tr("pgsd:drain");
            {
            std::unique_lock<std::mutex> lk(m_mutex);
            m_cv_done.wait(lk, [this] { return m_outstanding == 0 || !m_error.empty(); });
            }
        (void)hipSetDevice(m_cfg.device);
        hipError_t e = hipStreamSynchronize(m_pack_stream);
        if (e == hipSuccess && m_copy_used.exchange(false))
            e = hipStreamSynchronize(m_copy_stream); // (a frame of direct chunks never touched it)
        if (e != hipSuccess)
            fail(std::string("stream synchronize: ") + hipGetErrorString(e));
        else
            {
            // direct chunks: their launch has finished (pack stream synchronised above), this thread writes them
            std::vector<DirectWrite> list;
                {
                std::lock_guard<std::mutex> g(m_mutex);
                list.swap(m_direct);
                }
            if (!list.empty())
                write_direct(list);
            }
        collect_timings();
        if (!staged_open())
            {
            release_events();
            for (auto& a : m_arenas)
                a.used = 0;
            m_dused = 0;
            }
        if (failed())
            {
            // let the writers finish what they hold before the caller tears anything down
            std::unique_lock<std::mutex> lk(m_mutex);
            m_cv_done.wait_for(lk, std::chrono::seconds(30), [this] { return m_outstanding == 0; });
            {
            if (m_io_error && m_io_errno)
                errno = m_io_errno;
            return m_io_error ? PGSD_ERROR_IO : PGSD_ERROR_DEVICE;
            }
            }
        return PGSD_SUCCESS;
        }

    void stats(pgsd_device_stats* out, int reset)
        {
        std::lock_guard<std::mutex> g(m_mutex);
        *out = m_stats;
        if (reset)
            memset(&m_stats, 0, sizeof(m_stats));
        }

    std::string error()
        {
        std::lock_guard<std::mutex> g(m_mutex);
        return m_error;
        }

    private:
    struct Slab
        {
        char* host = nullptr;
        hipEvent_t copied = nullptr;
        };
    struct Arena
        {
        char* base;
        size_t cap, used;
        };
    struct ReadReq
        {
        pgsd_unpack_job job;
        uint64_t N;
        size_t pieces_left;
        hipEvent_t all_copied;
        };
    struct Staged
        {
        std::vector<DeviceChunk> chunks;
        hipEvent_t packed;
        size_t open;
        bool ramped;
        bool direct; // packed into the pinned host arena, written by drain() / kick_direct()
        };
    struct CopyJob
        {
        const char* dsrc;
        size_t bytes;
        long long file_offset;
        hipEvent_t packed;
        bool ramp;
        };

    // Bytes of the piece that starts at `off`.  The writer cannot start before the first piece has
    // crossed PCIe, so the first chunk of a frame begins with a 1 MiB and a 4 MiB piece (0.02 ms
    // instead of 0.3 ms until the first pwrite: 5 % of a 1 M-particle frame) before full slabs follow.
    size_t piece_len(size_t off, size_t bytes, bool ramp) const
        {
        size_t cap = (size_t)m_cfg.slab_bytes;
        if (ramp && off == 0)
            cap = std::min(cap, (size_t)1 << 20);
        else if (ramp && off < ((size_t)5 << 20))
            cap = std::min(cap, (size_t)4 << 20);
        return std::min(cap, bytes - off);
        }

    hipError_t alloc_slab(Slab& s)
        {
        hipError_t e = hipHostMalloc((void**)&s.host, m_cfg.slab_bytes, hipHostMallocDefault);
        if (e == hipSuccess)
            e = hipEventCreateWithFlags(&s.copied, hipEventDisableTiming);
        return e;
        }

    void fail(const std::string& msg, bool io = false, int io_errno = 0)
        {
        std::lock_guard<std::mutex> g(m_mutex);
        if (m_error.empty())
            {
            m_error = msg;
            m_io_error = io;
            m_io_errno = io_errno;
            }
        m_cv_done.notify_all();
        m_cv_slabs.notify_all();
        }

    bool failed()
        {
        std::lock_guard<std::mutex> g(m_mutex);
        return !m_error.empty();
        }

    // Staging is recycled when nothing is in flight (cheap, also serves asynchronously sealed
    // frames that have drained on their own); if frames are produced faster than the file takes
    // them, block once staging exceeds the soft cap instead of growing without bound.
    int recycle_staging()
        {
        size_t used = m_dused;
        for (auto& a : m_arenas)
            used += a.used;
        if (used == 0)
            return PGSD_SUCCESS;
        bool idle;
            {
            std::lock_guard<std::mutex> g(m_mutex);
            if (!m_staged.empty() || !m_direct.empty())
                return PGSD_SUCCESS; // packed chunks still wait for their place in the file
            idle = m_outstanding == 0 && m_reads_outstanding == 0 && m_jobs.empty();
            }
            {
            // ... and so may chunks that were READ: between pgsd_read_chunk_device(wait = false) and
            // pgsd_device_wait_read their bytes sit in the staging (arena or pinned direct arena) until the unpack
            // launch has taken them -- a direct read counts no outstanding piece
            std::lock_guard<std::mutex> g(m_copy_mutex);
            if (!m_unpack_pending.empty())
                return PGSD_SUCCESS;
            }
        const size_t soft_cap = m_soft_cap;
        if (!idle && used < soft_cap)
            return PGSD_SUCCESS;
        if (!idle)
            {
            std::unique_lock<std::mutex> lk(m_mutex);
            m_cv_done.wait(lk, [this] { return (m_outstanding == 0 && m_reads_outstanding == 0) || !m_error.empty(); });
            }
        hipError_t e = hipStreamSynchronize(m_copy_stream);
        if (e != hipSuccess)
            {
            fail(std::string("stream synchronize: ") + hipGetErrorString(e));
            return PGSD_ERROR_DEVICE;
            }
        collect_timings();
        release_events();
        for (auto& a : m_arenas)
            a.used = 0;
        m_dused = 0;
        return failed() ? PGSD_ERROR_DEVICE : PGSD_SUCCESS;
        }

    bool direct_pending()
        {
        std::lock_guard<std::mutex> g(m_mutex);
        return !m_direct.empty();
        }

    // Room for `bytes` in the pinned, device-mapped arena of the direct path?  The arena is pinned once
    // (16 x the threshold, at least 8 MiB: asynchronously sealed frames may pile up in it); when it is full
    // the launch simply takes the HBM staging road.
    bool direct_reserve(size_t bytes)
        {
        if (!m_dhost)
            {
            if (m_direct_failed)
                return false;
            size_t cap = std::max((size_t)8 << 20, m_direct_max * 16);
            void* dev = nullptr;
            if (hipHostMalloc((void**)&m_dhost, cap, hipHostMallocMapped) != hipSuccess
                || hipHostGetDevicePointer(&dev, m_dhost, 0) != hipSuccess)
                {
                (void)hipGetLastError();
                if (m_dhost)
                    (void)hipHostFree(m_dhost);
                m_dhost = nullptr;
                m_direct_failed = true; // no pinned arena: everything goes through HBM staging
                return false;
                }
            m_ddev = (char*)dev;
            m_dcap = cap;
            m_dused = 0;
            }
        return m_dcap - m_dused >= bytes;
        }

    struct DirectWrite
        {
        const char* host;
        size_t bytes;
        long long file_offset;
        };

    // pwrite the packed bytes of direct chunks (their launch is known to have finished).  Chunks that follow each
    // other in the file -- a rank's share of a one-rank file, or of any chunk list whose rows this rank owns
    // alone -- go out in ONE pwritev: a small frame with the full schema is 14-20 chunks of a few KiB, and the
    // system call was most of what each of them cost.
    void write_direct(const std::vector<DirectWrite>& list)
        {
        std::vector<struct iovec> iov;
        for (size_t i = 0; i < list.size();)
            {
            size_t j = i;
            size_t bytes = 0;
            iov.clear();
            while (j < list.size() && list[j].file_offset == list[i].file_offset + (long long)bytes
                   && (j == i || m_coalesce))
                {
                if (list[j].bytes > 0)
                    iov.push_back({(void*)list[j].host, list[j].bytes});
                bytes += list[j].bytes;
                j++;
                }
            TraceRange tr("pgsd:pwrite_direct file_off=%llu bytes=%llu", (unsigned long long)list[i].file_offset, bytes);
            auto t0 = std::chrono::steady_clock::now();
            int w = iov.empty() ? 0
                    : iov.size() == 1
                        ? pwrite_locked(m_fd, iov[0].iov_base, iov[0].iov_len, list[i].file_offset, m_shared)
                        : pwritev_locked(m_fd, iov.data(), (int)iov.size(), list[i].file_offset, m_shared);
            double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (w != 0)
                {
                fail(std::string("pwrite: ") + strerror(-w), true, -w);
                return;
                }
            std::lock_guard<std::mutex> g(m_mutex);
            m_stats.written_bytes += bytes;
            m_stats.write_ms += ms;
            i = j;
            }
        }

    // Events are taken from (and returned to) a pool: creating two to four per launch was a measurable part
    // of a small frame.  tracked = handed back by release_events() with the other per-frame events.
    hipEvent_t get_event(bool timing, bool tracked = true)
        {
        hipEvent_t ev = nullptr;
            {
            std::lock_guard<std::mutex> g(m_mutex);
            auto& pool = timing ? m_pool_timing : m_pool_plain;
            if (!pool.empty())
                {
                ev = pool.back();
                pool.pop_back();
                }
            }
        if (!ev)
            {
            hipError_t e = timing ? hipEventCreate(&ev) : hipEventCreateWithFlags(&ev, hipEventDisableTiming);
            if (e != hipSuccess)
                {
                fail(std::string("hipEventCreate: ") + hipGetErrorString(e));
                return nullptr;
                }
            }
        if (tracked)
            {
            std::lock_guard<std::mutex> g(m_mutex);
            (timing ? m_misc_timing_events : m_misc_events).push_back(ev);
            }
        return ev;
        }

    // the answer words of compare(): pinned and device-mapped, plus the early-exit words in HBM.  Made by the first
    // comparison -- or by init() when the caller asked for everything up front (prealloc_mib)
    int compare_buffers()
        {
        // (each step on its own: a call that failed half-way is picked up where it stopped by the next one)
        if (!m_cmp_host)
            {
            void* h = nullptr;
            HIP_TRY(hipHostMalloc(&h, CMP_MAX_JOBS * sizeof(uint32_t), hipHostMallocMapped));
            memset(h, 0, CMP_MAX_JOBS * sizeof(uint32_t));
            m_cmp_host = (uint32_t*)h;
            }
        if (!m_cmp_host_dev)
            {
            void* hd = nullptr;
            HIP_TRY(hipHostGetDevicePointer(&hd, m_cmp_host, 0));
            m_cmp_host_dev = (uint32_t*)hd;
            }
        if (!m_cmp_dev)
            {
            void* d = nullptr;
            HIP_TRY(hipMalloc(&d, CMP_MAX_JOBS * sizeof(uint32_t)));
            // on the stream the kernels run on: a null-stream memset is not ordered with a non-blocking stream
            hipError_t me = hipMemsetAsync(d, 0, CMP_MAX_JOBS * sizeof(uint32_t), m_pack_stream);
            if (me != hipSuccess)
                {
                (void)hipFree(d);
                fail(std::string("hipMemsetAsync: ") + hipGetErrorString(me));
                return PGSD_ERROR_DEVICE;
                }
            m_cmp_dev = (uint32_t*)d;
            }
        return PGSD_SUCCESS;
        }

    int arena_alloc(size_t bytes, void** out)
        {
        size_t need = (bytes + 255) & ~(size_t)255;
        if (need == 0)
            need = 256;
        for (auto& a : m_arenas)
            if (a.cap - a.used >= need)
                {
                *out = a.base + a.used;
                a.used += need;
                return PGSD_SUCCESS;
                }
        size_t cap = std::max(need, (size_t)256 << 20);
        Arena a;
        a.base = nullptr;
        HIP_TRY(hipMalloc((void**)&a.base, cap));
        a.cap = cap;
        a.used = need;
        m_arenas.push_back(a);
        *out = a.base;
        return PGSD_SUCCESS;
        }

    void release_events()
        {
        std::lock_guard<std::mutex> g(m_mutex);
        for (auto& p : m_pack_events)
            {
            m_pool_timing.push_back(p.first);
            m_pool_timing.push_back(p.second);
            }
        m_pack_events.clear();
        for (auto& p : m_copy_events)
            {
            m_pool_timing.push_back(p.first);
            m_pool_timing.push_back(p.second);
            }
        m_copy_events.clear();
        for (auto e : m_misc_events)
            m_pool_plain.push_back(e);
        m_misc_events.clear();
        for (auto e : m_misc_timing_events)
            m_pool_timing.push_back(e);
        m_misc_timing_events.clear();
        }

    void collect_timings()
        {
        std::lock_guard<std::mutex> g(m_mutex);
        for (auto& p : m_pack_events)
            {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess)
                m_stats.pack_ms += ms;
            }
        for (auto& p : m_copy_events)
            {
            float ms = 0;
            if (hipEventElapsedTime(&ms, p.first, p.second) == hipSuccess)
                m_stats.d2h_ms += ms;
            }
        }

    // feeds the copy stream: one hipMemcpyAsync per slab-sized piece
    void dispatch_loop()
        {
        (void)hipSetDevice(m_cfg.device);
        for (;;)
            {
            CopyJob job;
                {
                std::unique_lock<std::mutex> lk(m_mutex);
                m_cv_jobs.wait(lk, [this] { return m_stop || !m_jobs.empty(); });
                if (m_jobs.empty())
                    return;
                job = m_jobs.front();
                m_jobs.pop_front();
                }
            bool bad = failed();
            if (!bad && hipStreamWaitEvent(m_copy_stream, job.packed, 0) != hipSuccess)
                {
                fail("hipStreamWaitEvent failed");
                bad = true;
                }
            hipEvent_t c0 = nullptr, c1 = nullptr;
            if (!bad && m_cfg.profile)
                {
                c0 = get_event(true, false);
                c1 = get_event(true, false);
                if (c0 && c1)
                    (void)hipEventRecord(c0, m_copy_stream);
                }
            for (size_t off = 0, n = 0; off < job.bytes; off += n)
                {
                n = piece_len(off, job.bytes, job.ramp);
                if (bad || failed())
                    {
                    bad = true;
                    piece_done(0, 0);
                    continue;
                    }
                int si = -1;
                bool grow = false;
                    {
                    std::unique_lock<std::mutex> lk(m_mutex);
                    grow = m_free_slabs.empty() && m_slabs_ready < m_cfg.n_slabs && m_error.empty();
                    if (!grow)
                        {
                        m_cv_slabs.wait(lk, [this] { return m_stop || !m_free_slabs.empty() || !m_error.empty(); });
                        if (!m_free_slabs.empty() && m_error.empty())
                            {
                            si = (int)m_free_slabs.front();
                            m_free_slabs.pop_front();
                            }
                        }
                    }
                if (grow)
                    {
                    // only this thread grows the ring; the slot exists already (the vector is full-sized)
                    hipError_t ge = alloc_slab(m_slabs[m_slabs_ready]);
                    if (ge == hipSuccess)
                        si = (int)m_slabs_ready++;
                    else
                        fail(std::string("hipHostMalloc (pinned slab): ") + hipGetErrorString(ge));
                    }
                if (si < 0)
                    {
                    bad = true;
                    piece_done(0, 0);
                    continue;
                    }
                Slab& s = m_slabs[(size_t)si];
                TraceRange tr("pgsd:d2h_enqueue file_off=%llu bytes=%llu", (unsigned long long)(job.file_offset + (long long)off), n);
                m_copy_used.store(true);
                hipError_t e = hipMemcpyAsync(s.host, job.dsrc + off, n, hipMemcpyDeviceToHost, m_copy_stream);
                if (e == hipSuccess)
                    e = hipEventRecord(s.copied, m_copy_stream);
                if (e != hipSuccess)
                    {
                    fail(std::string("hipMemcpyAsync D2H: ") + hipGetErrorString(e));
                    release_slab(si);
                    bad = true;
                    piece_done(0, 0);
                    continue;
                    }
                long long foff = job.file_offset + (long long)off;
                writer_pool_submit(m_pool, [this, si, n, foff] { write_piece(si, n, foff); });
                }
            if (c0 && c1)
                {
                (void)hipEventRecord(c1, m_copy_stream);
                std::lock_guard<std::mutex> g(m_mutex);
                m_copy_events.push_back({c0, c1});
                }
            }
        }

    int acquire_slab()
        {
        std::unique_lock<std::mutex> lk(m_mutex);
        m_cv_slabs.wait(lk, [this] { return m_stop || !m_free_slabs.empty() || !m_error.empty(); });
        if (m_free_slabs.empty() || !m_error.empty())
            return -1;
        int si = (int)m_free_slabs.front();
        m_free_slabs.pop_front();
        return si;
        }

    void read_done()
        {
        std::lock_guard<std::mutex> g(m_mutex);
        if (m_reads_outstanding > 0)
            m_reads_outstanding--;
        m_cv_done.notify_all();
        }

    // all chunks whose H2D copies are enqueued: one unpack launch per distinct row count, behind
    // the copies
    void launch_pending_unpacks()
        {
        std::vector<std::shared_ptr<ReadReq>> pending;
            {
            std::lock_guard<std::mutex> g(m_copy_mutex);
            pending.swap(m_unpack_pending);
            }
        while (!pending.empty() && !failed())
            {
            const uint64_t N = pending.front()->N;
            std::vector<pgsd_unpack_job> jobs;
            std::vector<std::shared_ptr<ReadReq>> rest;
            hipError_t e = hipSuccess;
            for (auto& r : pending)
                {
                if (r->N != N)
                    {
                    rest.push_back(r);
                    continue;
                    }
                jobs.push_back(r->job);
                if (e == hipSuccess && r->all_copied) // (direct reads have no copy to wait for)
                    e = hipStreamWaitEvent(m_pack_stream, r->all_copied, 0);
                }
            // The destinations belong to the caller: whatever its stream still has in flight on them
            // (a caching allocator hands out blocks whose previous owner may not have finished) comes
            // first, exactly as the pack waits for the producers of its sources.
            hipEvent_t ready = nullptr;
            if (e == hipSuccess)
                {
                ready = get_event(false);
                e = ready ? hipEventRecord(ready, m_source_stream) : hipErrorOutOfMemory;
                }
            if (e == hipSuccess)
                e = hipStreamWaitEvent(m_pack_stream, ready, 0);
            std::string err;
            if (e != hipSuccess)
                fail(std::string("read pipeline event: ") + hipGetErrorString(e));
            else if (launch_unpack((uint32_t)jobs.size(), jobs.data(), N, m_pack_stream, &err) != PGSD_SUCCESS)
                fail(err);
            pending.swap(rest);
            }
        }

    void read_piece(std::shared_ptr<ReadReq> req, char* dst, size_t n, long long foff)
        {
        (void)hipSetDevice(m_cfg.device);
        ReadEngine* const reader = m_reader; // the engine may outlive this pipeline, not the other way round
        int si = failed() ? -1 : reader->get_slab();
        bool last = false;
        bool ok = si >= 0;
        if (ok)
            {
            ReadEngine::Slab& s = reader->slabs[(size_t)si];
            // pread in one go; a short read means the file is shorter than its index claims
            TraceRange tr("pgsd:pread file_off=%llu bytes=%llu", (unsigned long long)foff, n);
            size_t got = 0;
            while (got < n)
                {
                ssize_t r = io_pread(m_fd, s.host + got, n - got, foff + (long long)got);
                if (r < 0 && errno == EINTR)
                    continue;
                if (r <= 0)
                    break;
                got += (size_t)r;
                }
            if (got != n)
                {
                fail("pread returned fewer bytes than the chunk holds", true);
                ok = false;
                }
            }
            {
            std::lock_guard<std::mutex> g(m_copy_mutex);
            if (ok)
                {
                ReadEngine::Slab& s = reader->slabs[(size_t)si];
                m_copy_used.store(true);
                hipError_t e = hipMemcpyAsync(dst, s.host, n, hipMemcpyHostToDevice, m_copy_stream);
                if (e == hipSuccess)
                    e = hipEventRecord(s.copied, m_copy_stream);
                if (e != hipSuccess)
                    {
                    fail(std::string("hipMemcpyAsync H2D: ") + hipGetErrorString(e));
                    ok = false;
                    }
                }
            last = (--req->pieces_left == 0);
            if (last && !failed())
                {
                // Every piece of this chunk has been enqueued on the copy stream before this point.
                // The unpack itself is deferred to wait_read(): the chunks of a frame then go through
                // ONE launch in which chunks restoring the same array are assembled into whole rows.
                hipError_t e = hipEventRecord(req->all_copied, m_copy_stream);
                if (e != hipSuccess)
                    fail(std::string("read pipeline event: ") + hipGetErrorString(e));
                else
                    m_unpack_pending.push_back(req);
                }
            }
        if (si >= 0)
            {
            if (ok)
                (void)hipEventSynchronize(reader->slabs[(size_t)si].copied);
            reader->put_slab(si);
            }
        (void)last;
        // Last touch of the pipeline by this piece: the destructor (and wait_read) wait for the count
        // of PIECES to reach zero, so no straggler of a finished chunk is left behind.
        read_done();
        }

    void write_piece(int si, size_t n, long long foff)
        {
        (void)hipSetDevice(m_cfg.device);
        Slab& s = m_slabs[(size_t)si];
        hipError_t e;
            {
            TraceRange tr("pgsd:wait_d2h file_off=%llu bytes=%llu", (unsigned long long)foff, n);
            e = hipEventSynchronize(s.copied);
            }
        double ms = 0;
        if (e != hipSuccess)
            fail(std::string("hipEventSynchronize(copy): ") + hipGetErrorString(e));
        else
            {
            TraceRange tr("pgsd:pwrite file_off=%llu bytes=%llu", (unsigned long long)foff, n);
            auto t0 = std::chrono::steady_clock::now();
            int w = pwrite_locked(m_fd, s.host, n, foff, m_shared);
            ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (w != 0)
                fail(std::string("pwrite: ") + strerror(-w), true, -w);
            }
        release_slab(si);
        piece_done(n, ms);
        }

    void release_slab(int si)
        {
            {
            std::lock_guard<std::mutex> g(m_mutex);
            m_free_slabs.push_back((uint32_t)si);
            }
        m_cv_slabs.notify_all(); // the write ring and the read ring share this condition variable
        }

    void piece_done(size_t n, double write_ms)
        {
        std::lock_guard<std::mutex> g(m_mutex);
        m_stats.d2h_bytes += n;
        m_stats.written_bytes += n;
        m_stats.write_ms += write_ms;
        if (--m_outstanding == 0)
            m_cv_done.notify_all();
        }

    pgsd_device_config m_cfg;
    int m_fd;
    bool m_shared; // other processes write the same file
    bool m_ok = false;
    hipStream_t m_pack_stream = nullptr, m_copy_stream = nullptr;
    hipStream_t m_source_stream = nullptr; // null stream unless the caller names another
    std::vector<Slab> m_slabs;
    uint32_t m_slabs_ready = 0; // slabs allocated so far (grown by the dispatcher thread only)
    std::deque<uint32_t> m_free_slabs;
    std::vector<Arena> m_arenas;
    std::deque<CopyJob> m_jobs;
    std::map<int, Staged> m_staged; // packed chunks whose file offsets are not known yet
    int m_next_ticket = 1;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> m_pack_events, m_copy_events;
    std::vector<hipEvent_t> m_misc_events, m_misc_timing_events;
    std::vector<hipEvent_t> m_pool_plain, m_pool_timing; // idle events (timing disabled / enabled)
    // direct path: pinned host arena the kernels of small launches pack into (host and device view)
    char* m_dhost = nullptr;
    char* m_ddev = nullptr;
    size_t m_dcap = 0, m_dused = 0, m_direct_max = 0;
    bool m_direct_failed = false;
    uint32_t* m_cmp_host = nullptr;        // compare(): answers, pinned; its device alias; the early-exit words in HBM
    uint32_t* m_cmp_host_dev = nullptr;
    uint32_t* m_cmp_dev = nullptr;
    uint32_t m_cmp_gen = 0;
    bool m_coalesce = true;                // neighbours in the file leave in one pwritev (write_direct)
    size_t m_soft_cap = (size_t)6 << 30;   // staging held by frames on their way before stage() waits (PGSD_STAGING_CAP_MIB)
    std::vector<DirectWrite> m_direct;     // committed direct chunks waiting for their pwrite (m_mutex)
    WriterPool* m_pool = nullptr;
    ReadEngine* m_reader = nullptr; // shared reader threads + pinned ring of this device
    std::atomic<bool> m_copy_used {false}; // something was enqueued on the copy stream since drain() last synchronised it
    std::mutex m_copy_mutex; // serialises enqueues on the copy / pack streams from reader threads
    size_t m_reads_outstanding = 0;
    std::vector<std::shared_ptr<ReadReq>> m_unpack_pending; // guarded by m_copy_mutex
    std::thread m_dispatcher;
    std::mutex m_mutex;
    std::condition_variable m_cv_jobs, m_cv_slabs, m_cv_done;
    size_t m_outstanding = 0;
    bool m_stop = false;
    std::string m_error;
    bool m_io_error = false;
    bool m_numa = false;    // m_numa_cpus = CPUs of the GPU's NUMA node (two-socket hosts)
    cpu_set_t m_numa_cpus;
    int m_io_errno = 0; // errno of the failed write (worker thread), handed to the caller's thread
    pgsd_device_stats m_stats = {};
    };

DevicePipeline* device_pipeline_create(const pgsd_device_config& cfg, int fd, bool shared_file, std::string* err)
    {
    DevicePipeline* p = new DevicePipeline(cfg, fd, shared_file);
    if (p->init() != PGSD_SUCCESS)
        {
        if (err)
            *err = p->error();
        delete p;
        return nullptr;
        }
    return p;
    }

void device_pipeline_destroy(DevicePipeline* p)
    {
    delete p;
    }

int device_pipeline_device(DevicePipeline* p)
    {
    return p ? p->device() : -1;
    }

int device_pipeline_submit(DevicePipeline* p, std::vector<DeviceChunk>& chunks, uint64_t N, std::string* err)
    {
    int rc = p->submit(chunks, N);
    if (rc != PGSD_SUCCESS && err)
        *err = p->error();
    return rc;
    }

int device_pipeline_stage(DevicePipeline* p, std::vector<DeviceChunk>& chunks, uint64_t N, int* ticket, std::string* err)
    {
    int rc = p->stage(chunks, N, ticket);
    if (rc != PGSD_SUCCESS && err)
        *err = p->error();
    return rc;
    }

int device_pipeline_commit(DevicePipeline* p, int ticket, size_t index, long long file_offset, void* host_dst,
                           std::string* err)
    {
    int rc = p->commit(ticket, index, file_offset, host_dst);
    if (rc != PGSD_SUCCESS && err)
        *err = p->error();
    return rc;
    }

int device_pipeline_compare(DevicePipeline* p, int ticket, size_t first, size_t count, const void* const* ref,
                            const uint64_t* ref_bytes, uint8_t* equal, std::string* err)
    {
    int rc = p->compare(ticket, first, count, ref, ref_bytes, equal);
    if (rc != PGSD_SUCCESS && err)
        *err = p->error();
    return rc;
    }

int device_pipeline_copy_staged(DevicePipeline* p, int ticket, size_t first, size_t count, void* const* dst,
                                std::string* err)
    {
    int rc = p->copy_staged(ticket, first, count, dst);
    if (rc != PGSD_SUCCESS && err)
        *err = p->error();
    return rc;
    }

void device_pipeline_kick(DevicePipeline* p)
    {
    p->kick_direct();
    }

void device_pipeline_write_host(DevicePipeline* p, const void* data, size_t bytes, long long file_offset)
    {
    p->write_host(data, bytes, file_offset);
    }

bool device_pipeline_single_writer(DevicePipeline* p)
    {
    return p->single_writer();
    }

int device_pipeline_wait_packed(DevicePipeline* p, std::string* err)
    {
    int rc = p->wait_packed();
    if (rc != PGSD_SUCCESS && err)
        *err = p->error();
    return rc;
    }

int device_pipeline_drain(DevicePipeline* p, std::string* err)
    {
    int rc = p->drain();
    if (rc != PGSD_SUCCESS && err)
        *err = p->error();
    return rc;
    }

int device_pipeline_read(DevicePipeline* p, long long file_offset, size_t bytes, const pgsd_unpack_job& job, uint64_t N,
                         std::string* err)
    {
    int rc = p->read_submit(file_offset, bytes, job, N);
    if (rc != PGSD_SUCCESS && err)
        *err = p->error();
    return rc;
    }

int device_pipeline_wait_read(DevicePipeline* p, std::string* err)
    {
    int rc = p->wait_read();
    if (rc != PGSD_SUCCESS && err)
        *err = p->error();
    return rc;
    }

void device_pipeline_set_source_stream(DevicePipeline* p, void* stream)
    {
    p->set_source_stream(stream);
    }

void device_pipeline_stats(DevicePipeline* p, pgsd_device_stats* out, int reset)
    {
    p->stats(out, reset);
    }

    } // namespace pgsd_amd

extern "C" int pgsd_device_release_parked(void)
    try
    {
    return pgsd_amd::release_parked_sets();
    }
catch (...)
    {
        pgsd_amd::abi_guard();
        return 0;
    }
