// pgsd_internal.hpp -- declarations shared by the translation units of libpgsd_amd.so.
#ifndef PGSD_INTERNAL_HPP
#define PGSD_INTERNAL_HPP

#include "pgsd.h"
#include "pgsd_private.h"

#include <cstdint>
#include <sched.h>
#include <sys/uio.h>
#include <functional>
#include <memory>
#include <string>
#include <vector>

namespace pgsd_amd
    {
void set_last_error(const std::string& s);
const char* last_error();
uint64_t last_error_serial(); // grows with every set_last_error of this thread: "did the callee leave a message?"
// Called from the catch-all of every C-ABI entry point (function-try-blocks): no C++ exception
// crosses the boundary.  Maps the exception in flight to a pgsd_error and records its text.
int abi_guard() noexcept;
pgsd_comm default_comm();
// The installed communicator is shared by reference: a handle keeps the one it was opened with alive, so
// pgsd_comm_finalize / pgsd_comm_init_* while files are open cannot pull the context (shm mapping, RCCL
// communicator, callbacks) from under them; its destroy hook runs when the last user lets go.
struct CommBox
    {
    pgsd_comm c;
    explicit CommBox(const pgsd_comm& comm) : c(comm) { }
    CommBox(const CommBox&) = delete;
    CommBox& operator=(const CommBox&) = delete;
    ~CommBox()
        {
        if (c.destroy)
            c.destroy(c.ctx);
        }
    };
std::shared_ptr<CommBox> default_comm_box();

// ---- phase timeline: roctx ranges around the phases of a frame (pack launch, frame exchange, each
// device->host piece, each pwrite, drain), the counterpart of the reference's PGSD_ACTIVATE_LOGGER
// output (pgsd.c:27, 1034, 1156, 2231).  Off unless PGSD_TRACE is set to a non-zero value; then
// librocprofiler-sdk-roctx is dlopen'ed (no link-time dependency) and `rocprofv3 --marker-trace` shows
// the ranges next to the kernels and copies they bracket.
bool trace_on();
void trace_push(const char* name);
void trace_pop();
struct TraceRange
    {
    bool on;
    explicit TraceRange(const char* name) : on(trace_on())
        {
        if (on)
            trace_push(name);
        }
    // name with one number (piece offset, byte count ...): formatted only when tracing
    TraceRange(const char* fmt, unsigned long long a, unsigned long long b = 0);
    TraceRange(const TraceRange&) = delete;
    TraceRange& operator=(const TraceRange&) = delete;
    ~TraceRange()
        {
        if (on)
            trace_pop();
        }
    };

inline int comm_barrier(const pgsd_comm& c)
    {
    if (c.size == 1)
        return PGSD_SUCCESS;
    if (c.barrier)
        return c.barrier(c.ctx) == 0 ? PGSD_SUCCESS : PGSD_ERROR_COMM;
    std::vector<char> all((size_t)c.size);
    char one = 0;
    return c.allgather(c.ctx, &one, all.data(), 1) == 0 ? PGSD_SUCCESS : PGSD_ERROR_COMM;
    }

// ---- file back ends (pgsd_io.cpp): POSIX, or the reference's own MPI-IO calls with PGSD_IO=mpiio.  The descriptor of
// io_open is what every other io_* call (and pwrite_full / pread_some below, which sit on them) takes.
int io_open(const char* path, int oflags, int mode);
int io_close(int fd);
int io_truncate(int fd, long long size);
long long io_file_size(int fd); // -1 + errno
ssize_t io_pwrite(int fd, const void* buf, size_t bytes, long long offset);
ssize_t io_pread(int fd, void* buf, size_t bytes, long long offset);
const char* io_backend_name(); // "posix" | "mpiio"

// ---- host IO: a pool of pwrite threads shared by the host and the device path ----
class WriterPool;
WriterPool* writer_pool_create(unsigned n_threads, const cpu_set_t* cpus = nullptr);
bool numa_cpus_of_pci_device(const char* pci_bus_id, cpu_set_t* out);
void writer_pool_destroy(WriterPool*);
void writer_pool_submit(WriterPool*, std::function<void()> fn);
// Write [buf, buf+bytes) at `offset` of fd, split over the pool; blocks until done.
// Returns 0 or -errno.
int writer_pool_pwrite_sync(WriterPool*, int fd, const void* buf, size_t bytes, long long offset,
                            bool shared_file = false);
// pwrite_full under an advisory flock when several processes write the same file
int pwrite_locked(int fd, const void* buf, size_t bytes, long long offset, bool shared_file);
int pwritev_locked(int fd, struct iovec* iov, int n, long long offset, bool shared_file); // one contiguous file range
// plain full-length pwrite / pread loops (0 / -errno; pread leaves a short tail untouched)
int pwrite_full(int fd, const void* buf, size_t bytes, long long offset);
void pread_some(int fd, void* buf, size_t bytes, long long offset);
void pread_parallel(int fd, void* buf, size_t bytes, long long offset); // >= 64 MiB: a few threads

// ---- device pipeline (pgsd_device.cpp); created lazily by the first device call ----
class DevicePipeline;
struct DeviceChunk
    {
    pgsd_pack_job job;      // dst filled in by the pipeline (device staging)
    uint64_t N;             // rows of this rank
    long long file_offset;  // where this rank's rows start in the file; <0: copy into host_dst
    void* host_dst;         // for small buffered chunks: synchronous copy target
    };
DevicePipeline* device_pipeline_create(const pgsd_device_config& cfg, int fd, bool shared_file, std::string* err);
void device_pipeline_destroy(DevicePipeline*);
int device_pipeline_device(DevicePipeline*); // the HIP device the pipeline runs on
// one fused pack launch for `chunks` (all share N), then async copy + write of each
int device_pipeline_submit(DevicePipeline*, std::vector<DeviceChunk>& chunks, uint64_t N, std::string* err);
// the same in two steps: pack now (one fused launch, returns a ticket), say later where chunk `index` of the
// ticket goes: a file offset, a host buffer (synchronous copy), or nowhere (file_offset < 0, host_dst null)
int device_pipeline_stage(DevicePipeline*, std::vector<DeviceChunk>& chunks, uint64_t N, int* ticket, std::string* err);
int device_pipeline_commit(DevicePipeline*, int ticket, size_t index, long long file_offset, void* host_dst,
                           std::string* err);
// staged, not yet committed chunks [first, first + count) of a ticket: packed rows == ref[i] (device memory; shorter
// than the chunk: repeating)?
// (one kernel + one stream wait) / copied into dst[i] (device memory, asynchronous on the pack stream)
int device_pipeline_compare(DevicePipeline*, int ticket, size_t first, size_t count, const void* const* ref,
                            const uint64_t* ref_bytes, uint8_t* equal, std::string* err);
int device_pipeline_copy_staged(DevicePipeline*, int ticket, size_t first, size_t count, void* const* dst,
                                std::string* err);
// asynchronous seal: chunks of the direct (small-frame) path are handed to the writer thread now
void device_pipeline_kick(DevicePipeline*);
// a few host bytes (metadata of an asynchronously sealed frame) through the writer thread, in FIFO order behind the
// pieces already queued; copied.  single_writer: that order is only defined with one writer thread
void device_pipeline_write_host(DevicePipeline*, const void* data, size_t bytes, long long file_offset);
bool device_pipeline_single_writer(DevicePipeline*);
int device_pipeline_wait_packed(DevicePipeline*, std::string* err);
void device_pipeline_set_source_stream(DevicePipeline*, void* stream);
// read side: rows at `file_offset` -> staging -> unpack into job.dst (job.src is filled in)
int device_pipeline_read(DevicePipeline*, long long file_offset, size_t bytes, const pgsd_unpack_job& job, uint64_t N,
                         std::string* err);
int device_pipeline_wait_read(DevicePipeline*, std::string* err);
int device_pipeline_drain(DevicePipeline*, std::string* err);
void device_pipeline_stats(DevicePipeline*, pgsd_device_stats* out, int reset);

size_t sizeof_type(uint32_t type);
    } // namespace pgsd_amd

#endif
