/* pgsd_private.h -- entry points of libpgsd_amd.so that are NOT part of the public C ABI (include/pgsd.h).
 *
 * They are exported from the library because the parity tests, bench.py's kernel legs, the tools under tools/ and the
 * Cython binding reach them across the shared-object boundary, but no caller of the file format needs them and they
 * may change without a PGSD_ABI_VERSION bump:
 *
 *   bare kernels      pgsd_pack_fields, pgsd_unpack_fields: the HIP kernels without a file (tests/test_gpu_pack.py,
 *                     bench_legs.py, tools/pack_bench.py, tools/unpack_bench.py)
 *   queue plumbing    pgsd_frame_exchange (the scenario driver's `dump`), pgsd_set_deferred_rows (pgsd.fl only: it
 *                     owns the host arrays it queues)
 *   housekeeping      pgsd_device_release_parked, pgsd_reload_tuning
 *   binding helpers   pgsd_device_of, pgsd_device_copy (pgsd.fl keeps library-owned device buffers on the pipeline's
 *                     device and clones them without a tensor library)
 */
#ifndef PGSD_PRIVATE_H
#define PGSD_PRIVATE_H

#include "pgsd.h"

#ifdef __cplusplus
extern "C"
    {
#endif

    /* Bare kernels, no file: pack into caller-provided device buffers on `stream`
       (a hipStream_t passed as void*; NULL = the null stream). */
    struct pgsd_pack_job
        {
        void* dst;         /* device pointer, N*M elements of dst_type, 16-byte aligned */
        uint32_t dst_type; /* enum pgsd_type */
        uint32_t M;
        struct pgsd_field_desc src;
        };
    /* kernel_ms (may be NULL): receives the time from the begin of the first to the end of the last kernel of the call
       as the dispatches themselves stamp it (what rocprofv3 reports per kernel; no launch latency); the call then
       synchronises `stream`.  Measurement only. */
    int pgsd_pack_fields(uint32_t n_jobs, const struct pgsd_pack_job* jobs, uint64_t N, void* stream, float* kernel_ms);

    /* Bare unpack kernel: dense chunk rows already in device memory -> destination arrays. */
    struct pgsd_unpack_job
        {
        const void* src;   /* device pointer, N*M elements of src_type, 16-byte aligned */
        uint32_t src_type; /* enum pgsd_type of the chunk */
        uint32_t M;
        struct pgsd_field_dst dst;
        };
    int pgsd_unpack_fields(uint32_t n_jobs, const struct pgsd_unpack_job* jobs, uint64_t N, void* stream);

    /* Batched mode only: with `on`, the host rows of pgsd_write_chunk(..., all == true, data) are BORROWED UNTIL
       THE FRAME'S EXCHANGE instead of for the call -- the caller promises to leave them alone until the next
       pgsd_end_frame / pgsd_flush / pgsd_frame_exchange / pgsd_close (the contract device sources have anyway).
       Such a chunk then waits in the queue like every other instead of resolving it at once, and a frame costs
       ONE exchange whatever mix of host and device chunks it holds.  (The reference's contract -- rows borrowed
       for the call -- is the default; pgsd.fl turns this on where it holds the arrays itself.) */
    int pgsd_set_deferred_rows(struct pgsd_handle* handle, int on);

    /* Perform the batched frame exchange now (collective; nothing is flushed): afterwards the queue is empty and
       the handle's mirror is current.  No-op when nothing is queued. */
    int pgsd_frame_exchange(struct pgsd_handle* handle);

    /* Give every parked pipeline set (include/pgsd.h, end of part 3) back to the runtime now; returns the number of
       sets freed.  Not to be called while another thread opens or closes a handle. */
    int pgsd_device_release_parked(void);

    /* The PGSD_* tuning variables (launch shapes, kernel choices, direct-path threshold ...) are read ONCE, when the
       library first needs them.  Tools that A/B variants inside one process change the environment and call this to
       have it read again. */
    void pgsd_reload_tuning(void);

    /* The HIP device the handle's pipeline runs on (the pipeline is created if it does not exist yet: the device
       configured with pgsd_device_configure, else the current one at that moment); negative: a pgsd_error. */
    int pgsd_device_of(struct pgsd_handle* handle);

    /* `bytes` bytes from src to dst: device memory on `device` (-1: the current one) or host memory, either side
       (hipMemcpyDefault); complete on return. */
    int pgsd_device_copy(int device, void* dst, const void* src, size_t bytes);

#ifdef __cplusplus
    }
#endif

#endif /* PGSD_PRIVATE_H */
