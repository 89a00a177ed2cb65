#!/usr/bin/env python3
"""bench.py -- snapshot pack+write throughput of the MI355X-native PGSD writer.

One "step" = one frame of the hot path on one batch of synthetic particle data that is
already resident in HBM: fused HIP pack of position / velocity / typeid from HOOMD-style float4
arrays, ONE allgather of the ranks' chunk sizes (RCCL over xGMI for N > 1; gives every rank's row
count -> file offsets) in flight while the kernel runs, hipMemcpyAsync to pinned slabs, pwrite
into one shared GSD file on tmpfs, index commit (pgsd_end_frame).

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU; the ranks share ONE output file, each writing its own partition (weak
scaling: 10 M particles per GPU).  Either the driver launches the ranks (torch.distributed.run:
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment, and WORLD_SIZE must equal --gpus), or
-- WORLD_SIZE unset -- this script starts them itself: the parent process spawns N rank children
BEFORE anything touches torch or the GPU, relays rank 0's JSON line and exits with their status
(the reference's harness finds its ranks the same way inside the program, benchmark-write.cc:24-45).
Rank 0 prints one JSON line (contract in the task description) with `roofline` for the pack
kernel, `exchange_us` for the frame's one allgather and `cpu_baseline` for the reference's CPU
path on the host's cores (P ranks for N = P GPUs; `nproc` stated).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip-level parameters)
ALGO_BYTES_PER_PARTICLE = 56   # read pos.xyz + vel.xyz + typeid (28 B) + write the three chunks (28 B)
PAYLOAD_BYTES_PER_PARTICLE = 28


def numa_local_cores(n_ranks):
    """One host core per rank, taken from the NUMA-local cores of GPU r (r = 0 .. n_ranks - 1; sysfs
    /sys/bus/pci/devices/<bdf>/local_cpulist) -- SURVEY 8(d): the CPU baseline runs 'one host process per GPU pinned to
    that GPU's NUMA-local cores'.  None when the topology cannot be read (then the ranks are bound to consecutive cores)."""
    try:
        import torch

        def cpus(text):
            out = []
            for part in text.strip().split(","):
                a, _, b = part.partition("-")
                out += list(range(int(a), int(b or a) + 1))
            return out

        allowed = set(os.sched_getaffinity(0))
        taken, cores = set(), []
        for r in range(n_ranks):
            p = torch.cuda.get_device_properties(r % max(torch.cuda.device_count(), 1))
            bdf = "%04x:%02x:%02x.0" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
            with open("/sys/bus/pci/devices/%s/local_cpulist" % bdf) as fh:
                local = [c for c in cpus(fh.read()) if c in allowed and c not in taken]
            if not local:
                return None
            cores.append(local[0])
            taken.add(local[0])
        return cores
    except Exception:  # no sysfs entry, an older torch without the pci_* properties, a masked affinity ...
        return None


def cpu_baseline_reference(n_particles, frames, out_dir, ranks=1, schema="pvi"):
    """Time the reference ITSELF (oracle/_ref/ref_bench = the reference's pgsd.c compiled in the
    build container + tests/drivers/ref_bench.c) under MPICH on this host's cores: `ranks` MPI ranks
    (one core each, one per GPU of the run, as SURVEY 8(d) prescribes: rank r pinned to a core of GPU r's NUMA node,
    `mpiexec -bind-to user:<cores>`; consecutive cores, `-bind-to core`, where the topology cannot be read) writing
    `n_particles` each, plus a second run on more ranks
    (min(4 * ranks, nproc)) of the same total workload to show what more cores buy.  None when it cannot run."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        return None
    nproc = os.cpu_count() or 1
    more = min(4 * ranks, nproc)
    path = os.path.join(out_dir, "pgsd_bench_ref_%d.gsd" % os.getpid())
    res = {}
    cores = numa_local_cores(ranks)
    binding = ("mpiexec -bind-to user:%s (rank r on a core of GPU r's NUMA node)" % ",".join(map(str, cores))
               if cores else "mpiexec -bind-to core (GPU topology not readable: not NUMA-matched)")
    try:
        for r in sorted({ranks, more}):
            bind = ["-bind-to", "user:" + ",".join(map(str, cores))] if (cores and r == ranks) else ["-bind-to", "core"]
            out = subprocess.run([mpiexec] + bind + ["-n", str(r), exe, str(n_particles * ranks), str(frames),
                                  path, schema], capture_output=True, timeout=300, check=True).stdout.decode()
            res[r] = json.loads(out.strip().splitlines()[-1])
    except Exception as e:  # missing MPI runtime, time-out, ...
        print("bench.py: reference baseline unavailable (%s), using the oracle port" % e, file=sys.stderr)
        return None
    finally:
        if os.path.exists(path):
            os.unlink(path)
    return {"value": round(res[ranks]["GBps"], 4), "unit": "GB/s", "cores": ranks, "kind": "reference",
            "nproc": nproc, "ranks": ranks,
            "more_cores": {"ranks": more, "value": round(res[more]["GBps"], 4)},
            "binding": binding, "schema": schema,
            "sample": "%d frames x %d particles per rank (%s): C pack loop out of float4 arrays + the reference's "
                      "pgsd_write_chunk/pgsd_end_frame (MPICH 3.3.2 MPI-IO), %d rank(s) = %d bound core(s) of %d (-bind-to), "
                      "one shared file on %s; same workload on %d ranks: %.3f GB/s"
                      % (frames, n_particles, {"pvi": "pos+vel+typeid", "sph": "14 SPH chunks, 112 B/particle",
                                               "union": "19 chunks SPH + upstream HOOMD, 164 B/particle"}[schema],
                         ranks, ranks, nproc, out_dir, more, res[more]["GBps"])}


def cpu_baseline(n_particles, frames, out_dir):
    """Time the CPU restatement of the reference path (oracle/, kind="port") on host cores:
    pack float4 -> N x 3 with the C loop a CPU caller runs, then the reference's
    write_chunk/end_frame sequence, single thread like one MPI rank of the reference."""
    import numpy as np
    import scenario as S
    lib = S.oracle_lib()
    rng = np.random.default_rng(1234)
    pos = ((rng.random((n_particles, 4), dtype=np.float32) - 0.5) * 100.0).astype(np.float32)
    vel = rng.standard_normal((n_particles, 4), dtype=np.float32)
    tid = rng.permutation(n_particles).astype(np.uint32).reshape(-1, 1)
    o_pos = np.empty((n_particles, 3), dtype=np.float32)
    o_vel = np.empty((n_particles, 3), dtype=np.float32)
    path = os.path.join(out_dir, "pgsd_bench_cpu_%d.gsd" % os.getpid())
    rc = ctypes.c_int(0)
    h = lib.oracle_create_and_open(path.encode(), 1, b"bench", b"hoomd", lib.oracle_make_version(1, 4), 1, 0,
                                   ctypes.byref(rc))
    assert rc.value == 0

    def frame(i):
        lib.oracle_pack_rows(o_pos.ctypes.data, 9, pos.ctypes.data, 9, n_particles, 3, 4, 0, None, 0)
        lib.oracle_pack_rows(o_vel.ctypes.data, 9, vel.ctypes.data, 9, n_particles, 3, 4, 0, None, 0)
        step = np.array([[i]], dtype=np.uint64)
        S.oracle_write_chunk(lib, h, "configuration/step", 4, [step], 1, 1, 1, [0], [1], False)
        for name, arr, t, M in (("particles/position", o_pos, 9, 3), ("particles/velocity", o_vel, 9, 3),
                                ("particles/typeid", tid, 3, 1)):
            S.oracle_write_chunk(lib, h, name, t, [arr], M, n_particles, M, [0], [n_particles * M], True)
        assert lib.oracle_end_frame(h) == 0

    frame(0)  # warm-up (first touch of the output arrays and the file)
    t0 = time.perf_counter()
    for i in range(frames):
        frame(i + 1)
    dt = time.perf_counter() - t0
    lib.oracle_close(h)
    os.unlink(path)
    gbs = frames * n_particles * PAYLOAD_BYTES_PER_PARTICLE / dt / 1e9
    return {"value": round(gbs, 4), "unit": "GB/s", "cores": 1, "kind": "port", "nproc": os.cpu_count() or 1, "ranks": 1,
            "sample": "%d frames x %d particles (pos+vel+typeid), float4 -> chunk pack in C + "
                      "reference write sequence, 1 thread, file on %s" % (frames, n_particles, out_dir)}


PACK_KERNEL = "pack_rows_kernel"   # the dominant kernel of the default layout (pgsd_pack.hip); dense scalar
                                   # arrays of the full schemas additionally go through pack_copy_kernel


def pack_source_sha256():
    """Hash of the pack kernels' source: pgsd_pack.hip and the device helpers it shares (pgsd_kernels.hpp)."""
    import hashlib
    h = hashlib.sha256()
    for name in ("pgsd_pack.hip", "pgsd_kernels.hpp"):
        with open(os.path.join(ROOT, "pgsd-sph_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def pmc_pass(counter, child_args, timeout):
    """One rocprofv3 counter pass over a short child run of this script; returns
    {kernel name: [value per dispatch]} for the pack kernels, or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    d = tempfile.mkdtemp(prefix="pgsd_pmc_", dir="/tmp")
    try:
        # the program itself follows `--` (no env/bash hop: the profiler's library has the GPU initialised)
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.abspath(__file__)] + child_args
        env = {k: v for k, v in os.environ.items() if k != "PGSD_IO"}      # the counter passes write through the POSIX back end
        r = subprocess.run(cmd, timeout=timeout, capture_output=True, cwd="/tmp", env=dict(env, TMPDIR="/tmp"))
        if r.returncode != 0:
            return None
        out = {}
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as f:
                for row in csv.DictReader(f):
                    name = row.get("Kernel_Name", "")
                    if row.get("Counter_Name") == counter and ("pgsd_amd::pack_" in name):
                        out.setdefault(name, []).append(float(row["Counter_Value"]))
        return out or None
    except Exception as e:  # time-out, unreadable output, ...
        print("bench.py: %s pass failed (%s)" % (counter, e), file=sys.stderr)
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def under_a_profiler():
    """Is this process itself running under rocprofv3 (its tool library preloaded)?  Then no counter passes are started
    from inside it: a profiler's child would inherit the outer tool next to its own."""
    return ("rocprofiler" in os.environ.get("LD_PRELOAD", "") or "ROCP_TOOL_LIBRARIES" in os.environ
            or any(k.startswith("ROCPROF_") for k in os.environ))


def measure_traffic_live(args):
    """-> (HBM bytes per frame's pack launch(es), source text, detail) from two PMC passes, or Nones."""
    if under_a_profiler():
        print("bench.py: running under a profiler: no nested counter passes (roofline.traffic from profiles/pack_traffic.json)",
              file=sys.stderr)
        return None, None, None
    child = ["--gpus", "1", "--steps", "4", "--warmup", "1", "--particles", str(args.particles), "--schema", args.schema,
             "--dir", args.dir, "--no-cpu-baseline", "--traffic", "off", "--no-stall-test", "--no-exchange-probe", "--no-legs"]
    if args.separate_id:
        child.append("--separate-id")
    fetch = pmc_pass("FETCH_SIZE", child, 240)
    write = pmc_pass("WRITE_SIZE", child, 240) if fetch else None
    if not fetch or not write:
        return None, None, None
    import statistics
    # per frame: the median dispatch of every pack kernel of a frame, summed over the kernels
    fetch_kib = sum(statistics.median(v) for v in fetch.values())
    write_kib = sum(statistics.median(v) for v in write.values())
    read_bytes = int(fetch_kib * 1024 * 2)     # gfx950: FETCH_SIZE counts half of a wide coalesced read stream
    write_bytes = int(write_kib * 1024)
    detail = {"FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib, "read_bytes": read_bytes,
              "write_bytes": write_bytes, "kernels": sorted(fetch),
              "corrections": "bytes = KiB*1024; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM)"}
    return read_bytes + write_bytes, "live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of this run", detail


def traffic_from_file(N, args):
    """The committed PMC result, only while it describes the kernel source that is being run."""
    try:
        with open(os.path.join(ROOT, "profiles", "pack_traffic.json")) as tf:
            tj = json.load(tf)
        if tj.get("particles") == N and args.schema == "pvi" and not args.separate_id:
            if tj.get("pack_source_sha256") != pack_source_sha256():
                return None, "profiles/pack_traffic.json is older than pgsd_pack.hip / pgsd_kernels.hpp: not reported"
            return tj["hbm_bytes_per_launch"], "file: profiles/pack_traffic.json (" + tj.get("source", "") + ")"
    except (OSError, ValueError, KeyError):
        pass
    return None, None


def probe_rccl_exchange(device, n=300, nbytes=512):
    """Median / p99 wall time (us) of the library's allgather on a ONE-rank RCCL communicator, or None."""
    try:
        from pgsd import _lib
        lib = _lib.lib
        uid = (ctypes.c_uint8 * 128)()
        import pgsd.dist as pdist
        if lib.pgsd_comm_rccl_unique_id(uid) != 0 or pdist.init_rccl(bytes(uid), 0, 1, int(device)) != 0:
            print("bench.py: no RCCL exchange probe (%s)" % _lib.last_error(), file=sys.stderr)
            return None
        send = (ctypes.c_uint8 * nbytes)()
        recv = (ctypes.c_uint8 * nbytes)()
        times = []
        for i in range(n + 20):
            t0 = time.perf_counter()
            rc = lib.pgsd_comm_allgather(send, recv, nbytes)
            dt = time.perf_counter() - t0
            if rc != 0:
                lib.pgsd_comm_finalize()
                return None
            if i >= 20:
                times.append(dt * 1e6)
        lib.pgsd_comm_finalize()
        times.sort()
        return {"backend": "rccl, 1 rank (no xGMI hop)", "bytes": nbytes, "samples": n,
                "median_us": round(times[len(times) // 2], 1), "p99_us": round(times[int(len(times) * 0.99) - 1], 1),
                "min_us": round(times[0], 1)}
    except Exception as e:  # measurement aside: never fails the run
        print("bench.py: RCCL exchange probe failed (%s)" % e, file=sys.stderr)
        return None


def fstype_of(path):
    """File system type of the mount `path` lives on (/proc/mounts, longest mount point that is a prefix)."""
    try:
        real = os.path.realpath(path)
        best, kind = "", None
        with open("/proc/mounts") as f:
            for line in f:
                parts = line.split()
                if len(parts) < 3:
                    continue
                mp = parts[1].replace("\\040", " ")
                if (real == mp or real.startswith(mp.rstrip("/") + "/")) and len(mp) >= len(best):
                    best, kind = mp, parts[2]
        return kind
    except OSError:
        return None


def overlap_slowdown(f, fields, np, torch, frames=4):
    """How much slower does a kernel run while a snapshot drains?  A queue of identical HBM-bound launches (copy of a
    512 MiB tensor: every CU busy, 1 GiB of traffic each) on a stream of its own is timed with HIP events ALONE and
    WHILE `frames` frames are appended with asynchronous seals and drained (`frame_sync`): the pack kernels, the
    device->host blit kernels and the pwrite()s run beside it.  tools/overlap_probe.py is the long form (also an
    fp32 GEMM queue); profiles/r04_overlap_*.  The reference has no counterpart: its data is host resident."""
    try:
        src = torch.empty(1 << 27, dtype=torch.float32, device="cuda").normal_()
        dst = torch.empty_like(src)
        side = torch.cuda.Stream()

        def drain():
            t0 = time.perf_counter()
            for i in range(frames):
                f.write_chunk("configuration/step", np.array([2 * 10 ** 6 + i], dtype=np.uint64), write_all=False)
                f.write_chunks(fields, offset="auto")
                f.end_frame(wait=False)
                f.wait_packed()
            f.frame_sync()
            return time.perf_counter() - t0

        def queue(n):
            with torch.cuda.stream(side):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(n):
                    dst.copy_(src)
                e1.record()
            return e0, e1

        t_drain = drain()
        e0, e1 = queue(20)
        side.synchronize()
        per = e0.elapsed_time(e1) / 20.0
        n = max(20, int(t_drain * 1e3 * 0.9 / per))          # a queue about as long as the drain
        alone, under, cover = [], [], []
        for _ in range(2):
            e0, e1 = queue(n)
            side.synchronize()
            alone.append(e0.elapsed_time(e1) / n)
        for _ in range(2):
            e0, e1 = queue(n)
            dt = drain()
            side.synchronize()
            ms = e0.elapsed_time(e1)
            under.append(ms / n)
            cover.append(min(1.0, dt * 1e3 / ms))
        a, u = min(alone), min(under)
        return {"slowdown_pct": round((u / a - 1.0) * 100.0, 2), "kernel": "copy of 512 MiB (1 GiB of HBM traffic per launch)",
                "launches": n, "ms_per_launch_alone": round(a, 4), "ms_per_launch_under_drain": round(u, 4),
                "frames_drained": frames, "drain_covers_fraction_of_queue": round(min(cover), 3)}
    except Exception as e:  # a measurement aside: never fails the run
        print("bench.py: overlap measurement failed (%s)" % e, file=sys.stderr)
        return None


def launch_ranks(n, argv):
    """Parent of a self-launched N-rank run: start N children of this script with RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set (what torch.distributed.run would export), pass rank 0's stdout through and
    return a non-zero status if any rank fails.  The parent never imports torch and never touches the GPU."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    import signal

    def stop_children(signum, frame):           # the launcher is told to stop: so are its ranks (exactly these PIDs)
        for q in procs:
            if q.poll() is None:
                q.terminate()
        raise SystemExit(128 + signum)

    for sig in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sig, stop_children)
    status = 0
    pending = list(procs)
    failed_at = None
    try:
        while pending:
            for p in list(pending):
                rc = p.poll()
                if rc is None:
                    continue
                pending.remove(p)
                if rc != 0 and status == 0:
                    status = rc if rc > 0 else 1
                    failed_at = time.monotonic()
            # a rank that died leaves the others inside a collective: they get a few seconds to say why they stop
            # (the same refusal, a communicator time-out), then they are ended (exactly these children)
            if failed_at is not None and pending and time.monotonic() - failed_at > 5.0:
                for q in pending:
                    q.terminate()
                failed_at = float("inf")
            if pending:
                time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.terminate()
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles", type=int, default=10_000_000, help="particles per GPU")
    ap.add_argument("--dir", default=os.environ.get("PGSD_BENCH_DIR", "/dev/shm"))
    ap.add_argument("--dir-fallback", action="store_true",
                    help="when --dir lacks room for the run, write to /tmp or the working directory instead of stopping "
                         "(the file system then differs from run to run: config.target_fstype says which it was)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-stall-test", action="store_true", help="skip the asynchronous-sealing stall measurement")
    ap.add_argument("--no-exchange-probe", action="store_true",
                    help="N=1: skip timing one exchange on a one-rank RCCL communicator")
    ap.add_argument("--slab-mib", type=int, default=0)
    ap.add_argument("--slabs", type=int, default=0)
    ap.add_argument("--writers", type=int, default=0)
    ap.add_argument("--separate-id", action="store_true", help="typeid from its own uint32 array instead of pos.w")
    ap.add_argument("--schema", choices=["pvi", "sph", "union"], default="pvi",
                    help="pvi: position+velocity+typeid (headline); sph: the full PGSD-SPH particle schema, "
                         "112 B/particle in 15 chunks; union: plus the upstream HOOMD attributes, 164 B/particle "
                         "(BASELINE config 4 workloads)")
    ap.add_argument("--comm", choices=["auto", "rccl", "torch"], default="auto",
                    help="N>1: auto = the library's own RCCL communicator, torch.distributed callbacks if it cannot be "
                         "built; rccl = no fallback; torch = callbacks only")
    ap.add_argument("--traffic", choices=["live", "file", "off"], default="live",
                    help="roofline.traffic (HBM bytes of one pack launch from the PMC counters): live = two child runs "
                         "of this script under rocprofv3 (--pmc FETCH_SIZE, --pmc WRITE_SIZE, separate passes; N=1 "
                         "only), falling back to file = profiles/pack_traffic.json when that is not possible")
    ap.add_argument("--declared-partition", action="store_true",
                    help="pgsd_set_partition instead of the per-frame allgather: every rank's row count is declared once "
                         "(weak scaling: the same on all ranks), frames then cost NO collective (collectives_per_frame 0). "
                         "The default keeps the north_star's shape: one allgather of chunk sizes per frame")
    ap.add_argument("--io", choices=["posix", "mpiio"], default="posix",
                    help="file back end: posix = pwrite at the offsets the reference passes to MPI_File_write_at (default); "
                         "mpiio = those very calls (PGSD_IO=mpiio, plugin libpgsd_amd_mpiio.so); MPI is initialised "
                         "in-process (MPI_Init_thread through ctypes, every rank a singleton: the file is opened on "
                         "MPI_COMM_SELF); PGSD_LIBMPI names the libmpi to load")
    ap.add_argument("--no-legs", action="store_true",
                    help="N=1, --schema pvi: skip the `legs` (BASELINE configs 2, 4, 5 and the tag-order gather measured "
                         "after the headline; bench_legs.py)")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="N>1 rehearsal on a one-GPU box: every rank uses cuda:0 and a gloo group")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be at least 1")

    # --gpus N is the number of ranks of the run, always.  Ranks started by a launcher (WORLD_SIZE set) must
    # be N of them; without a launcher and N > 1 this process becomes the launcher -- before torch or HIP are
    # touched, so no GPU-initialised process ever starts or replaces another.
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
        world, rank, local_rank = 1, 0, 0
    else:
        world = int(os.environ["WORLD_SIZE"])
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world != args.gpus:
            raise SystemExit("bench.py: launched with WORLD_SIZE=%d but --gpus %d: the two must agree "
                             "(the JSON's n_gpus is the number of ranks that ran)" % (world, args.gpus))

    # RCCL's intra-node transport shares device buffers between the rank processes by IPC handle, and the hosts
    # of this pool only support dmabuf handles: with the legacy mode hipIpcGetMemHandle fails with "invalid
    # argument" and ncclCommInitRank with it.  The image exports the setting already; a rank gets it here too when a
    # launcher scrubbed the environment -- the same way whether torch.distributed.run or this script started the
    # ranks, and before anything loads HIP.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL's version banner at
    # ncclCommInitRank, for one): from here on file descriptor 1 IS stderr, and the line goes out through a
    # private duplicate of the real stdout at the very end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    mpi = None
    if args.io == "mpiio":
        # the caller initialises MPI, as with the reference; MPI_THREAD_SERIALIZED: the pipeline's writer thread calls
        # into MPI-IO too (the library serialises the calls)
        os.environ["PGSD_IO"] = "mpiio"
        mpi = ctypes.CDLL(os.environ.get("PGSD_LIBMPI", "/opt/conda/lib/libmpi.so.12"), mode=ctypes.RTLD_GLOBAL)
        provided = ctypes.c_int(-1)
        if mpi.MPI_Init_thread(None, None, 2, ctypes.byref(provided)) != 0 or provided.value < 2:
            raise SystemExit("bench.py --io mpiio: MPI_Init_thread(MPI_THREAD_SERIALIZED) failed (provided %d)" % provided.value)

    import numpy as np
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.rehearse_shared_gpu:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit("bench.py: --gpus %d but this node shows %d GPU(s); one rank per GPU is the contract "
                         "(--rehearse-shared-gpu lets the ranks share cuda:0 over a gloo group, for rehearsals only)"
                         % (world, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    red_dev = "cuda"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.rehearse_shared_gpu:
            dist.init_process_group(backend="gloo")
            red_dev = "cpu"
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import pgsd.fl as fl
    import pgsd.dist as pdist
    comm_backend = "self"
    if world > 1:
        if args.comm == "torch":
            comm_backend = pdist.init_from_torch(device=local_rank, prefer_rccl=False)
        else:
            try:
                # native RCCL communicator (ncclAllGather over xGMI issued from the C++ library)
                comm_backend = pdist.init_from_torch(device=local_rank)
            except RuntimeError as e:  # keep the run alive on a host-callback communicator
                if args.comm == "rccl":
                    raise
                print("bench.py: RCCL communicator unavailable (%s); using torch.distributed callbacks" % e,
                      file=sys.stderr)
                comm_backend = pdist.init_from_torch(device=local_rank, prefer_rccl=False)

    N = args.particles
    # HBM-resident particle arrays of this rank and the chunks they feed (bench_legs.make_fields: HOOMD's own
    # device layout, Scalar4 pos = x, y, z, __int_as_scalar(type); Scalar4 vel = vx, vy, vz, mass; the full schemas
    # add HOOMD-SPH's Scalar4 / int4 / scalar arrays)
    import bench_legs
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    fields, payload_bpp, algo_bpp, layout, _arrays = bench_legs.make_fields(args.schema, N, g, torch, np, fl,
                                                                            separate_id=args.separate_id)

    # The run appends world x (warmup + steps [+ 3 stall-test frames]) frames to ONE file: 2.24 GB per frame at eight
    # ranks.  A target without that much room would end the run in ENOSPC half-way (a tmpfs is also memory of the
    # job): rank 0 looks first.  When --dir is too small the run STOPS with the figures -- a 1 -> 8 curve whose points
    # landed on different file systems (tmpfs 7 GB/s, page cache 11 GB/s) would not be a curve -- unless
    # --dir-fallback allows the first of /tmp and the working directory that has the room.  Either way the JSON names
    # the directory and its file system type (config.target_dir / config.target_fstype).
    extra = 4 + 5 * 4 if world == 1 and not args.no_stall_test else 1    # 3 stall-test frames, 5 drains of 4 frames beside a kernel queue
    need = world * (args.warmup + args.steps + extra) * N * payload_bpp + (256 << 20)
    choice = [None]
    if rank == 0:
        import shutil
        for cand in ((args.dir, "/tmp", os.getcwd()) if args.dir_fallback else (args.dir,)):
            try:
                if shutil.disk_usage(cand).free >= need * 1.05 and os.access(cand, os.W_OK):
                    choice = [cand]
                    break
            except OSError:
                continue
    if world > 1:
        dist.broadcast_object_list(choice, src=0)
    if choice[0] is None:
        raise SystemExit("bench.py: the run writes %.1f GB (%d ranks x %d frames x %.2f GB) and %s has not that much room%s: "
                         "fewer --steps, --dir <larger target>, or --dir-fallback to let the run move to /tmp or the "
                         "working directory"
                         % (need / 1e9, world, args.warmup + args.steps + extra, N * payload_bpp / 1e9, args.dir,
                            " (nor /tmp, nor the working directory)" if args.dir_fallback else ""))
    if choice[0] != args.dir:
        print("bench.py: %s has less than the %.1f GB the run writes: writing to %s instead (--dir-fallback)"
              % (args.dir, need / 1e9, choice[0]), file=sys.stderr)
        args.dir = choice[0]
    target_fstype = fstype_of(args.dir)
    path = os.path.join(args.dir, "pgsd_bench_%s.gsd" % os.environ.get("MASTER_PORT", str(os.getpid())))
    f = fl.open(path, "w", application="pgsd_amd bench", schema="hoomd", schema_version=[1, 4])
    f.configure_device(device=local_rank, slab_bytes=args.slab_mib << 20, n_slabs=args.slabs,
                       n_writers=args.writers, profile=True)

    # ONE collective per frame (pgsd_set_frame_exchange): the replicated step chunk and the fused device
    # chunks are queued -- the pack kernel is launched at once, it needs no file offset -- and end_frame's
    # single allgather (one ncclAllGather over xGMI on the RCCL back end, in flight while the kernel runs)
    # carries every rank's chunk sizes, from which each rank derives N_global, its first row and all file
    # offsets (offset="auto": the MPI_Allgather of benchmark-write.cc:39-45 is inside that exchange).
    if args.declared_partition:
        f.set_partition([N] * world)
    else:
        f.frame_exchange = True

    def step(i):
        f.write_chunk("configuration/step", np.array([i], dtype=np.uint64), write_all=False)
        f.write_chunks(fields, offset="auto")
        f.end_frame()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    f.device_stats(reset=True)
    f.exchange_stats(reset=True)
    fence()
    coll0 = f.collective_count
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    stats = f.device_stats()
    xstats = f.exchange_stats()
    collectives_per_frame = (f.collective_count - coll0) / max(args.steps, 1)

    # What a simulation is blocked for per snapshot when it seals frames asynchronously: issue the
    # frame, wait for the pack kernels only (the arrays may then change), let copy + write run on.
    stall_ms, overlap = None, None
    if world == 1 and not args.no_stall_test:
        stalls = []
        for i in range(3):
            f.frame_sync()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            f.write_chunk("configuration/step", np.array([10 ** 6 + i], dtype=np.uint64), write_all=False)
            f.write_chunks(fields, offset="auto")
            f.end_frame(wait=False)
            f.wait_packed()
            stalls.append((time.perf_counter() - t1) * 1e3)
        f.frame_sync()
        stall_ms = round(min(stalls), 3)
        # ... and what the draining snapshot costs the kernels that run meanwhile (the copies are shader blits,
        # __amd_rocclr_copyBuffer: they share the CUs with the simulation)
        overlap = overlap_slowdown(f, fields, np, torch)
    f.close()

    exch_mean = xstats["total_us"] / max(xstats["count"], 1)
    exchange_us = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        pk = torch.tensor([stats["pack_ms"] / max(stats["pack_launches"], 1)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(pk, op=dist.ReduceOp.MAX)
        pack_ms = float(pk.item())
        # the frame's one allgather as each rank's calling thread saw it (pgsd_get_exchange_stats): the mean
        # of the slowest rank, the fastest rank's mean (the last to arrive waits least: closest to the
        # transport's own latency) and the worst single exchange anywhere
        xs = torch.tensor([exch_mean, -exch_mean, xstats["max_us"], -xstats["min_us"]], dtype=torch.float64,
                          device=red_dev)
        dist.all_reduce(xs, op=dist.ReduceOp.MAX)
        exchange_us = {"mean": round(float(xs[0]), 1), "mean_fastest_rank": round(-float(xs[1]), 1),
                       "max": round(float(xs[2]), 1), "min": round(-float(xs[3]), 1),
                       "per_rank_count": xstats["count"], "bytes_per_rank": 512,
                       "what": "wall time of the allgather call on the rank's thread = transport + the wait for the "
                               "slowest rank to arrive (ranks writing one shared file arrive one write apart); `min` "
                               "(the quickest single exchange on any rank: a last arriver) is the closest to the "
                               "transport's own latency"}
    else:
        pack_ms = stats["pack_ms"] / max(stats["pack_launches"], 1)
    if rank == 0:
        try:
            os.unlink(path)
        except OSError:
            pass

    if world > 1:
        dist.barrier()
        pdist.finalize()          # tears down the library's RCCL communicator on every rank
        dist.destroy_process_group()
    if rank != 0:
        if mpi is not None:
            mpi.MPI_Finalize()
        return

    # N = 1 has no exchange (a single rank skips it).  What ONE 512-byte exchange costs on the RCCL back end is
    # probed on a one-rank RCCL communicator instead: pinned -> device copy, ncclAllGather, device -> pinned copy,
    # stream synchronize (pgsd_comm_rccl.cpp) -- the floor under the N > 1 figure, no xGMI hop in it.
    exchange_probe = None
    if world == 1 and not args.no_exchange_probe:
        exchange_probe = probe_rccl_exchange(local_rank)

    # HBM bytes of one pack launch from the PMC counters (MI355X_MICROARCH.md, HBM section: FETCH_SIZE and
    # WRITE_SIZE in separate passes, KiB units, FETCH_SIZE doubled on gfx950 for wide coalesced reads)
    traffic, traffic_source, traffic_detail = None, None, None
    if args.traffic == "live" and world == 1:
        traffic, traffic_source, traffic_detail = measure_traffic_live(args)
    if traffic is None and args.traffic != "off":
        traffic, traffic_source = traffic_from_file(N, args)

    total_bytes = world * args.steps * N * payload_bpp
    value = total_bytes / dt / 1e9
    achieved = algo_bpp * N / (pack_ms * 1e-3) / 1e9 if pack_ms > 0 else 0.0
    out = {
        "metric": "snapshot pack+write GB/s at 10M particles/GPU",
        "value": round(value, 3),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {"workload": "%d particles/GPU, %s packed from %s, "
                               "%s allgather of chunk sizes (one per frame), one shared GSD file on %s"
                               % (N, {"pvi": "position+velocity+typeid",
                                      "sph": "the 14 per-particle chunks of the PGSD-SPH schema (112 B/particle)",
                                      "union": "the 19 per-particle chunks of the SPH schema + upstream HOOMD attributes "
                                               "(164 B/particle)"}[args.schema], layout, comm_backend, args.dir),
                   "schema": args.schema, "io": args.io, "target_dir": args.dir, "target_fstype": target_fstype,
                   "particles_per_gpu": N, "payload_bytes_per_frame_per_gpu": N * payload_bpp,
                   "parallelism": "particle-partition x%d" % world},
        "comm_backend": comm_backend,
        "placement": "declared partition (pgsd_set_partition)" if args.declared_partition else "one allgather of chunk sizes per frame",
        "collectives_per_frame": collectives_per_frame,
        "exchange_us": exchange_us,
        "exchange_probe": exchange_probe,
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                     "traffic_detail": traffic_detail,
                     "kernel": PACK_KERNEL, "avg_ms": round(pack_ms, 5),
                     "algorithmic_bytes_per_launch": algo_bpp * N},
        "pack_aggregate": {"algorithmic_GBps": round(world * achieved, 1), "launches_per_rank": int(stats["pack_launches"]),
                           "note": "n_gpus x the slowest rank's pack kernel rate (independent per-GPU rates, nothing shared); "
                                   "`value` is the shared-file rate"},
        "snapshot_stall_ms": stall_ms,
        "snapshot_overlap_slowdown_pct": overlap["slowdown_pct"] if overlap else None,
        "snapshot_overlap": overlap,
        "pipeline": {"d2h_GBps": round(stats["d2h_bytes"] / max(stats["d2h_ms"], 1e-9) / 1e6, 2),
                     "write_GBps_per_writer": round(stats["written_bytes"] / max(stats["write_ms"], 1e-9) / 1e6, 2)},
    }
    if not args.no_cpu_baseline:
        # the reference's CPU path on this host's cores: one MPI rank per GPU of the run (the other ranks of this
        # run have left by now), each bound to a core; a bounded sample (18 GB of file at most, 10-30 s)
        frames = max(4, 64 // world) if args.schema == "pvi" else max(3, 16 // world)
        out["cpu_baseline"] = cpu_baseline_reference(N, frames, args.dir, ranks=world, schema=args.schema)
        if out["cpu_baseline"] is None and args.schema == "pvi":
            out["cpu_baseline"] = cpu_baseline(N, 8, args.dir)
    if world == 1 and args.schema == "pvi" and not args.no_legs:
        # the other BASELINE configurations, into the same line (bench_legs.py): config 2 by SURVEY 8(d)'s protocol,
        # config 4 (sph, union), config 5's read, the tag-order gather -- kernel time, algorithmic bytes, fraction of
        # the HBM peak and PMC traffic each
        import shutil
        room = shutil.disk_usage(args.dir).free
        if room < 8 * N * 164 + (256 << 20):
            out["legs"], out["legs_note"] = [], "skipped: %s has %.1f GB free, the legs write up to %.1f GB" % (
                args.dir, room / 1e9, 8 * N * 164 / 1e9)
        else:
            t_legs = time.perf_counter()
            out["legs"], out["legs_traffic_source"] = bench_legs.run_legs(N, args.dir, target_fstype,
                                                                          traffic=(args.traffic == "live" and not under_a_profiler()))
            out["legs_wall_s"] = round(time.perf_counter() - t_legs, 1)
            # ... and their essentials as flat scalars inside `roofline` (what the driver's record keeps of the line)
            out["roofline"].update(bench_legs.flat_summary(out["legs"]))
    if mpi is not None:
        mpi.MPI_Finalize()
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)


if __name__ == "__main__":
    main()
