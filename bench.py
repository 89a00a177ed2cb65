#!/usr/bin/env python3
"""bench.py -- snapshot pack+write throughput of the MI355X-native PGSD writer.

One "step" = one frame of the hot path on one batch of synthetic particle data that is
already resident in HBM: RCCL allgather of the per-rank row counts (file offsets), fused
HIP pack of position / velocity / typeid from HOOMD-style float4 arrays, hipMemcpyAsync to
pinned slabs, pwrite into one shared GSD file on tmpfs, index commit (pgsd_end_frame).

    python bench.py --gpus N --steps K --warmup W

For N > 1 the driver launches it with torch.distributed.run (one rank per GPU); ranks share
ONE output file, each writing its own partition (weak scaling: 10 M particles per GPU).
Rank 0 prints one JSON line (contract in the task description) with `roofline` for the pack
kernel and `cpu_baseline` for the CPU restatement of the reference path (oracle/, N=1 only).
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


def cpu_baseline_reference(n_particles, frames, out_dir):
    """Time the reference ITSELF (oracle/_ref/ref_bench = the reference's pgsd.c compiled in the
    build container + tests/drivers/ref_bench.c) under MPICH on this host: one rank = one core,
    plus a 4-rank run of the same total workload for orientation.  None when it cannot run."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_bench")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(exe) and os.path.exists(mpiexec)):
        return None
    path = os.path.join(out_dir, "pgsd_bench_ref_%d.gsd" % os.getpid())
    res = {}
    try:
        for ranks in (1, 4):
            out = subprocess.run([mpiexec, "-n", str(ranks), exe, str(n_particles), str(frames), path],
                                 capture_output=True, timeout=300, check=True).stdout.decode()
            res[ranks] = json.loads(out.strip().splitlines()[-1])
    except Exception as e:  # missing MPI runtime, time-out, ...
        print("bench.py: reference baseline unavailable (%s), using the oracle port" % e, file=sys.stderr)
        return None
    finally:
        if os.path.exists(path):
            os.unlink(path)
    return {"value": round(res[1]["GBps"], 4), "unit": "GB/s", "cores": 1, "kind": "reference",
            "sample": "%d frames x %d particles (pos+vel+typeid): C pack loop out of float4 arrays + the "
                      "reference's pgsd_write_chunk/pgsd_end_frame (MPICH 3.3.2 MPI-IO), 1 rank on 1 core, file "
                      "on %s; the same workload on 4 ranks: %.3f GB/s" % (frames, n_particles, out_dir, res[4]["GBps"])}


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
    return {"value": round(gbs, 4), "unit": "GB/s", "cores": 1, "kind": "port",
            "sample": "%d frames x %d particles (pos+vel+typeid), float4 -> chunk pack in C + "
                      "reference write sequence, 1 thread, file on %s" % (frames, n_particles, out_dir)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--particles", type=int, default=10_000_000, help="particles per GPU")
    ap.add_argument("--dir", default=os.environ.get("PGSD_BENCH_DIR", "/dev/shm"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="N>1 rehearsal on a one-GPU box: every rank uses cuda:0 and a gloo group")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.rehearse_shared_gpu:
        local_rank = 0
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
    # HOOMD's own device layout (ParticleData: Scalar4 pos = x, y, z, __int_as_scalar(type);
    # Scalar4 vel = vx, vy, vz, mass).  --separate-id keeps the id in its own uint32 array as
    # SURVEY.md 8(d) sketches; both layouts have 56 algorithmic bytes per particle.
    g = torch.Generator(device="cuda").manual_seed(1234 + rank)
    pos = (torch.rand((N, 4), generator=g, device="cuda") - 0.5) * 100.0
    vel = torch.randn((N, 4), generator=g, device="cuda")
    tid = torch.randperm(N, generator=g, device="cuda").to(torch.int32)
    if args.separate_id:
        id_field = fl.DeviceField.from_tensor(tid, out_dtype=np.uint32)
        layout = "float4 pos, float4 vel, separate uint32 id array"
    else:
        pos[:, 3] = tid.view(torch.float32)
        id_field = fl.DeviceField.from_tensor(pos, columns=(3, 4), out_dtype=np.uint32, bitcast=True)
        layout = "HOOMD Scalar4 arrays: pos=(x,y,z,typeid bits), vel=(vx,vy,vz,mass)"
    fields = [("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
              ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3))),
              ("particles/typeid", id_field)]

    payload_bpp, algo_bpp = PAYLOAD_BYTES_PER_PARTICLE, ALGO_BYTES_PER_PARTICLE
    if args.schema in ("sph", "union"):
        # hoomd.py:167-184: typeid, mass, body, position, velocity, slength, density, pressure, energy,
        # auxiliary1-4, image -- from HOOMD-SPH-style device arrays (Scalar4 / int3-as-int4 / int)
        dpe = torch.rand((N, 4), generator=g, device="cuda")                 # density, pressure, energy, slength
        aux = [torch.randn((N, 4), generator=g, device="cuda") for _ in range(4)]
        img = torch.randint(-2, 3, (N, 4), generator=g, device="cuda", dtype=torch.int32)
        body = torch.full((N,), -1, device="cuda", dtype=torch.int32)
        fields = [("particles/typeid", id_field),
                  ("particles/mass", fl.DeviceField.from_tensor(vel, columns=(3, 4))),
                  ("particles/body", fl.DeviceField.from_tensor(body)),
                  ("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
                  ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3))),
                  ("particles/slength", fl.DeviceField.from_tensor(dpe, columns=(3, 4))),
                  ("particles/density", fl.DeviceField.from_tensor(dpe, columns=(0, 1))),
                  ("particles/pressure", fl.DeviceField.from_tensor(dpe, columns=(1, 2))),
                  ("particles/energy", fl.DeviceField.from_tensor(dpe, columns=(2, 3)))]
        fields += [("particles/auxiliary%d" % (k + 1), fl.DeviceField.from_tensor(a, columns=(0, 3)))
                   for k, a in enumerate(aux)]
        fields.append(("particles/image", fl.DeviceField.from_tensor(img, columns=(0, 3))))
        payload_bpp, algo_bpp = 112, 224
        layout = "HOOMD-SPH device arrays (Scalar4 pos/vel/dpe/aux1-4, int4 image, int body), full SPH schema"
    if args.schema == "union":
        # BASELINE config 4: the SPH set plus the upstream HOOMD particle attributes (pgsd.tex:508-521):
        # charge, diameter (Scalar arrays), moment_inertia (Scalar3), orientation, angmom (Scalar4)
        charge = torch.randn((N,), generator=g, device="cuda")
        diameter = torch.rand((N,), generator=g, device="cuda")
        inertia = torch.rand((N, 3), generator=g, device="cuda")
        orient = torch.randn((N, 4), generator=g, device="cuda")
        angmom = torch.randn((N, 4), generator=g, device="cuda")
        fields += [("particles/charge", fl.DeviceField.from_tensor(charge)),
                   ("particles/diameter", fl.DeviceField.from_tensor(diameter)),
                   ("particles/moment_inertia", fl.DeviceField.from_tensor(inertia)),
                   ("particles/orientation", fl.DeviceField.from_tensor(orient)),
                   ("particles/angmom", fl.DeviceField.from_tensor(angmom))]
        payload_bpp, algo_bpp = 164, 328
        layout = "HOOMD-SPH device arrays + upstream HOOMD attributes (charge, diameter, moment_inertia, orientation, angmom)"

    path = os.path.join(args.dir, "pgsd_bench_%s.gsd" % os.environ.get("MASTER_PORT", str(os.getpid())))
    f = fl.open(path, "w", application="pgsd_amd bench", schema="hoomd", schema_version=[1, 4])
    f.configure_device(device=local_rank, slab_bytes=args.slab_mib << 20, n_slabs=args.slabs,
                       n_writers=args.writers, profile=True)

    def step(i):
        # per-rank file offsets: allgather of the local row counts (RCCL over xGMI for N > 1)
        counts, row0, n_global = pdist.partition_rows(N)
        f.write_chunk("configuration/step", np.array([i], dtype=np.uint64), write_all=False)
        f.write_chunks(fields, offset=counts, rank=rank)
        f.end_frame()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    f.device_stats(reset=True)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    fence()
    dt = time.perf_counter() - t0
    stats = f.device_stats()

    # What a simulation is blocked for per snapshot when it seals frames asynchronously: issue the
    # frame, wait for the pack kernels only (the arrays may then change), let copy + write run on.
    stall_ms = None
    if world == 1:
        stalls = []
        for i in range(3):
            f.frame_sync()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            counts, row0, n_global = pdist.partition_rows(N)
            f.write_chunk("configuration/step", np.array([10 ** 6 + i], dtype=np.uint64), write_all=False)
            f.write_chunks(fields, offset=counts, rank=rank)
            f.end_frame(wait=False)
            f.wait_packed()
            stalls.append((time.perf_counter() - t1) * 1e3)
        f.frame_sync()
        stall_ms = round(min(stalls), 3)
    f.close()

    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        pk = torch.tensor([stats["pack_ms"] / max(stats["pack_launches"], 1)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(pk, op=dist.ReduceOp.MAX)
        pack_ms = float(pk.item())
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
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # HBM bytes per launch from the committed PMC passes (rocprofv3 cannot run inside this
    # process); only reported when it was measured for this particle count
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "pack_traffic.json")) as tf:
            tj = json.load(tf)
        if tj.get("particles") == N and args.schema == "pvi" and not args.separate_id:
            traffic = tj["hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass

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
        "config": {"workload": "%d particles/GPU, position+velocity+typeid packed from %s, "
                               "%s allgather of row counts, one shared GSD file on %s"
                               % (N, layout, comm_backend, args.dir),
                   "particles_per_gpu": N, "payload_bytes_per_frame_per_gpu": N * payload_bpp,
                   "parallelism": "particle-partition x%d" % world},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                     "kernel": "pack_tiles_kernel", "avg_ms": round(pack_ms, 5),
                     "algorithmic_bytes_per_launch": algo_bpp * N},
        "pack_aggregate": {"algorithmic_GBps": round(world * achieved, 1), "launches_per_rank": int(stats["pack_launches"]),
                           "note": "sum over ranks of the pack kernel rate (independent kernels, one per GPU)"},
        "snapshot_stall_ms": stall_ms,
        "pipeline": {"d2h_GBps": round(stats["d2h_bytes"] / max(stats["d2h_ms"], 1e-9) / 1e6, 2),
                     "write_GBps_per_writer": round(stats["written_bytes"] / max(stats["write_ms"], 1e-9) / 1e6, 2)},
    }
    if world == 1 and not args.no_cpu_baseline and args.schema == "pvi":
        out["cpu_baseline"] = cpu_baseline_reference(N, 64, args.dir) or cpu_baseline(N, 8, args.dir)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
