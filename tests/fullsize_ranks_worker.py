"""BASELINE configs 3 and 4 at their own size AND rank count on one GPU: P ranks as P threads of this process
(the boxes let six PROCESSES use the card), each with a communicator, a handle and a device pipeline of its own,
every rank packing its n rows of shared Scalar4 arrays on cuda:0 and all of them writing ONE file.

    python fullsize_ranks_worker.py <shm|rccl> <P> <rows per rank> <config3|config4> <directory>

config3: position + velocity + typeid out of two float4 arrays (the id's bits in position.w), configuration/step
         replicated, TWO frames -- at 8 x 10 M rows 2.24 GB per frame, the second frame crosses 4 GiB of file offset
         with eight writers (reference: every rank's rows at file_size + offset * sz, pgsd.c:2225-2249; the caller's
         partition benchmark-write.cc:33-45).
config4: the 112 B/particle PGSD-SPH set (hoomd.py:167-184; 14 per-particle chunks out of eight Scalar4 / int4
         arrays and one scalar array), ONE frame -- 8.96 GB at 8 x 10 M rows.

The process then lets the CPU oracle write the P-rank file of host copies of the same values and prints

    RESULT size_mine=<bytes> size_ref=<bytes> sha_mine=<hex> sha_ref=<hex> collectives=<per frame, per rank> write_s=<s>

(sha256 over 64 MiB pieces; the caller asserts).  Exit status 0 only when sizes and digests are equal."""
import hashlib
import os
import sys
import threading
import time
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import pgsd.dist as pdist
import pgsd.fl as fl
import scenario as S

# chunk -> (pgsd type id, columns), the order of hoomd.py:167-184
SPH = [("typeid", 3, 1), ("mass", 9, 1), ("body", 7, 1), ("position", 9, 3), ("velocity", 9, 3),
       ("slength", 9, 1), ("density", 9, 1), ("pressure", 9, 1), ("energy", 9, 1),
       ("auxiliary1", 9, 3), ("auxiliary2", 9, 3), ("auxiliary3", 9, 3), ("auxiliary4", 9, 3), ("image", 7, 3)]


def make_frames(what, N):
    """[(device arrays of the whole system, {chunk: host (N, M) array})] per frame; values from torch's generator."""
    g = torch.Generator(device="cuda").manual_seed(1234)
    frames = []
    for frame in range(2 if what == "config3" else 1):
        pos4 = (torch.rand((N, 4), generator=g, device="cuda") - 0.5) * 100.0
        pos4[:, 3] = torch.randint(0, 7, (N,), generator=g, device="cuda", dtype=torch.int32).view(torch.float32)
        vel4 = torch.randn((N, 4), generator=g, device="cuda")
        dev = {"pos4": pos4, "vel4": vel4}
        host = {"typeid": pos4[:, 3:4].contiguous().view(torch.int32).cpu().numpy().view(np.uint32),
                "position": pos4[:, :3].cpu().numpy(), "velocity": vel4[:, :3].cpu().numpy()}
        if what == "config4":
            dev["dpe4"] = torch.rand((N, 4), generator=g, device="cuda")
            dev["aux"] = [torch.randn((N, 4), generator=g, device="cuda") for _ in range(4)]
            dev["img4"] = torch.randint(-2, 3, (N, 4), generator=g, device="cuda", dtype=torch.int32)
            dev["body"] = torch.randint(-1, 50, (N,), generator=g, device="cuda", dtype=torch.int32)
            host.update(mass=vel4[:, 3:4].cpu().numpy(), body=dev["body"].cpu().numpy().reshape(-1, 1),
                        slength=dev["dpe4"][:, 3:4].cpu().numpy(), density=dev["dpe4"][:, 0:1].cpu().numpy(),
                        pressure=dev["dpe4"][:, 1:2].cpu().numpy(), energy=dev["dpe4"][:, 2:3].cpu().numpy(),
                        image=dev["img4"][:, :3].cpu().numpy())
            for k in range(4):
                host["auxiliary%d" % (k + 1)] = dev["aux"][k][:, :3].cpu().numpy()
        frames.append((dev, host))
    torch.cuda.synchronize()
    return frames


def rank_fields(what, dev, lo, hi):
    """This rank's rows [lo, hi) of the shared arrays as the chunk list of one fused launch."""
    D = fl.DeviceField.from_tensor
    pos4, vel4 = dev["pos4"][lo:hi], dev["vel4"][lo:hi]
    if what == "config3":
        return [("particles/position", D(pos4, columns=(0, 3))), ("particles/velocity", D(vel4, columns=(0, 3))),
                ("particles/typeid", D(pos4, columns=(3, 4), out_dtype=np.uint32, bitcast=True))]
    dpe4, img4 = dev["dpe4"][lo:hi], dev["img4"][lo:hi]
    fields = {"typeid": D(pos4, columns=(3, 4), out_dtype=np.uint32, bitcast=True), "mass": D(vel4, columns=(3, 4)),
              "body": D(dev["body"][lo:hi]), "position": D(pos4, columns=(0, 3)), "velocity": D(vel4, columns=(0, 3)),
              "slength": D(dpe4, columns=(3, 4)), "density": D(dpe4, columns=(0, 1)), "pressure": D(dpe4, columns=(1, 2)),
              "energy": D(dpe4, columns=(2, 3)), "image": D(img4, columns=(0, 3))}
    for k in range(4):
        fields["auxiliary%d" % (k + 1)] = D(dev["aux"][k][lo:hi], columns=(0, 3))
    return [("particles/" + name, fields[name]) for name, _, _ in SPH]


def rank_main(rank, P, n, kind, path, what, frames, shm_name, uid, errors, stats):
    try:
        torch.cuda.set_device(0)
        comm = pdist.create_shm(shm_name, rank, P) if kind == "shm" else pdist.create_rccl(uid, rank, P, 0)
        f = fl.open(path, "w", application="app", schema="hoomd", schema_version=[1, 4], comm=comm)
        assert f.rank == rank and f.nprocs == P
        f.frame_exchange = True                  # ONE allgather per frame: chunk sizes, the partition derived from it
        c0 = f.collective_count
        for i, (dev, _) in enumerate(frames):
            f.write_chunk("configuration/step", np.array([i], dtype=np.uint64), write_all=False)
            f.write_chunks(rank_fields(what, dev, rank * n, (rank + 1) * n), offset="auto")
            f.end_frame()
        stats[rank] = (f.collective_count - c0) / float(len(frames))
        f.close()
        pdist.release(comm)
    except Exception:  # pragma: no cover
        import traceback
        errors.append((rank, traceback.format_exc()))


def oracle_file(path, P, n, what, frames):
    from test_gpu_file import _oracle_frames
    spec = SPH if what == "config4" else [("position", 9, 3), ("velocity", 9, 3), ("typeid", 3, 1)]
    out = []
    for i, (_, host) in enumerate(frames):
        chunks = [("configuration/step", 4, 1, False, [np.array([[i]], dtype=np.uint64)] * P)]
        for name, t, M in spec:
            a = host[name]
            assert a.flags.c_contiguous and a.shape == (P * n, M)
            chunks.append(("particles/" + name, t, M, True, [a[r * n:(r + 1) * n] for r in range(P)]))
        out.append(chunks)
    _oracle_frames(path, P, out)


def digest(path, out, key):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for block in iter(lambda: fh.read(64 << 20), b""):
            h.update(block)
    out[key] = h.hexdigest()


def main():
    kind, P, n, what, d = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    mine = os.path.join(d, "pgsd_full8_mine_%d.gsd" % os.getpid())
    ref = os.path.join(d, "pgsd_full8_ref_%d.gsd" % os.getpid())
    try:
        torch.cuda.set_device(0)
        frames = make_frames(what, P * n)
        shm_name = "pgsdfull_%s" % uuid.uuid4().hex[:10]
        uid = pdist.rccl_unique_id() if kind == "rccl" else None
        errors, stats = [], [None] * P
        threads = [threading.Thread(target=rank_main, args=(r, P, n, kind, mine, what, frames, shm_name, uid, errors, stats))
                   for r in range(P)]
        t0 = time.perf_counter()
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=300)
        write_s = time.perf_counter() - t0
        if errors or any(t.is_alive() for t in threads):
            for rank, tb in errors:
                sys.stderr.write("rank %d:\n%s\n" % (rank, tb))
            sys.stderr.flush()
            os._exit(1)             # threads stuck in a collective cannot be joined
        for dev, _ in frames:
            dev.clear()
        torch.cuda.empty_cache()
        oracle_file(ref, P, n, what, frames)
        sha = {}
        hashers = [threading.Thread(target=digest, args=(p, sha, k)) for p, k in ((mine, "mine"), (ref, "ref"))]
        for t in hashers:
            t.start()
        for t in hashers:
            t.join()
        sizes = os.path.getsize(mine), os.path.getsize(ref)
        print("RESULT size_mine=%d size_ref=%d sha_mine=%s sha_ref=%s collectives=%s write_s=%.3f"
              % (sizes[0], sizes[1], sha["mine"], sha["ref"], ",".join("%g" % s for s in stats), write_s))
        sys.stdout.flush()
        ok = sizes[0] == sizes[1] and sha["mine"] == sha["ref"]
    finally:
        for p in (mine, ref):
            if os.path.exists(p):
                os.unlink(p)
    os._exit(0 if ok else 1)


if __name__ == "__main__":
    main()
