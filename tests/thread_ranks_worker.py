"""P ranks as P THREADS of one process, every rank with a communicator and a file object of its own
(`pgsd.dist.create_shm` / `create_rccl` + `pgsd.fl.open(..., comm=)`), all packing on cuda:0.

Run as a script by tests/test_gpu_eight_ranks.py (the GPU boxes let at most six PROCESSES use the card at once,
so eight ranks can only meet on it as threads):

    python thread_ranks_worker.py <shm|rccl> <P> <out.gsd> <posvelid|uneven> [host]

(`host`: the rows come from numpy arrays through pgsd_write_chunk -- the CPU suite's check of the per-handle
communicators; no GPU involved.)

`posvelid`: the chunk sequence and closed-form values of tests/golden/scenarios/posvelid.scn (config 3's shape:
position + velocity + typeid from float4 arrays in HBM, step / N replicated), 1000 particles split evenly -- the
file must equal the reference-written golden posvelid.p<P>.gsd.  `uneven`: a partition with an empty and a
one-row rank, three frames, checked against the oracle by the caller.  The frame exchange is batched: ONE
collective per frame."""
import os
import sys
import threading
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import pgsd.dist as pdist
import pgsd.fl as fl
import scenario as S

UNEVEN = [5000, 0, 1, 77, 4096, 333, 2, 1500]


def embed4(rows3, w):
    out = np.zeros((rows3.shape[0], 4), dtype=rows3.dtype)
    out[:, :3] = rows3
    out[:, 3] = w
    return out


def rank_main(rank, P, kind, path, what, shm_name, uid, errors, stats, host):
    try:
        if not host:
            torch.cuda.set_device(0)
        comm = pdist.create_shm(shm_name, rank, P) if kind == "shm" else pdist.create_rccl(uid, rank, P, 0)
        f = fl.open(path, "w", application="pgsd_amd_test" if what == "posvelid" else "app", schema="hoomd",
                    schema_version=[1, 4], comm=comm)
        assert f.rank == rank and f.nprocs == P
        f.frame_exchange = True
        if what == "posvelid":
            counts = S.dist_counts("even:1000", P)
            seeds = (1234, 1235, 1236)
        else:
            counts = UNEVEN[:P]
            seeds = (40, 41, 42)
        n, row0 = counts[rank], sum(counts[:rank])
        c0 = f.collective_count
        for frame, seed in enumerate(seeds):
            pos, vel, tid = S.gen_data(9, seed, row0, n, 3), S.gen_data(9, seed, row0, n, 3), S.gen_data(3, seed, row0, n, 1)
            f.write_chunk("configuration/step", S.gen_data(4, seed, 0, 1, 1), write_all=False)
            if frame == 0 and what == "posvelid":
                f.write_chunk("particles/N", S.gen_data(3, seed, 0, 1, 1), write_all=False)
            if host:
                f.write_chunk("particles/position", pos, offset="auto")
                f.write_chunk("particles/velocity", vel, offset="auto")
                if frame < 2 or what != "posvelid":
                    f.write_chunk("particles/typeid", tid, offset="auto")
                f.end_frame()
                continue
            dpos = torch.from_numpy(embed4(pos, tid[:, 0].view(np.float32))).cuda()
            dvel = torch.from_numpy(embed4(vel, np.float32(1.0))).cuda()
            fields = [("particles/position", fl.DeviceField.from_tensor(dpos, columns=(0, 3))),
                      ("particles/velocity", fl.DeviceField.from_tensor(dvel, columns=(0, 3)))]
            if frame < 2 or what != "posvelid":
                fields.append(("particles/typeid", fl.DeviceField.from_tensor(dpos, columns=(3, 4), out_dtype=np.uint32,
                                                                              bitcast=True)))
            f.write_chunks(fields, offset="auto")
            f.end_frame()
        stats[rank] = (f.collective_count - c0) / float(len(seeds))
        f.close()
        pdist.release(comm)
    except Exception:  # pragma: no cover
        import traceback
        errors.append((rank, traceback.format_exc()))


def main():
    kind, P, path, what = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    host = len(sys.argv) > 5 and sys.argv[5] == "host"
    shm_name = "pgsdthr_%s" % uuid.uuid4().hex[:10]
    uid = pdist.rccl_unique_id() if kind == "rccl" else None
    errors, stats = [], [None] * P
    threads = [threading.Thread(target=rank_main, args=(r, P, kind, path, what, shm_name, uid, errors, stats, host))
               for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=240)
    if errors or any(t.is_alive() for t in threads):
        for rank, tb in errors:
            sys.stderr.write("rank %d:\n%s\n" % (rank, tb))
        os._exit(1)                 # threads stuck in a collective cannot be joined
    print("RESULT collectives_per_frame=%s" % ",".join("%g" % s for s in stats))


if __name__ == "__main__":
    main()
