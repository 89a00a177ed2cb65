"""`HOOMDTrajectory.append` on per-handle communicators: P ranks as P THREADS of this process (one communicator and one
file object each, all on cuda:0), every per-particle array in HBM, the elision comparisons on the GPU and their votes
riding in each frame's one allgather.

    python hoomd_threads_worker.py <shm|rccl> <P> <seed> <out.gsd>

The frames are `test_hoomd_append_oracle.random_frames(seed, P)`; the caller compares the file with the model's.
`rccl`: the RCCL back end's code over the stand-in librccl (PGSD_RCCL_LIBRARY, PGSD_FAKE_RCCL_SYNC=1 set by the
caller) -- an exchange and a comparison launch then follow each other on the same thread."""
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
import pgsd.hoomd as hoomd
import test_gpu_config4 as C4
import test_hoomd_append_oracle as A


def main():
    kind, P, seed, mine = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    frames = A.random_frames(seed, P)
    shm = "pgsdthr_%s" % uuid.uuid4().hex[:10]
    uid = pdist.rccl_unique_id() if kind == "rccl" else None
    errors = []

    def rank_main(rank):
        try:
            torch.cuda.set_device(0)
            comm = pdist.create_shm(shm, rank, P) if kind == "shm" else pdist.create_rccl(uid, rank, P, 0)
            f = fl.open(mine, "w", application="pgsd.hoomd 3.2.0", schema="hoomd", schema_version=[1, 4], comm=comm)
            t = hoomd.HOOMDTrajectory(f)
            for k, g in enumerate(frames):
                fr = C4._device_frame(hoomd, fl, g, g["counts"], rank)
                if g["explicit"]:
                    fr.part_dist = np.array(g["counts"], dtype=np.uint64)
                t.append(fr, wait=(k % 2 == 0))
            t.close()
            pdist.release(comm)
        except Exception:  # pragma: no cover
            import traceback
            errors.append((rank, traceback.format_exc()))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    if errors or any(th.is_alive() for th in threads):
        sys.stderr.write(repr(errors))
        sys.stderr.flush()
        os._exit(1)
    print("OK")


if __name__ == "__main__":
    main()
