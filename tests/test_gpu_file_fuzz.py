"""Randomised FRAMES through the device write path against the oracle's file, byte for byte: random row counts on
both sides of the direct-path threshold (set to 64 KiB here, so frames of a few thousand rows already mix the two
roads: packed straight into pinned host memory and written by the sealing thread / staged in HBM, copied, written by
the pipeline), random field sets (float4 and double4 sources, the type id as bits of position.w, scalar and int3
arrays), replicated host chunks in between, the frame exchange batched or not or a declared partition
(pgsd_set_partition, offset='auto'), frames sealed synchronously or asynchronously, host per-particle arrays with and
without deferred rows, neighbouring direct chunks in one pwritev or one pwrite each.  Seeds are fixed."""
import os

import numpy as np
import pytest

import gpu_common as G
from test_gpu_file import _oracle_frames

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SEEDS = int(os.environ.get("PGSD_FUZZ_SEEDS", "24"))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_frames_match_the_oracle_file(seed, tmp_path, monkeypatch):
    import pgsd.fl as fl
    monkeypatch.setenv("PGSD_DIRECT_MAX_KIB", "64")
    rng = np.random.default_rng(7000 + seed)
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    batched = bool(rng.random() < 0.6)
    deferred = batched and bool(rng.random() < 0.5)
    declared = not batched and bool(rng.random() < 0.5)
    if rng.random() < 0.3:
        monkeypatch.setenv("PGSD_DIRECT_COALESCE", "0")
    f = fl.open(mine, "w", application="app", schema="hoomd", schema_version=[1, 4])
    if rng.random() < 0.3:
        f.configure_device(slab_bytes=int(rng.choice([4096, 65536, 1 << 20])), n_slabs=int(rng.integers(1, 4)))
    f.frame_exchange = batched
    if deferred:
        f.deferred_rows = True
    frames = []
    keep = []
    for frame in range(int(rng.integers(2, 6))):
        N = int(rng.choice([0, 1, 17, 300, 1000, 2500, 6000, 20_000, 60_000]))
        pos = G.rand_array(rng, (N, 4), np.float64 if rng.random() < 0.3 else np.float32)
        vel = G.rand_array(rng, (N, 4), np.float32)
        tid = rng.integers(0, 7, size=N, dtype=np.uint32)
        if pos.dtype == np.float32:
            pos[:, 3] = tid.view(np.float32)
        img = rng.integers(-4, 5, size=(N, 3)).astype(np.int32)
        dens = G.rand_array(rng, (N,), np.float32)
        dpos, dvel, dimg, ddens, dtid = dev(pos), dev(vel), dev(img), dev(dens), dev(tid.view(np.int32))
        keep += [dpos, dvel, dimg, ddens, dtid]
        chunks = []
        rows = "auto" if declared else np.array([N])
        if declared:
            f.set_partition([N])
        step = np.array([100 * frame + seed], dtype=np.uint64)
        f.write_chunk("configuration/step", step, write_all=False)
        chunks.append(("configuration/step", 4, 1, False, [step.reshape(1, 1)]))
        # a random subset of device fields, in one or two fused launches, a host array possibly in between
        cand = [("particles/position", fl.DeviceField.from_tensor(dpos, columns=(0, 3), out_dtype=np.float32), 9, 3,
                 G.oracle_pack(pos, 3, out_dtype=np.float32)),
                ("particles/velocity", fl.DeviceField.from_tensor(dvel, columns=(0, 3)), 9, 3, G.oracle_pack(vel, 3)),
                ("particles/mass", fl.DeviceField.from_tensor(dvel, columns=(3, 4)), 9, 1, G.oracle_pack(vel, 1, col0=3)),
                ("particles/image", dimg, 7, 3, img),
                ("particles/density", ddens, 9, 1, dens.reshape(-1, 1))]
        if pos.dtype == np.float32:
            cand.append(("particles/typeid", fl.DeviceField.from_tensor(dpos, columns=(3, 4), out_dtype=np.uint32, bitcast=True),
                         3, 1, tid.reshape(-1, 1)))
        else:
            cand.append(("particles/typeid", fl.DeviceField.from_tensor(dtid, out_dtype=np.uint32), 3, 1, tid.reshape(-1, 1)))
        picked = [cand[i] for i in sorted(rng.choice(len(cand), size=int(rng.integers(1, len(cand) + 1)), replace=False))]
        cut = int(rng.integers(0, len(picked) + 1))
        for group in (picked[:cut], picked[cut:]):
            if group:
                f.write_chunks([(name, field) for name, field, _, _, _ in group], offset=rows)
                chunks += [(name, t, M, True, [np.ascontiguousarray(exp)]) for name, _, t, M, exp in group]
            if group is picked[:cut] and rng.random() < 0.5:
                host = G.rand_array(rng, (N, 2), np.float32)
                keep.append(host)
                f.write_chunk("particles/host_field", host, offset=rows)
                chunks.append(("particles/host_field", 9, 2, True, [host]))
        if rng.random() < 0.5:
            box = G.rand_array(rng, (6,), np.float32)
            f.write_chunk("configuration/box", box, write_all=False)
            chunks.append(("configuration/box", 9, 1, False, [box.reshape(6, 1)]))
        f.end_frame(wait=bool(rng.random() < 0.6))
        frames.append(chunks)
    f.close()
    _oracle_frames(ref, 1, frames)
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read(), (seed, batched, deferred, declared)
