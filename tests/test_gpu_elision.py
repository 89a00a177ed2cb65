"""Elision of GPU-resident arrays (hoomd.py:654-694 for arrays that live in HBM): `pgsd_compare_staged_chunks` /
`pgsd_copy_staged_chunks` against numpy, and `HOOMDTrajectory.append` with device arrays against the same frames
appended from host arrays -- the files must be identical byte for byte."""
import numpy as np
import pytest

import gpu_common as G

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _fields(fl, pos4, vel4, dens):
    return [("particles/position", fl.DeviceField.from_tensor(pos4, columns=(0, 3))),
            ("particles/typeid", fl.DeviceField.from_tensor(pos4, columns=(3, 4), out_dtype=np.uint32, bitcast=True)),
            ("particles/velocity", fl.DeviceField.from_tensor(vel4, columns=(0, 3), out_dtype=np.float32)),
            ("particles/density", dens)]


@pytest.mark.parametrize("N", [0, 1, 1001, 40_000, 1 << 20])
def test_compare_and_copy_staged_against_numpy(N, tmp_gsd):
    """1001 rows: the direct (pinned host) staging, tails that are not whole 16-byte vectors; 2^20 rows: HBM staging.
    A difference in the very last element, in the first, none; a reference that is not 16-byte aligned; no reference."""
    import pgsd.fl as fl
    rng = np.random.default_rng(N + 3)
    pos = rng.standard_normal((N, 4)).astype(np.float32)
    pos[:, 3] = rng.integers(0, 5, size=N).astype(np.uint32).view(np.float32)
    vel = rng.standard_normal((N, 4))                                   # float64 source, float32 chunk
    dens = rng.standard_normal(N).astype(np.float32)
    dpos, dvel, ddens = dev(pos), dev(vel), dev(dens)
    packed = [np.ascontiguousarray(pos[:, :3]), np.ascontiguousarray(pos[:, 3]).view(np.uint32),
              np.ascontiguousarray(vel[:, :3]).astype(np.float32), dens]
    with fl.open(tmp_gsd, "w", application="app", schema="hoomd", schema_version=[1, 4]) as f:
        f.frame_exchange = True
        t = f.stage_chunks(_fields(fl, dpos, dvel, ddens))
        refs = f.copy_staged(t, 0, [p.nbytes for p in packed])
        assert f.compare_staged(t, 0, refs) == [True] * 4               # a chunk equals its own copy
        for r, p in zip(refs, packed):
            assert isinstance(r, fl.DeviceBuffer)                       # the library's own device memory (no torch)
            assert r.to_host().tobytes() == p.tobytes()                 # ... which holds the packed chunk
            assert torch.as_tensor(r, device="cuda").cpu().numpy().tobytes() == p.tobytes()   # __cuda_array_interface__
        f.write_staged(t, 0, 4, offset=np.array([N]))
        with pytest.raises(RuntimeError):
            f.compare_staged(t, 0, refs)                                # written chunks are gone
        f.end_frame()
        if N == 0:
            return
        # frame 1: the last element of the velocity and the first type id change
        dvel[N - 1, 2] += 1.0
        dpos[0, 3] = torch.tensor([9], dtype=torch.int32).view(torch.float32)[0]
        t = f.stage_chunks(_fields(fl, dpos, dvel, ddens))
        assert f.compare_staged(t, 0, refs) == [True, False, False, True]
        assert f.compare_staged(t, 1, [None, refs[2]]) == [False, False]
        # a reference at an odd address: the byte-wise road of the kernel
        odd = torch.empty(packed[3].nbytes + 4, dtype=torch.uint8, device="cuda")[4:]
        odd.copy_(torch.as_tensor(refs[3], device="cuda"))
        assert odd.data_ptr() % 16 != 0
        assert f.compare_staged(t, 3, [odd]) == [True]
        odd[-1] ^= 1
        assert f.compare_staged(t, 3, [odd]) == [False]
        if N > 1:
            with pytest.raises(ValueError):
                f.compare_staged(t, 3, [torch.empty(N * 4 + 1, dtype=torch.uint8, device="cuda")])
        f.write_staged(t, 1, 2, offset=np.array([N]))                   # only what differs is written
        f.end_frame()                                                   # ... the other two staged chunks are dropped
    with fl.open(tmp_gsd, "r") as f:
        assert f.nframes == 2
        assert not f.chunk_exists(1, "particles/position") and not f.chunk_exists(1, "particles/density")
        assert f.read_chunk(1, "particles/typeid")[0] == 9
        assert f.read_chunk(1, "particles/velocity")[N - 1, 2] == np.float32(vel[N - 1, 2] + 1.0)


def _frame(hoomd, fl, step, pos, tid, mass, vel, dens, on_gpu, keep):
    fr = hoomd.Frame()
    fr.configuration.step = step
    fr.particles.N = pos.shape[0]
    fr.particles.types = ["A", "B", "C"]
    if on_gpu:
        p4 = np.zeros((pos.shape[0], 4), np.float32)
        p4[:, :3], p4[:, 3] = pos, tid.view(np.float32)
        v4 = np.zeros((pos.shape[0], 4), np.float32)
        v4[:, :3], v4[:, 3] = vel, mass
        dp, dv, dd = dev(p4), dev(v4), dev(dens)
        keep += [dp, dv, dd]
        fr.particles.position = fl.DeviceField.from_tensor(dp, columns=(0, 3))
        fr.particles.typeid = fl.DeviceField.from_tensor(dp, columns=(3, 4), out_dtype=np.uint32, bitcast=True)
        fr.particles.velocity = fl.DeviceField.from_tensor(dv, columns=(0, 3))
        fr.particles.mass = fl.DeviceField.from_tensor(dv, columns=(3, 4))
        fr.particles.density = dd
    else:
        fr.particles.position, fr.particles.typeid, fr.particles.velocity = pos, tid, vel
        fr.particles.mass, fr.particles.density = mass, dens
    return fr


def _trajectory(rng, N, frames):
    """Type id and mass never change, the density changes once (frame 2) and goes back, positions always move."""
    tid = rng.integers(0, 3, size=N).astype(np.uint32)
    mass = (1.0 + rng.random(N)).astype(np.float32)        # != the default 1.0 anywhere
    dens0 = (2.0 + rng.random(N)).astype(np.float32)
    out = []
    for k in range(frames):
        dens = dens0 + np.float32(1.0) if k == 2 else dens0
        out.append((10 * k, rng.standard_normal((N, 3)).astype(np.float32), tid, mass,
                    rng.standard_normal((N, 3)).astype(np.float32), dens))
    return out


@pytest.mark.parametrize("N", [777, 200_000])
def test_append_elides_static_gpu_arrays_like_host_arrays(N, tmp_path):
    """The same frames from GPU-resident and from host arrays: identical files -- the density, which differs in frame 2
    only, is written there and elided again afterwards, on both paths."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    frames = _trajectory(np.random.default_rng(N), N, 5)
    a, b = str(tmp_path / "gpu.gsd"), str(tmp_path / "host.gsd")
    keep = []
    for path, on_gpu in ((a, True), (b, False)):
        with hoomd.open(path, "w") as t:
            for args in frames:
                t.append(_frame(hoomd, fl, *args, on_gpu, keep))
            if on_gpu:
                assert not t._dev_off
                assert set(t._dev_ref) == {"particles/" + n for n in ("position", "typeid", "velocity", "mass", "density")}
    with open(a, "rb") as fa, open(b, "rb") as fb:
        assert fa.read() == fb.read()
    with hoomd.open(a, "r") as t:
        f = t.file
        assert [f.chunk_exists(k, "particles/typeid") for k in range(5)] == [True, False, False, False, False]
        assert [f.chunk_exists(k, "particles/mass") for k in range(5)] == [True, False, False, False, False]
        assert [f.chunk_exists(k, "particles/density") for k in range(5)] == [True, False, True, False, False]
        assert all(f.chunk_exists(k, "particles/position") for k in range(5))
        for k, (step, pos, tid, mass, vel, dens) in enumerate(frames):
            fr = t[k]
            assert fr.configuration.step == step
            for name, want in (("position", pos), ("typeid", tid), ("mass", mass), ("velocity", vel), ("density", dens)):
                assert getattr(fr.particles, name).tobytes() == want.tobytes(), (k, name)


def test_once_mode_is_gone(tmp_path):
    """Round 3's `device_elision = 'once'` shortcut (stop comparing an array once it differed) was removed in round 5:
    any truthy value is the one remaining mode -- every array compared in every frame -- so the density that differs in
    frame 2 only is elided again in frames 3 and 4, as the host path elides it."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    frames = _trajectory(np.random.default_rng(5), 3000, 5)
    path, keep = str(tmp_path / "gpu.gsd"), []
    with hoomd.open(path, "w") as t:
        t.device_elision = 'once'
        assert not hasattr(t, "_dev_dynamic")
        for k, args in enumerate(frames):
            t.append(_frame(hoomd, fl, *args, True, keep))
    with hoomd.open(path, "r") as t:
        f = t.file
        assert [f.chunk_exists(k, "particles/typeid") for k in range(5)] == [True, False, False, False, False]
        assert [f.chunk_exists(k, "particles/density") for k in range(5)] == [True, False, True, False, False]


def test_append_device_elision_off_writes_everything(tmp_path):
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    frames = _trajectory(np.random.default_rng(4), 500, 3)
    path, keep = str(tmp_path / "t.gsd"), []
    with hoomd.open(path, "w") as t:
        t.device_elision = False
        for args in frames:
            t.append(_frame(hoomd, fl, *args, True, keep))
    with hoomd.open(path, "r") as t:
        assert all(t.file.chunk_exists(k, "particles/typeid") for k in range(3))


def test_append_to_an_existing_file_compares_with_rows_read_from_frame_0(tmp_path):
    """Reopened trajectory: frame 0's rows are not in HBM any more, they are read from the file
    (`read_chunk_device`) -- same files as the host path again."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    frames = _trajectory(np.random.default_rng(8), 3000, 4)
    a, b = str(tmp_path / "gpu.gsd"), str(tmp_path / "host.gsd")
    keep = []
    for path, on_gpu in ((a, True), (b, False)):
        with hoomd.open(path, "w") as t:
            t.append(_frame(hoomd, fl, *frames[0], on_gpu, keep))
        with hoomd.open(path, "r+") as t:
            t.append(_frame(hoomd, fl, *frames[1], on_gpu, keep))
            if on_gpu:      # every array was compared, each with rows read from the file
                assert set(t._dev_ref) == {"particles/" + n for n in ("position", "typeid", "velocity", "mass", "density")}
    with open(a, "rb") as fa, open(b, "rb") as fb:
        assert fa.read() == fb.read()


def test_particle_count_change_and_back_follows_the_host_path(tmp_path):
    """One rank: while the particle count differs from frame 0's nothing can equal frame 0 -- everything is written;
    when it is frame 0's count again the arrays are compared again, as `numpy.array_equal` compares host arrays of
    equal shape (hoomd.py:679-682).  Same file as the host path."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    rng = np.random.default_rng(2)
    f3 = _trajectory(rng, 400, 2)
    other = _trajectory(rng, 300, 1)[0]
    a, b = str(tmp_path / "gpu.gsd"), str(tmp_path / "host.gsd")
    keep = []
    for path, on_gpu in ((a, True), (b, False)):
        with hoomd.open(path, "w") as t:
            t.append(_frame(hoomd, fl, *f3[0], on_gpu, keep))
            t.append(_frame(hoomd, fl, *f3[1], on_gpu, keep))
            t.append(_frame(hoomd, fl, 99, *other[1:], on_gpu, keep))       # 300 particles
            back = (100,) + f3[0][1:]
            t.append(_frame(hoomd, fl, *back, on_gpu, keep))                 # 400 again, equal to frame 0: elided again
            if on_gpu:
                assert not t._dev_off
    with open(a, "rb") as fa, open(b, "rb") as fb:
        assert fa.read() == fb.read()
    with hoomd.open(a, "r") as t:
        ex = [t.file.chunk_exists(k, "particles/typeid") for k in range(4)]
        assert ex == [True, False, True, False]
        assert t[2].particles.N == 300 and t[3].particles.typeid.tobytes() == f3[0][2].tobytes()


# ---------------------------------------------------------------- two ranks: the outcome is agreed
def _one_sided_frames():
    """Frame 1: type id, mass, image and density as in frame 0 except ONE density value in the last row (the last
    rank's); frame 2: everything as in frame 0 again -- the density is elided again."""
    import test_hoomd_append_oracle as A
    f0 = A.global_frames()[0]
    n = f0["n"]
    frames = [f0]
    for k in (1, 2):
        p0 = f0["particles"]
        dens = np.array(p0["density"], dtype=np.float32, copy=True)
        if k == 1:
            dens[n - 1] += 1.0
        g = {"configuration": {"step": 100 + k, "dimensions": None, "box": f0["configuration"]["box"]},
             "particles": {"types": p0["types"], "typeid": p0["typeid"], "mass": p0["mass"], "image": p0["image"],
                           "density": dens, "position": np.asarray(p0["position"]) + np.float32(k),
                           "type_shapes": p0["type_shapes"]},
             "constraints": f0["constraints"], "log": {}, "state": {}, "n": n}
        frames.append(g)
    return frames


def _append_rank(rank, P, shm, path, q):
    try:
        import os
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "pgsd-sph_amd"))
        sys.path.insert(0, os.path.join(root, "tests"))
        import torch as _t
        import pgsd.fl as fl
        import pgsd.hoomd as hoomd
        from pgsd import _lib
        import test_gpu_config4 as C4
        import test_gpu_elision as me
        import test_hoomd_append_oracle as A
        __import__("pgsd.dist").dist.init_shm(shm, rank, P)
        _t.cuda.set_device(0)
        t = hoomd.open(path, "w")
        for g in me._one_sided_frames():
            t.append(C4._device_frame(hoomd, fl, g, A.partition(g["n"], P), rank))
        dyn = []                        # (round 3's 'once' mode kept a set of arrays taken off the comparisons here)
        t.close()
        _lib.lib.pgsd_comm_finalize()
        q.put((rank, "ok", dyn))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), None))
        raise


def test_two_ranks_agree_on_what_is_elided(tmp_path):
    """A difference in ONE rank's rows: both ranks write the chunk (the vote rides in the frame's allgather), and the
    file is the model's (tests/test_hoomd_append_oracle.py: GPU-resident arrays are decided like host arrays)."""
    import multiprocessing as mp
    import uuid
    import test_hoomd_append_oracle as A
    P = 2
    ref, mine = str(tmp_path / "ref.gsd"), str(tmp_path / "mine.gsd")
    written = A.expected_file(ref, P, device=True, frames=_one_sided_frames())
    assert "particles/density" in written[1] and "particles/density" not in written[2]
    assert "particles/typeid" not in written[1] and "particles/mass" not in written[2]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shm = "pgsdgpu_%s" % uuid.uuid4().hex[:10]
    procs = [ctx.Process(target=_append_rank, args=(r, P, shm, mine, q)) for r in range(P)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(msg == "ok" for _, msg, _ in results), results
    assert results[0][2] == results[1][2] == []               # nothing is taken off the comparisons
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()


def _reopen_rank(rank, P, shm, path, explicit, q):
    try:
        import os
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "pgsd-sph_amd"))
        sys.path.insert(0, os.path.join(root, "tests"))
        import torch as _t
        import pgsd.fl as fl
        import pgsd.hoomd as hoomd
        from pgsd import _lib
        import test_gpu_elision as me
        __import__("pgsd.dist").dist.init_shm(shm, rank, P)
        _t.cuda.set_device(0)
        counts = [700, 300]
        row0 = sum(counts[:rank])
        frames = me._trajectory(np.random.default_rng(77), sum(counts), 3)
        keep = []

        def local(args):
            step, pos, tid, mass, vel, dens = args
            s = slice(row0, row0 + counts[rank])
            fr = me._frame(hoomd, fl, step, pos[s], tid[s], mass[s], vel[s], dens[s], True, keep)
            if explicit:
                fr.part_dist = np.array(counts, dtype=np.uint64)
            return fr

        t = hoomd.open(path, "w")
        t.append(local(frames[0]))
        t.close()
        t = hoomd.open(path, "r+")
        t.append(local(frames[1]))
        t.append(local(frames[2]))
        refs = sorted(t._dev_ref)
        t.close()
        _lib.lib.pgsd_comm_finalize()
        q.put((rank, "ok", refs))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), None))
        raise


@pytest.mark.parametrize("explicit", [False, True])
def test_two_ranks_reopened_file_reads_each_ranks_rows_of_frame_0(explicit, tmp_path):
    """A trajectory reopened by two ranks: frame 0's rows come from the FILE, each rank its own (rank 1 from row 700
    on).  With the partition stated by the caller (`Frame.part_dist`) the first appended frame is already compared;
    without, the partition is known after that frame's exchange and the comparisons start with the next one."""
    import multiprocessing as mp
    import uuid
    import pgsd.hoomd as hoomd
    P = 2
    path = str(tmp_path / "t.gsd")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shm = "pgsdgpu_%s" % uuid.uuid4().hex[:10]
    procs = [ctx.Process(target=_reopen_rank, args=(r, P, shm, path, explicit, q)) for r in range(P)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(msg == "ok" for _, msg, _ in results), results
    assert results[0][2] == results[1][2] == ["particles/" + n for n in ("density", "mass", "position", "typeid", "velocity")]
    frames = _trajectory(np.random.default_rng(77), 1000, 3)
    with hoomd.open(path, "r") as t:
        f = t.file
        assert [f.chunk_exists(k, "particles/typeid") for k in range(3)] == [True, not explicit, False]
        assert [f.chunk_exists(k, "particles/density") for k in range(3)] == [True, not explicit, True]
        assert all(f.chunk_exists(k, "particles/position") for k in range(3))
        for k, (step, pos, tid, mass, vel, dens) in enumerate(frames):
            fr = t[k]
            for name, want in (("position", pos), ("typeid", tid), ("mass", mass), ("velocity", vel), ("density", dens)):
                assert getattr(fr.particles, name).tobytes() == want.tobytes(), (k, name)


def test_device_reader_keeps_frame_0_rows_of_elided_arrays(tmp_path):
    """`read_frame_device` on a trajectory whose static arrays were elided: the rows of frame 0 are read from the file
    once and handed out as copies -- changing what a read returned changes nothing a later read returns."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    frames = _trajectory(np.random.default_rng(12), 6000, 4)
    path, keep = str(tmp_path / "t.gsd"), []
    with hoomd.open(path, "w") as t:
        for args in frames:
            t.append(_frame(hoomd, fl, *args, True, keep))
    with hoomd.open(path, "r") as t:
        assert not t.file.chunk_exists(3, "particles/typeid")
        for sweep in range(2):
            for k in (1, 2, 3, 0, 3):
                step, pos, tid, mass, vel, dens = frames[k]
                fr = t.read_frame_device(k, part=(1000, 4000))
                s = slice(1000, 5000)
                for name, want in (("position", pos), ("typeid", tid), ("mass", mass), ("velocity", vel), ("density", dens)):
                    got = getattr(fr.particles, name)
                    assert got.cpu().numpy().tobytes() == want[s].tobytes(), (sweep, k, name)
                fr.particles.typeid.zero_()                 # the caller's copy, not the trajectory's
                fr.particles.mass.fill_(-1.0)
            assert set(t._frame0_dev_cache) == {"particles/typeid", "particles/mass", "particles/density"}
        fr = t.read_frame_device(1, part=(0, 6000))         # another partition: its own rows
        assert fr.particles.typeid.cpu().numpy().tobytes() == frames[0][2].tobytes()
        assert t._frame0_dev_part == (0, 6000) and set(t._frame0_dev_cache) == {"particles/typeid", "particles/mass", "particles/density"}


def _random_rank(rank, P, shm, path, seed, q):
    try:
        import os
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "pgsd-sph_amd"))
        sys.path.insert(0, os.path.join(root, "tests"))
        import torch as _t
        import pgsd.fl as fl
        import pgsd.hoomd as hoomd
        from pgsd import _lib
        import test_gpu_config4 as C4
        import test_hoomd_append_oracle as A
        __import__("pgsd.dist").dist.init_shm(shm, rank, P)
        _t.cuda.set_device(0)
        t = hoomd.open(path, "w")
        for k, g in enumerate(A.random_frames(seed, P)):
            fr = C4._device_frame(hoomd, fl, g, g["counts"], rank)
            if g["explicit"]:
                fr.part_dist = np.array(g["counts"], dtype=np.uint64)
            t.append(fr, wait=(k % 2 == 1))
        t.close()
        _lib.lib.pgsd_comm_finalize()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("seed,P", [(1, 2), (8, 2), (9, 3), (13, 2), (23, 3), (4, 3)])
def test_random_multi_rank_device_trajectories_match_the_model(seed, P, tmp_path):
    """The random trajectories of tests/test_hoomd_append_oracle.py with every per-particle array in HBM, two or three
    ranks sharing the GPU: the file is the model's -- GPU-resident arrays are decided exactly as host arrays are
    (every rank's rows against its rows of frame 0, the comparisons ending with a change of the partition)."""
    import multiprocessing as mp
    import uuid
    import test_hoomd_append_oracle as A
    ref, mine = str(tmp_path / "ref.gsd"), str(tmp_path / "mine.gsd")
    A.expected_file(ref, P, device=True, frames=A.random_frames(seed, P))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shm = "pgsdgpu_%s" % uuid.uuid4().hex[:10]
    procs = [ctx.Process(target=_random_rank, args=(r, P, shm, mine, seed, q)) for r in range(P)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(msg == "ok" for _, msg in results), results
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()


def test_equality_is_numpys_nan_signed_zero_and_defaults(tmp_path):
    """The GPU comparison decides as `numpy.array_equal` does (hoomd.py:679-682): a static array holding a NaN is
    WRITTEN (NaN != NaN), +0.0 against -0.0 is elided; and the default-value half of the rule: an array that equals the
    default everywhere is not written when frame 0 has no such chunk -- in frame 0 itself too.  Files identical to the
    host path's."""
    import pgsd.hoomd as hoomd
    N = 9000
    dens = np.linspace(1, 2, N).astype(np.float32)
    dens[7] = np.nan
    energy0 = np.full(N, 3.0, np.float32)
    energy0[5] = 0.0
    energy1 = energy0.copy()
    energy1[5] = -0.0
    body = np.full(N, -1, np.int32)                              # the default
    vel0 = np.zeros((N, 3), np.float32)                         # the default
    vel0[N - 1, 2] = -0.0                                       # ... by value
    vel2 = vel0.copy()
    vel2[N // 2, 1] = 1e-30
    a, b = str(tmp_path / "gpu.gsd"), str(tmp_path / "host.gsd")
    for path, put in ((a, dev), (b, lambda x: x)):
        with hoomd.open(path, "w") as t:
            for k, (energy, vel) in enumerate(((energy0, vel0), (energy1, vel0), (energy1, vel2))):
                fr = hoomd.Frame()
                fr.configuration.step = k
                fr.particles.N = N
                fr.particles.density = put(dens)
                fr.particles.energy = put(energy)
                fr.particles.body = put(body)
                fr.particles.velocity = put(vel)
                t.append(fr)
    with open(a, "rb") as fa, open(b, "rb") as fb:
        assert fa.read() == fb.read()
    with hoomd.open(a, "r") as t:
        ex = lambda name: [t.file.chunk_exists(k, "particles/" + name) for k in range(3)]
        assert ex("density") == [True, True, True] and ex("energy") == [True, False, False]
        assert ex("body") == [False, False, False] and ex("velocity") == [False, False, True]
        assert t[1].particles.density.tobytes() == dens.tobytes()
        assert t[2].particles.velocity.tobytes() == vel2.tobytes()


def test_compare_staged_short_references_repeat(tmp_gsd):
    """`compare_staged` with a reference shorter than the chunk: the reference repeats -- 4096 rows of a value stand for
    any number of rows; 16-byte vectors that straddle rows (12-byte rows), a difference in the very last row, in a row
    beyond the first period, NaN rows (never equal), unaligned chunk sizes."""
    import pgsd.fl as fl
    N = 100_003
    row = np.array([0.5, -2.0, 7.25], np.float32)
    pos = np.zeros((N, 4), np.float32)
    pos[:, :3] = row
    ids = np.full(N, 7, np.int32)
    dpos, dids = dev(pos), dev(ids)
    ref_rows = torch.from_numpy(np.tile(row, (4096, 1)).view(np.uint8).reshape(-1)).cuda()
    ref_ids = torch.from_numpy(np.full(4096, 7, np.int32).view(np.uint8)).cuda()
    with fl.open(tmp_gsd, "w", application="app", schema="hoomd", schema_version=[1, 4]) as f:
        f.frame_exchange = True

        def same():
            t = f.stage_chunks([("particles/position", fl.DeviceField.from_tensor(dpos, columns=(0, 3))),
                                ("particles/body", dids)])
            out = f.compare_staged(t, 0, [ref_rows, ref_ids])
            f.end_frame()
            return out
        assert same() == [True, True]
        dpos[N - 1, 2] = 7.5
        assert same() == [False, True]
        dpos[N - 1, 2] = 7.25
        dids[50_000] = 8
        assert same() == [True, False]
        dids[50_000] = 7
        dpos[4097, 0] = float("nan")
        ref_nan = ref_rows.clone().view(torch.float32)
        ref_nan[4097 % 4096 * 3] = float("nan")                 # the same bits at the same place: still not equal
        t = f.stage_chunks([("particles/position", fl.DeviceField.from_tensor(dpos, columns=(0, 3)))])
        assert f.compare_staged(t, 0, [ref_nan.view(torch.uint8)]) == [False]
        with pytest.raises(RuntimeError):
            f.compare_staged(t, 0, [ref_rows[:1200]])           # too short to repeat
        f.end_frame()


def test_long_trajectory_with_reopen_matches_the_host_path(tmp_path):
    """240 frames -- several relocations of the on-disk index, asynchronous seals among them, the trajectory closed
    and reopened twice (frame 0's rows then come from the file) -- with GPU-resident arrays and with host arrays:
    the same file."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    rng = np.random.default_rng(31)
    N = 1500
    tid = rng.integers(0, 3, size=N).astype(np.uint32)
    mass = (1.0 + rng.random(N)).astype(np.float32)
    dens = (2.0 + rng.random(N)).astype(np.float32)
    frames = [(k, rng.standard_normal((N, 3)).astype(np.float32), tid, mass,
               rng.standard_normal((N, 3)).astype(np.float32), dens) for k in range(240)]
    a, b = str(tmp_path / "gpu.gsd"), str(tmp_path / "host.gsd")
    keep = []
    for path, on_gpu in ((a, True), (b, False)):
        for lo, hi, mode in ((0, 90, "w"), (90, 170, "r+"), (170, 240, "r+")):
            with hoomd.open(path, mode) as t:
                for k in range(lo, hi):
                    fr = _frame(hoomd, fl, *frames[k], on_gpu, keep)
                    fr.log["step_sq"] = np.array([float(k * k)])
                    t.append(fr, wait=(k % 4 != 1) if on_gpu else True)
                if on_gpu:
                    t.file.frame_sync()
            del keep[:]
    with open(a, "rb") as fa, open(b, "rb") as fb:
        assert fa.read() == fb.read()
    with hoomd.open(a, "r") as t:
        assert len(t) == 240 and not t.file.chunk_exists(239, "particles/mass")
        assert t[239].particles.mass.tobytes() == mass.tobytes() and t[200].log["step_sq"][0] == 40000.0


def test_elision_by_hand_as_integration_md_shows_it(tmp_gsd):
    """The snippet of INTEGRATION.md section 3, as written there."""
    import pgsd.fl as fl
    N = 5000
    pos4 = torch.rand((N, 4), device="cuda")
    mass = torch.rand((N,), device="cuda")
    f = fl.open(tmp_gsd, "w", application="app", schema="hoomd", schema_version=[1, 4])
    frame0 = None
    for k in range(3):
        if k:
            pos4[:, :3] += 1.0
        fields = [("particles/position", fl.DeviceField.from_tensor(pos4, columns=(0, 3))), ("particles/mass", mass)]
        first_frame = k == 0
        t = f.stage_chunks(fields)
        if first_frame:
            frame0 = f.copy_staged(t, 0, list(t[2]))
            same = [False] * len(fields)
        else:
            same = f.compare_staged(t, 0, frame0)
        for i, s in enumerate(same):
            if not s:
                f.write_staged(t, i, 1, offset="auto")
        f.end_frame()
        assert same == ([False, False] if first_frame else [False, True])
    f.close()
    with fl.open(tmp_gsd, "r") as g:
        assert g.nframes == 3 and g.chunk_exists(2, "particles/position") and not g.chunk_exists(2, "particles/mass")
        assert g.read_chunk(2, "particles/position").tobytes() == pos4[:, :3].contiguous().cpu().numpy().tobytes()


@pytest.mark.parametrize("kind", ["shm", "rccl"])
def test_four_ranks_as_threads_append_through_pgsd_hoomd(kind, tmp_path):
    """`HOOMDTrajectory.append` on per-handle communicators: four ranks as four THREADS of one process (one
    communicator and one file object each, all on cuda:0), arrays in HBM, the elision votes riding in each frame's
    one allgather -- over the shm back end and over the RCCL back end's code (stand-in librccl: the exchange's
    polling and the comparison launch share a thread) -- the file is the model's."""
    import os
    import subprocess
    import sys
    import product
    import test_hoomd_append_oracle as A
    P, seed = 4, 13
    ref, mine = str(tmp_path / "ref.gsd"), str(tmp_path / "mine.gsd")
    A.expected_file(ref, P, device=True, frames=A.random_frames(seed, P))
    env = dict(os.environ)
    env.pop("PGSD_RCCL_LIBRARY", None)
    if kind == "rccl":
        product.build()
        env.update(PGSD_RCCL_LIBRARY=os.path.join(product.TBUILD, "libpgsd_fake_rccl.so"), PGSD_FAKE_RCCL_SYNC="1")
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hoomd_threads_worker.py")
    p = subprocess.run([sys.executable, worker, kind, str(P), str(seed), mine], env=env, capture_output=True, text=True,
                       timeout=400)
    assert p.returncode == 0 and p.stdout.strip().endswith("OK"), (p.stdout[-300:], p.stderr[-3000:])
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()


def test_exact_mode_elides_an_array_that_returns_to_frame_0(tmp_path):
    """`device_elision = 'exact'` (round 3's name for what is now the default): no array is ever taken off the
    comparisons, so the density that differs in frame 2 only is elided again from frame 3 on -- the host path's file
    for all five frames."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    frames = _trajectory(np.random.default_rng(21), 4000, 5)
    a, b = str(tmp_path / "gpu.gsd"), str(tmp_path / "host.gsd")
    keep = []
    for path, on_gpu in ((a, True), (b, False)):
        with hoomd.open(path, "w") as t:
            t.device_elision = 'exact'
            for args in frames:
                t.append(_frame(hoomd, fl, *args, on_gpu, keep))
            if on_gpu:
                assert len(t._dev_ref) == 5
    with open(a, "rb") as fa, open(b, "rb") as fb:
        assert fa.read() == fb.read()
    with hoomd.open(a, "r") as t:
        assert [t.file.chunk_exists(k, "particles/density") for k in range(5)] == [True, False, True, False, False]


def test_device_reader_without_default_rows(tmp_path):
    """`read_frame_device(defaults=False)`: attributes the file does not hold stay None instead of default rows."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    frames = _trajectory(np.random.default_rng(3), 800, 2)
    path, keep = str(tmp_path / "t.gsd"), []
    with hoomd.open(path, "w") as t:
        for args in frames:
            t.append(_frame(hoomd, fl, *args, True, keep))
    with hoomd.open(path, "r") as t:
        a, b = t.read_frame_device(1), t.read_frame_device(1, defaults=False)
        assert a.particles.body is not None and int(a.particles.body[0]) == -1 and b.particles.body is None
        assert b.particles.image is None and b.particles.mass.cpu().numpy().tobytes() == frames[0][3].tobytes()
        assert b.particles.position.cpu().numpy().tobytes() == frames[1][1].tobytes()


@pytest.mark.parametrize("seed", range(40))
def test_compare_staged_random_cases_against_numpy(seed, tmp_gsd):
    """`compare_staged` against `numpy.array_equal` on random chunks: float32 / float64 / int32, 1-4 columns, sizes on
    both sides of the direct-path threshold, NaNs and zeros of either sign sprinkled in, one element changed somewhere
    (or nothing), zero signs flipped -- with a full-size reference and with a short one that repeats."""
    import pgsd.fl as fl
    rng = np.random.default_rng(7000 + seed)
    dt = [np.float32, np.float64, np.int32][seed % 3]
    M = int(rng.integers(1, 5))
    N = int(rng.choice([1, 5, 333, 4096, 4097, 70_001, 300_000]))
    periodic = bool(rng.random() < 0.5) and N * M * np.dtype(dt).itemsize > 16384
    if periodic:
        rows = 4096 if (4096 * M * np.dtype(dt).itemsize) % 16 == 0 else 4096 * 4
        pattern = (rng.standard_normal((rows, M)) * 3).astype(dt)
        if dt != np.int32 and rng.random() < 0.5:
            pattern.reshape(-1)[rng.integers(0, pattern.size, size=20)] = 0.0
        a = np.tile(pattern, (N // rows + 1, 1))[:N].copy()
        ref = pattern
    else:
        a = (rng.standard_normal((N, M)) * 3).astype(dt)
        if dt != np.int32 and rng.random() < 0.5:
            a.reshape(-1)[rng.integers(0, a.size, size=max(1, a.size // 40))] = 0.0
        ref = a.copy()
    what = rng.choice(["same", "flip_zeros", "change_one", "nan_both", "nan_chunk"])
    if what == "flip_zeros" and dt != np.int32:
        z = a == 0
        a[z] = -a[z]                                        # equal by value, different bytes
    elif what == "change_one":
        i = (int(rng.integers(0, N)), int(rng.integers(0, M)))
        a[i] = a[i] + 1 if dt == np.int32 else a[i] * 2 + 1
    elif what == "nan_both" and dt != np.int32 and not periodic:
        i = (int(rng.integers(0, N)), int(rng.integers(0, M)))
        a[i] = ref[i] = np.nan                              # the same bits on both sides: still not equal
    elif what == "nan_chunk" and dt != np.int32:
        a[int(rng.integers(0, N)), int(rng.integers(0, M))] = np.nan
    full = np.tile(ref, (N // ref.shape[0] + 1, 1))[:N] if periodic else ref
    want = bool(np.array_equal(a, full))
    da = torch.from_numpy(a).cuda()
    dref = torch.from_numpy(np.ascontiguousarray(ref).view(np.uint8).reshape(-1)).cuda()
    with fl.open(tmp_gsd, "w", application="app", schema="hoomd", schema_version=[1, 4]) as f:
        f.frame_exchange = True
        t = f.stage_chunks([("log/x", da)])
        assert f.compare_staged(t, 0, [dref]) == [want], (seed, dt, N, M, periodic, what)
        f.end_frame()
