"""Edge cases of the device entry points (through pgsd.fl / the C ABI)."""
import ctypes
import os

import numpy as np
import pytest

import gpu_common as G
import scenario as S

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_replicated_small_chunk_from_device_takes_the_buffered_path(tmp_path):
    """write_all=False with device data: packed, copied synchronously and appended to the
    small-chunk buffer exactly like host data (pgsd.c:2160-2202)."""
    import pgsd.fl as fl
    a, b = str(tmp_path / "dev.gsd"), str(tmp_path / "host.gsd")
    box = np.array([[10, 11, 12, 0, 0, 0]], dtype=np.float64).T.copy()      # (6,1) float64 on the GPU
    step = np.array([42], dtype=np.uint64)
    for path, on_gpu in ((a, True), (b, False)):
        with fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
            for frame in range(3):
                if on_gpu:
                    f.write_chunk('configuration/box',
                                  fl.DeviceField.from_tensor(dev(box), out_dtype=np.float32), write_all=False)
                    f.write_chunk('configuration/step', dev(step.view(np.int64)).view(torch.int64), write_all=False)
                else:
                    f.write_chunk('configuration/box', box.astype(np.float32), write_all=False)
                    f.write_chunk('configuration/step', step.view(np.int64), write_all=False)
                f.end_frame()
    with open(a, 'rb') as fa, open(b, 'rb') as fb:
        assert fa.read() == fb.read()


def test_zero_rows_and_errors(tmp_gsd):
    import pgsd.fl as fl
    from pgsd import _lib
    with fl.open(tmp_gsd, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
        empty = torch.zeros((0, 4), dtype=torch.float32, device="cuda")
        f.write_chunks([('particles/position', fl.DeviceField.from_tensor(empty, columns=(0, 3)))],
                       offset=np.array([0]))
        f.end_frame()
        assert f.nframes == 1 and f.chunk_exists(0, 'particles/position')
        assert f.read_chunk(0, 'particles/position').shape == (0, 3)
        src = torch.zeros((8, 4), dtype=torch.float32, device="cuda")
        with pytest.raises(RuntimeError, match="Invalid pgsd argument"):
            f.write_chunk('bad', fl.DeviceField(src.data_ptr(), np.float32, 8, 5, stride=4))     # M > stride
        with pytest.raises(RuntimeError, match="Invalid pgsd argument"):
            f.write_chunk('bad', fl.DeviceField(src.data_ptr(), np.float32, 8, 3, stride=4, out_dtype=np.int32))
        with pytest.raises(ValueError):
            f.write_chunks([('a', src), ('b', torch.zeros((9, 4), device="cuda"))])
        f.write_chunk('ok', src)
        f.wait_packed()
        f.end_frame()
    with fl.open(tmp_gsd, 'r') as f:
        src = torch.zeros((8, 4), dtype=torch.float32, device="cuda")
        with pytest.raises(RuntimeError, match="File must be writable"):
            f.write_chunk('nope', src)
        assert f.read_chunk(1, 'ok').shape == (8, 4)


def test_device_and_host_chunks_interleave_with_index_relocation(tmp_path):
    """40 frames x 5 chunks (host and device mixed) force two on-disk index relocations while the
    device pipeline is in use; compare with the oracle."""
    import pgsd.fl as fl
    N = 3001
    rng = np.random.default_rng(8)
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    lib = S.oracle_lib()
    rc = ctypes.c_int(0)
    h = lib.oracle_create_and_open(ref.encode(), 1, b'app', b'hoomd', lib.oracle_make_version(1, 4), 1, 0,
                                   ctypes.byref(rc))
    f = fl.open(mine, 'w', application='app', schema='hoomd', schema_version=[1, 4])
    f.configure_device(slab_bytes=16 * 1024, n_slabs=2)
    for frame in range(40):
        pos4 = G.rand_array(rng, (N, 4), np.float32)
        img = rng.integers(-1, 2, size=(N, 3)).astype(np.int32)
        dens = G.rand_array(rng, (N,), np.float64)
        step = np.array([frame], dtype=np.uint64)
        dpos, ddens = dev(pos4), dev(dens)
        f.write_chunk('configuration/step', step, write_all=False)
        f.write_chunk('particles/position', fl.DeviceField.from_tensor(dpos, columns=(0, 3)), offset=np.array([N]))
        f.write_chunk('particles/image', img, offset=np.array([N]))
        f.write_chunks([('particles/density', fl.DeviceField.from_tensor(ddens, out_dtype=np.float32)),
                        ('particles/w', fl.DeviceField.from_tensor(dpos, columns=(3, 4)))], offset=np.array([N]))
        f.end_frame()
        for name, t, M, all_, arr in (('configuration/step', 4, 1, False, step.reshape(1, 1)),
                                      ('particles/position', 9, 3, True, np.ascontiguousarray(pos4[:, :3])),
                                      ('particles/image', 7, 3, True, img),
                                      ('particles/density', 9, 1, True, dens.astype(np.float32).reshape(-1, 1)),
                                      ('particles/w', 9, 1, True, np.ascontiguousarray(pos4[:, 3:4]))):
            n = arr.shape[0]
            assert S.oracle_write_chunk(lib, h, name, t, [arr], M, n, M, [0], [n * M], all_) == 0
        assert lib.oracle_end_frame(h) == 0
    f.close()
    assert lib.oracle_close(h) == 0
    with open(mine, 'rb') as a, open(ref, 'rb') as b:
        assert a.read() == b.read()


def test_filtered_snapshot_select_then_gather(tmp_path):
    """A filtered snapshot: flags -> pgsd.fl.select_rows (device scan) -> count into the row-count
    exchange -> gather-pack of the selected rows; equals the host-side boolean indexing."""
    import pgsd.dist as pdist
    import pgsd.fl as fl
    N = 200_003
    rng = np.random.default_rng(12)
    pos4 = G.rand_array(rng, (N, 4), np.float32)
    keep = rng.random(N) < 0.37
    dpos, dkeep = dev(pos4), dev(keep)
    index, k = fl.select_rows(dkeep)
    assert k == int(keep.sum())
    assert (index.cpu().numpy() == np.nonzero(keep)[0]).all()
    counts, row0, n_global = pdist.partition_rows(k)
    assert n_global == k and row0 == 0
    path = str(tmp_path / "filtered.gsd")
    with fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
        f.write_chunk('particles/N', np.array([k], dtype=np.uint32), write_all=False)
        f.write_chunks([('particles/position', fl.DeviceField.from_tensor(dpos, columns=(0, 3), order=index)),
                        ('particles/w', fl.DeviceField.from_tensor(dpos, columns=(3, 4), order=index))],
                       offset=counts)
        f.end_frame()
    with fl.open(path, 'r') as f:
        assert f.read_chunk(0, 'particles/position').tobytes() == np.ascontiguousarray(pos4[keep, :3]).tobytes()
        assert f.read_chunk(0, 'particles/w').tobytes() == np.ascontiguousarray(pos4[keep, 3]).tobytes()
    none, k0 = fl.select_rows(torch.zeros(1000, dtype=torch.uint8, device="cuda"))
    assert k0 == 0 and none.numel() == 0


def test_pipeline_soak_random_frames(tmp_path):
    """Stress the slab ring / event hand-offs: 120 frames of random size (including empty ones)
    through tiny slabs and two writer threads, single and fused device writes mixed with host
    chunks; the whole file must equal the oracle's."""
    import pgsd.fl as fl
    rng = np.random.default_rng(2024)
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    lib = S.oracle_lib()
    rc = ctypes.c_int(0)
    h = lib.oracle_create_and_open(ref.encode(), 1, b'app', b'hoomd', lib.oracle_make_version(1, 4), 1, 0,
                                   ctypes.byref(rc))
    f = fl.open(mine, 'w', application='app', schema='hoomd', schema_version=[1, 4])
    f.configure_device(slab_bytes=32 * 1024, n_slabs=3, n_writers=2)
    for frame in range(120):
        N = int(rng.choice([0, 1, 17, 1023, 1024, 5000, 33333, 60000]))
        pos4 = G.rand_array(rng, (N, 4), np.float64)
        vel4 = G.rand_array(rng, (N, 4), np.float32)
        aux = G.rand_array(rng, (N, 3), np.float32)
        dpos, dvel = dev(pos4), dev(vel4)
        part = np.array([N])
        chunks = [('configuration/step', 4, 1, False, np.array([[frame]], dtype=np.uint64))]
        f.write_chunk('configuration/step', np.array([frame], dtype=np.uint64), write_all=False)
        if frame % 3 == 0:
            f.write_chunk('particles/position', fl.DeviceField.from_tensor(dpos, columns=(0, 3), out_dtype=np.float32),
                          offset=part)
            f.write_chunk('particles/auxiliary1', aux, offset=part)
            f.write_chunk('particles/velocity', fl.DeviceField.from_tensor(dvel, columns=(0, 3)), offset=part)
            order = ['particles/position', 'particles/auxiliary1', 'particles/velocity']
        else:
            f.write_chunks([('particles/velocity', fl.DeviceField.from_tensor(dvel, columns=(0, 3))),
                            ('particles/position', fl.DeviceField.from_tensor(dpos, columns=(0, 3), out_dtype=np.float32))],
                           offset=part)
            f.write_chunk('particles/auxiliary1', aux, offset=part)
            order = ['particles/velocity', 'particles/position', 'particles/auxiliary1']
        data = {'particles/position': (9, 3, pos4[:, :3].astype(np.float32)),
                'particles/velocity': (9, 3, np.ascontiguousarray(vel4[:, :3])),
                'particles/auxiliary1': (9, 3, aux)}
        for name in order:
            t, M, arr = data[name]
            chunks.append((name, t, M, True, np.ascontiguousarray(arr)))
        f.end_frame()
        for name, t, M, all_, arr in chunks:
            n = arr.shape[0]
            assert S.oracle_write_chunk(lib, h, name, t, [arr], M, n, M, [0], [n * M], all_) == 0
        assert lib.oracle_end_frame(h) == 0
    st = f.device_stats()
    assert st['written_bytes'] == st['pack_bytes_out']
    f.close()
    assert lib.oracle_close(h) == 0
    with open(mine, 'rb') as a, open(ref, 'rb') as b:
        assert a.read() == b.read()


def test_async_end_frame_keeps_layout_and_snapshots_values(tmp_path):
    """end_frame(wait=False): the simulation only waits for the pack kernels, the copies and file
    writes run behind it.  The source arrays are overwritten in place right after wait_packed(), so
    the file must hold the values as they were when each frame was packed; 60 frames x 3 chunks also
    cross an on-disk index relocation, where the call degrades to the synchronous path.  The file
    must equal the oracle's (same layout as with synchronous end_frame)."""
    import pgsd.fl as fl
    N = 40_000
    rng = np.random.default_rng(77)
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    lib = S.oracle_lib()
    rc = ctypes.c_int(0)
    h = lib.oracle_create_and_open(ref.encode(), 1, b'app', b'hoomd', lib.oracle_make_version(1, 4), 1, 0,
                                   ctypes.byref(rc))
    f = fl.open(mine, 'w', application='app', schema='hoomd', schema_version=[1, 4])
    f.configure_device(slab_bytes=64 * 1024, n_slabs=4)
    dpos = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
    dvel = torch.zeros((N, 4), dtype=torch.float32, device="cuda")
    for frame in range(60):
        pos4 = G.rand_array(rng, (N, 4), np.float32)
        vel4 = G.rand_array(rng, (N, 4), np.float32)
        dpos.copy_(torch.from_numpy(pos4), non_blocking=False)      # the "simulation" overwrites in place
        dvel.copy_(torch.from_numpy(vel4), non_blocking=False)
        step = np.array([frame], dtype=np.uint64)
        f.write_chunk('configuration/step', step, write_all=False)
        f.write_chunks([('particles/position', fl.DeviceField.from_tensor(dpos, columns=(0, 3))),
                        ('particles/velocity', fl.DeviceField.from_tensor(dvel, columns=(0, 3)))],
                       offset=np.array([N]))
        f.end_frame(wait=False)
        f.wait_packed()                                              # now the arrays may change again
        for name, t, M, all_, arr in (('configuration/step', 4, 1, False, step.reshape(1, 1)),
                                      ('particles/position', 9, 3, True, np.ascontiguousarray(pos4[:, :3])),
                                      ('particles/velocity', 9, 3, True, np.ascontiguousarray(vel4[:, :3]))):
            n = arr.shape[0]
            assert S.oracle_write_chunk(lib, h, name, t, [arr], M, n, M, [0], [n * M], all_) == 0
        assert lib.oracle_end_frame(h) == 0
    f.frame_sync()
    assert f.nframes == 60
    np.testing.assert_array_equal(f.read_chunk(59, 'particles/velocity'), vel4[:, :3])
    f.close()
    assert lib.oracle_close(h) == 0
    with open(mine, 'rb') as a, open(ref, 'rb') as b:
        assert a.read() == b.read()


def test_reading_back_right_after_an_asynchronous_frame(tmp_gsd):
    """No frame_sync between end_frame(wait=False) and the read: the read itself waits for the
    bytes that are still on their way (host read and device read)."""
    import pgsd.fl as fl
    N = 600_000
    pos = torch.randn((N, 4), device="cuda")
    with fl.open(tmp_gsd, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
        f.configure_device(slab_bytes=256 * 1024, n_slabs=2)          # a slow, narrow pipeline
        for i in range(3):
            f.write_chunks([('particles/position', fl.DeviceField.from_tensor(pos, columns=(0, 3)))],
                           offset=np.array([N]))
            f.end_frame(wait=False)
        got = f.read_chunk(2, 'particles/position')
        assert torch.equal(torch.from_numpy(got), pos[:, :3].cpu())
        f.write_chunks([('particles/position', fl.DeviceField.from_tensor(pos, columns=(0, 3)))], offset=np.array([N]))
        f.end_frame(wait=False)
        back = f.read_chunk_device(3, 'particles/position')
        assert torch.equal(back, pos[:, :3].contiguous())


def test_handles_give_back_their_device_and_pinned_memory(tmp_path):
    """open -> device write -> device read -> close, many times: staging arenas, pinned rings,
    streams and events are released with the handle."""
    import resource
    import pgsd.fl as fl
    N = 200_000
    pos = torch.randn((N, 4), device="cuda")

    def cycle(i):
        path = str(tmp_path / ("h%d.gsd" % (i % 3)))
        with fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
            f.write_chunks([('particles/position', fl.DeviceField.from_tensor(pos, columns=(0, 3)))], offset=np.array([N]))
            f.end_frame()
        with fl.open(path, 'r') as f:
            f.read_chunk_device(0, 'particles/position')

    for i in range(3):
        cycle(i)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    for i in range(40):
        cycle(i)
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
    assert free0 - free1 < 64 << 20, (free0, free1)              # one handle holds a 256 MiB arena
    assert (rss1 - rss0) * 1024 < 512 << 20, (rss0, rss1)        # ... and 256 + 128 MiB of pinned rings; 40 leaked ones would be 15 GiB


@pytest.mark.parametrize("slab_bytes,n_slabs,writers", [(1000, 1, 1), (4096, 2, 3), (12345, 3, 2), (1 << 30, 1, 1)])
def test_odd_pipeline_geometries_write_the_same_bytes(slab_bytes, n_slabs, writers, tmp_path):
    """slab sizes that divide nothing, a single slab, more writers than slabs, one giant slab"""
    import pgsd.fl as fl
    N = 20_011
    rng = np.random.default_rng(slab_bytes % 977)
    pos4 = G.rand_array(rng, (N, 4), np.float32)
    img = rng.integers(-9, 9, size=(N, 3)).astype(np.int32)
    paths = []
    for cfg in (None, (slab_bytes, n_slabs, writers)):
        path = str(tmp_path / ("g%d.gsd" % len(paths)))
        with fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
            if cfg:
                f.configure_device(slab_bytes=cfg[0], n_slabs=cfg[1], n_writers=cfg[2])
            for i in range(2):
                f.write_chunks([('particles/position', fl.DeviceField.from_tensor(dev(pos4), columns=(0, 3))),
                                ('particles/image', dev(img))], offset=np.array([N]))
                f.end_frame()
        paths.append(path)
    with open(paths[0], 'rb') as a, open(paths[1], 'rb') as b:
        assert a.read() == b.read()
    with fl.open(paths[1], 'r') as f:
        if slab_bytes < (1 << 20):
            f.configure_device(slab_bytes=slab_bytes, n_slabs=n_slabs)
        np.testing.assert_array_equal(f.read_chunk_device(1, 'particles/image').cpu().numpy(), img)


@pytest.mark.parametrize("batched", [False, True])
def test_stage_now_write_later(batched, tmp_path):
    """pgsd_stage_chunks_device / pgsd_write_staged_chunks: the fused pack is launched at once, the chunks take their
    places in the frame later, in whatever order and grouping the caller writes them -- with host chunks in between
    -- and a staged chunk that is never written is dropped at end_frame.  The file equals the oracle's for the
    sequence actually written."""
    import pgsd.fl as fl
    from test_gpu_file import _oracle_frames
    rng = np.random.default_rng(21)
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    f = fl.open(mine, 'w', application='app', schema='hoomd', schema_version=[1, 4])
    f.frame_exchange = batched
    frames = []
    for frame, N in enumerate([1000, 150_000, 0, 7]):
        pos4 = G.rand_array(rng, (N, 4), np.float32)
        vel4 = G.rand_array(rng, (N, 4), np.float64)
        img = rng.integers(-3, 4, size=(N, 3)).astype(np.int32)
        dpos, dvel, dimg = dev(pos4), dev(vel4), dev(img)
        ticket = f.stage_chunks([("particles/position", fl.DeviceField.from_tensor(dpos, columns=(0, 3))),
                                 ("particles/velocity", fl.DeviceField.from_tensor(dvel, columns=(0, 3), out_dtype=np.float32)),
                                 ("particles/image", dimg),
                                 ("particles/never_written", fl.DeviceField.from_tensor(dpos, columns=(3, 4)))])
        step = np.array([frame], dtype=np.uint64)
        host = G.rand_array(rng, (N, 2), np.float32)
        f.write_chunk("configuration/step", step, write_all=False)
        f.write_staged(ticket, 1, 2, offset=np.array([N]))              # velocity, image
        f.write_chunk("particles/host_field", host, offset=np.array([N]))
        f.write_staged(ticket, 0, 1, offset=np.array([N]))              # position, after them
        with pytest.raises(RuntimeError):
            f.write_staged(ticket, 0, 1, offset=np.array([N]))          # already written
        f.end_frame()                                                   # drops `never_written`
        with pytest.raises(RuntimeError):
            f.write_staged(ticket, 3, 1, offset=np.array([N]))          # the ticket died with the frame
        frames.append([("configuration/step", 4, 1, False, [step.reshape(1, 1)]),
                       ("particles/velocity", 9, 3, True, [G.oracle_pack(vel4, 3, out_dtype=np.float32)]),
                       ("particles/image", 7, 3, True, [img]),
                       ("particles/host_field", 9, 2, True, [host]),
                       ("particles/position", 9, 3, True, [G.oracle_pack(pos4, 3)])])
    f.close()
    _oracle_frames(ref, 1, frames)
    with open(mine, 'rb') as a, open(ref, 'rb') as b:
        assert a.read() == b.read()
    g = fl.open(mine, 'r')
    assert not g.chunk_exists(0, "particles/never_written") and g.nframes == 4
    g.close()


@pytest.mark.parametrize("coalesce", ["1", "0"])
def test_direct_chunks_in_one_pwritev_or_one_by_one_are_the_same_file(tmp_path, monkeypatch, coalesce):
    """Small-frame (direct) path: neighbouring chunks leave in one pwritev (default) or one pwrite each
    (PGSD_DIRECT_COALESCE=0); odd sizes so that the pieces are not contiguous in the pinned arena,
    several frames, sealed synchronously and asynchronously.  Reference: the same chunks from host arrays."""
    import pgsd.fl as fl
    monkeypatch.setenv("PGSD_DIRECT_COALESCE", coalesce)
    monkeypatch.setenv("PGSD_NO_PARKING", "1")           # a parked pipeline keeps the setting it was made with
    rng = np.random.default_rng(5)
    N = 1237
    arrays = [rng.standard_normal((N, 4)).astype(np.float32) for _ in range(9)]
    ids = rng.integers(0, 1 << 30, size=N).astype(np.uint32)
    a, b = str(tmp_path / "dev.gsd"), str(tmp_path / "host.gsd")
    with fl.open(a, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
        f.frame_exchange = True
        for frame in range(4):
            fields = [('particles/a%d' % k, fl.DeviceField.from_tensor(dev(x), columns=(0, 1 + k % 3)))
                      for k, x in enumerate(arrays)]
            fields.insert(4, ('particles/typeid', fl.DeviceField.from_tensor(dev(ids.view(np.int32)), out_dtype=np.uint32)))
            f.write_chunk('configuration/step', np.array([frame], dtype=np.uint64), write_all=False)
            f.write_chunks(fields, offset='auto')
            f.end_frame(wait=(frame % 2 == 0))
        f.frame_sync()
    with fl.open(b, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
        for frame in range(4):
            f.write_chunk('configuration/step', np.array([frame], dtype=np.uint64), write_all=False)
            for k, x in enumerate(arrays):
                if k == 4:
                    f.write_chunk('particles/typeid', ids)
                f.write_chunk('particles/a%d' % k, np.ascontiguousarray(x[:, :1 + k % 3]))
            f.end_frame()
    with open(a, 'rb') as fa, open(b, 'rb') as fb:
        assert fa.read() == fb.read()


def test_a_pending_device_read_keeps_its_staging_while_chunks_are_staged(tmp_path):
    """between `read_chunk_device(wait=False)` and `wait_read` a small read's bytes sit in the pinned
    arena of the direct path (no outstanding piece is counted for them); a stage() issued in between used to recycle
    that arena and pack over them before the unpack kernel had read them."""
    import pgsd.fl as fl
    rng = np.random.default_rng(5)
    N = 3000
    pos = rng.standard_normal((N, 3)).astype(np.float32)
    other = rng.standard_normal((N, 4)).astype(np.float32)
    path = str(tmp_path / "t.gsd")
    with fl.open(path, "w", application="a", schema="hoomd", schema_version=[1, 4]) as f:
        f.write_chunk("particles/position", pos, offset=np.array([N]))
        f.end_frame()
        f.frame_exchange = True
        dother = torch.from_numpy(other).cuda()
        for _ in range(4):
            got = f.read_chunk_device(0, "particles/position", wait=False)          # 36 KB: the direct road
            t = f.stage_chunks([("particles/velocity", fl.DeviceField.from_tensor(dother, columns=(0, 3)))])
            f.wait_packed()
            f.wait_read()
            assert got.cpu().numpy().tobytes() == pos.tobytes()
            f.write_staged(t, 0, 1, offset=np.array([N]))
            f.end_frame()
    with fl.open(path, "r") as f:
        assert f.nframes == 5 and f.read_chunk(3, "particles/velocity").tobytes() == np.ascontiguousarray(other[:, :3]).tobytes()


def test_parked_pipeline_resources_can_be_released(tmp_path):
    """A closed handle of the default geometry parks its streams, pinned slabs and staging for the next handle (at most
    two sets); `pgsd_device_release_parked` gives them back and the next handle builds its own again."""
    import pgsd.fl as fl
    from pgsd import _lib
    pos = torch.rand((5000, 4), device="cuda")

    def one_file(k):
        with fl.open(str(tmp_path / ("p%d.gsd" % k)), "w", application="a", schema="hoomd", schema_version=[1, 4]) as f:
            f.write_chunks([("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3)))], offset=np.array([5000]))
            f.end_frame()
    _lib.lib.pgsd_device_release_parked()
    one_file(0)
    assert _lib.lib.pgsd_device_release_parked() == 1
    assert _lib.lib.pgsd_device_release_parked() == 0
    one_file(1)
    one_file(2)                                      # adopts the set file 1 parked and parks it again
    assert _lib.lib.pgsd_device_release_parked() == 1
    with fl.open(str(tmp_path / "p2.gsd"), "r") as f:
        assert f.read_chunk(0, "particles/position").tobytes() == pos[:, :3].contiguous().cpu().numpy().tobytes()


class _BareDeviceArray(object):
    """What a GPU simulation that is not PyTorch hands over: an object that describes its device memory through
    ``__cuda_array_interface__`` and nothing else (HOOMD's GPU snapshot arrays, CuPy, Numba)."""

    def __init__(self, t, readonly=False):
        self._keep = t
        self.__cuda_array_interface__ = {
            "shape": tuple(t.shape), "typestr": np.dtype(str(t.dtype)[6:]).str, "data": (t.data_ptr(), readonly),
            "version": 3, "strides": None if t.is_contiguous() else tuple(s * t.element_size() for s in t.stride())}


def test_arrays_that_only_speak_the_cuda_array_interface(tmp_path):
    """`write_chunk`, `write_chunks`, `DeviceField.from_device_array` and `HOOMDTrajectory.append` take device arrays
    from any producer that implements ``__cuda_array_interface__``: dense arrays, the xyz columns of a Scalar4 array,
    strided rows, a gather index.  Same files as from torch tensors / host arrays."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    rng = np.random.default_rng(11)
    N = 4321
    pos4 = rng.standard_normal((N, 4)).astype(np.float32)
    tid = rng.integers(0, 5, size=N).astype(np.int32)
    dens = rng.standard_normal(N).astype(np.float64)
    order = rng.permutation(N).astype(np.int32)
    dpos4, dtid, ddens, dorder = dev(pos4), dev(tid), dev(dens), dev(order)
    a, b = str(tmp_path / "bare.gsd"), str(tmp_path / "torch.gsd")
    for path, wrap in ((a, _BareDeviceArray), (b, lambda t: t)):
        with fl.open(path, "w", application="a", schema="hoomd", schema_version=[1, 4]) as f:
            f.write_chunk("particles/typeid", wrap(dtid), offset=np.array([N]))
            f.write_chunks([("particles/position", fl.DeviceField.from_device_array(wrap(dpos4), columns=(0, 3))),
                            ("particles/charge", fl.DeviceField.from_device_array(wrap(dpos4), columns=(3, 4))),
                            ("particles/density", fl.DeviceField.from_device_array(wrap(ddens), out_dtype=np.float32))],
                           offset=np.array([N]))
            f.write_chunk("particles/velocity", fl.DeviceField.from_device_array(wrap(dpos4[:, 1:4]), order=wrap(dorder)),
                          offset=np.array([N]))
            f.end_frame()
    with open(a, "rb") as fa, open(b, "rb") as fb:
        assert fa.read() == fb.read()
    with fl.open(a, "r") as f:
        assert f.read_chunk(0, "particles/position").tobytes() == np.ascontiguousarray(pos4[:, :3]).tobytes()
        assert f.read_chunk(0, "particles/velocity").tobytes() == np.ascontiguousarray(pos4[order][:, 1:4]).tobytes()
        assert f.read_chunk(0, "particles/density").tobytes() == dens.astype(np.float32).tobytes()
    # through pgsd.hoomd, elision included: a bare device array, a host array -- the same trajectory file
    c, d = str(tmp_path / "h_bare.gsd"), str(tmp_path / "h_host.gsd")
    pos = np.ascontiguousarray(pos4[:, :3])
    dpos = dev(pos)
    for path, put in ((c, lambda h, t: _BareDeviceArray(t)), (d, lambda h, t: h)):
        with hoomd.open(path, "w") as t:
            for k in range(3):
                fr = hoomd.Frame()
                fr.configuration.step = k
                fr.particles.N = N
                fr.particles.position = put(pos, dpos)
                fr.particles.typeid = put(tid.view(np.uint32), dtid)
                t.append(fr)
    with open(c, "rb") as fc, open(d, "rb") as fd:
        assert fc.read() == fd.read()
    with pytest.raises(ValueError):
        fl.DeviceField.from_device_array(object())


def test_preallocated_staging_meets_no_allocation_during_the_run(tmp_path):
    """configure_device(prealloc_mib=...): the staging blocks and the whole ring of pinned slabs are allocated by the
    call, so a run of asynchronously sealed frames that pile up behind the file allocates nothing more (an allocation
    in the middle of a run costs a simulation 0.03 ... 38 ms and stalls its streams: DESIGN section 8); the file is the
    one the default, lazily growing pipeline writes."""
    import pgsd.fl as fl
    from pgsd import _lib
    N, frames = 2_000_000, 10                         # 56 MB of staging per frame: 560 MB pile up, 3 blocks of 256 MiB
    g = torch.Generator(device="cuda").manual_seed(77)
    pos = torch.rand((N, 4), device="cuda", generator=g)
    vel = torch.rand((N, 4), device="cuda", generator=g)

    def run(path, prealloc):
        _lib.lib.pgsd_device_release_parked()
        torch.cuda.synchronize()
        free = [torch.cuda.mem_get_info()[0]]
        with fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
            f.frame_exchange = True
            if prealloc:
                f.configure_device(prealloc_mib=prealloc)
            free.append(torch.cuda.mem_get_info()[0])
            for k in range(frames):
                f.write_chunk("configuration/step", np.array([k], dtype=np.uint64), write_all=False)
                f.write_chunks([('particles/position', fl.DeviceField.from_tensor(pos, columns=(0, 3))),
                                ('particles/velocity', fl.DeviceField.from_tensor(vel, columns=(0, 3))),
                                ('particles/mass', fl.DeviceField.from_tensor(vel, columns=(3, 4)))], offset=np.array([N]))
                f.end_frame(wait=False)
            free.append(torch.cuda.mem_get_info()[0])
            f.frame_sync()
        return free

    lazy = run(str(tmp_path / "lazy.gsd"), 0)
    pre = run(str(tmp_path / "pre.gsd"), 768)
    assert pre[0] - pre[1] >= 768 << 20, pre          # allocated by the call ...
    assert pre[1] - pre[2] < 8 << 20, pre             # ... and nothing during the ten frames
    assert lazy[0] - lazy[1] < 8 << 20, lazy          # the default: nothing before the first frame needs it
    with open(str(tmp_path / "lazy.gsd"), "rb") as a, open(str(tmp_path / "pre.gsd"), "rb") as b:
        assert a.read() == b.read()
    _lib.lib.pgsd_device_release_parked()


def test_a_preallocation_the_device_cannot_give_fails_the_call_and_nothing_else(tmp_path):
    """configure_device(prealloc_mib) beyond the device's memory: the call fails loudly (PGSD_ERROR_DEVICE, the runtime's
    message), what it had allocated is given back, and the handle works again once it is configured with something the
    device has."""
    import pgsd.fl as fl
    from pgsd import _lib
    _lib.lib.pgsd_device_release_parked()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    N = 10_000
    pos = torch.rand((N, 4), device="cuda")
    path = str(tmp_path / "p.gsd")
    with fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
        with pytest.raises(RuntimeError):
            f.configure_device(prealloc_mib=(free0 >> 20) + 4096)
        assert "hipMalloc" in _lib.last_error()
        torch.cuda.synchronize()
        assert free0 - torch.cuda.mem_get_info()[0] < 64 << 20          # the blocks it got before the one it did not
        f.configure_device(prealloc_mib=256)
        f.write_chunks([('particles/position', fl.DeviceField.from_tensor(pos, columns=(0, 3)))], offset=np.array([N]))
        f.end_frame()
    with fl.open(path, 'r') as f:
        assert f.read_chunk(0, 'particles/position').tobytes() == pos[:, :3].contiguous().cpu().numpy().tobytes()
    _lib.lib.pgsd_device_release_parked()


def test_a_failed_hip_call_of_the_caller_is_not_mistaken_for_a_failed_launch(tmp_path):
    """The runtime keeps the last error of a thread until somebody reads it.  A caller whose own hipMalloc has just
    failed (and who has not asked hipGetLastError) writes a snapshot: the library's launch checks must not find the
    caller's error and report a kernel launch that never failed."""
    import pgsd.fl as fl
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    N = 50_000
    pos = torch.rand((N, 4), device="cuda")
    flags = (torch.arange(N, device="cuda") % 3 != 0).to(torch.uint8)
    path = str(tmp_path / "p.gsd")
    with fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
        for k in range(2):
            p = ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(p), 1 << 60) != 0          # out of memory; the error stays in the slot
            if k == 0:
                f.write_chunks([('particles/position', fl.DeviceField.from_tensor(pos, columns=(0, 3)))], offset=np.array([N]))
                f.end_frame()
            else:
                idx, n = fl.select_rows(flags)
                assert n == int(flags.sum().item())
    with fl.open(path, 'r') as f:
        assert hip.hipMalloc(ctypes.byref(ctypes.c_void_p()), 1 << 60) != 0
        got = f.read_chunk_device(0, 'particles/position')              # the unpack launch
        assert got.cpu().numpy().tobytes() == pos[:, :3].contiguous().cpu().numpy().tobytes()


def test_reconfiguring_the_pipeline_in_the_middle_of_a_run(tmp_path):
    """pgsd_device_configure on a handle with asynchronously sealed frames still on their way: the old pipeline is
    drained and replaced (other ring geometry, preallocated staging), the frames written before, between and after are
    the ones a handle that was never reconfigured writes."""
    import pgsd.fl as fl
    N = 300_000
    g = torch.Generator(device="cuda").manual_seed(5)
    arrays = [torch.rand((N, 4), device="cuda", generator=g) for _ in range(8)]

    def frame(f, k, wait):
        f.write_chunk("configuration/step", np.array([k], dtype=np.uint64), write_all=False)
        f.write_chunks([('particles/position', fl.DeviceField.from_tensor(arrays[k], columns=(0, 3))),
                        ('particles/mass', fl.DeviceField.from_tensor(arrays[k], columns=(3, 4)))], offset=np.array([N]))
        f.end_frame(wait=wait)

    paths = [str(tmp_path / "plain.gsd"), str(tmp_path / "reconfigured.gsd")]
    for path, reconfigure in zip(paths, (False, True)):
        with fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
            f.frame_exchange = True
            for k in range(3):
                frame(f, k, wait=False)
            if reconfigure:
                f.configure_device(slab_bytes=1 << 20, n_slabs=3)
            for k in range(3, 6):
                frame(f, k, wait=False)
            if reconfigure:
                f.configure_device(prealloc_mib=256)
            for k in range(6, 8):
                frame(f, k, wait=(k == 7))
    with open(paths[0], "rb") as a, open(paths[1], "rb") as b:
        assert a.read() == b.read()
    with fl.open(paths[1], "r") as f:
        assert f.nframes == 8
        for k in (0, 4, 7):
            assert f.read_chunk(k, "particles/position").tobytes() == arrays[k][:, :3].contiguous().cpu().numpy().tobytes()


def test_a_small_staging_cap_makes_the_producer_wait_and_changes_nothing_else(tmp_path):
    """PGSD_STAGING_CAP_MIB: asynchronously sealed frames may hold that much HBM on their way to the file; a producer
    that runs ahead waits for the backlog when it is reached (a subprocess: the variable is read when the pipeline is
    made).  The file is the one a synchronous writer produces."""
    import subprocess
    import sys
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import pgsd.fl as fl
N = 1_000_000
g = torch.Generator(device="cuda").manual_seed(9)
pos = torch.rand((N, 4), device="cuda", generator=g)
for path, wait in ((sys.argv[1], False), (sys.argv[2], True)):
    with fl.open(path, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
        f.frame_exchange = True
        for k in range(40):                       # 16 MB of staging per frame: 640 MB against a cap of 128 MiB
            pos[:, 0] = float(k)
            f.write_chunks([('particles/position', fl.DeviceField.from_tensor(pos, columns=(0, 3))),
                            ('particles/mass', fl.DeviceField.from_tensor(pos, columns=(3, 4)))], offset=np.array([N]))
            f.end_frame(wait=wait)
            if not wait:
                f.wait_packed()
print("free", torch.cuda.mem_get_info()[0] >> 20)
""" % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pgsd-sph_amd")
    a, b = str(tmp_path / "async.gsd"), str(tmp_path / "sync.gsd")
    p = subprocess.run([sys.executable, "-c", code, a, b], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PGSD_STAGING_CAP_MIB="128", PGSD_NO_PARKING="1"))
    assert p.returncode == 0, p.stderr[-2000:]
    with open(a, "rb") as fa, open(b, "rb") as fb:
        assert fa.read() == fb.read()
