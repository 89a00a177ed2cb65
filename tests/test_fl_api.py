"""Behaviour of the file-layer API (pgsd.fl over the C ABI), host path, single rank.

The cases are the behavioural spec the reference's (un-ported, upstream-GSD) test_fl.py
states for this API, re-expressed against pgsd names and modes; the
cited lines are /root/reference/pgsd/pgsd/test/test_fl.py.
"""
import os
import pickle
import random

import numpy as np
import pytest

import pgsd.fl as fl
import pgsd.pypgsd as pypgsd
import scenario as S

DTYPES = ['uint8', 'uint16', 'uint32', 'uint64', 'int8', 'int16', 'int32', 'int64', 'float32', 'float64']


def create(path, mode='w', app='test_app', schema='none', ver=(1, 2)):
    return fl.open(path, mode, application=app, schema=schema, schema_version=list(ver))


@pytest.mark.parametrize("typ", DTYPES)
def test_dtype_round_trip(tmp_gsd, typ):
    """1-D, 2-D and zero-length arrays of all ten types (test_fl.py:29-88)."""
    data1d = np.array([1, 2, 3, 4, 5, 127], dtype=typ)
    data2d = np.array([[10, 20], [30, 40], [50, 80]], dtype=typ)
    data_zero = np.array([], dtype=typ)
    with create(tmp_gsd) as f:
        f.write_chunk('data1d', data1d)
        f.write_chunk('data2d', data2d)
        f.write_chunk('data_zero', data_zero)
        f.end_frame()
    for reader in (lambda: fl.open(tmp_gsd, 'r'), lambda: pypgsd.PGSDFile(open(tmp_gsd, 'rb'))):
        with reader() as f:
            r1, r2, r0 = f.read_chunk(0, 'data1d'), f.read_chunk(0, 'data2d'), f.read_chunk(0, 'data_zero')
            assert r1.dtype == data1d.dtype and r2.dtype == data2d.dtype
            np.testing.assert_array_equal(r1, data1d)
            np.testing.assert_array_equal(r2, data2d)
            assert r0.shape == (0,) and r0.dtype == data_zero.dtype
            with pytest.raises(KeyError):
                f.read_chunk(0, 'missing')


def test_metadata(tmp_gsd):
    """150 frames, header fields (test_fl.py:91-127)."""
    with create(tmp_gsd, app='test_metadata', schema='none', ver=(1, 2)) as f:
        assert f.mode == 'w'
        for i in range(150):
            f.write_chunk('data', np.array([i], dtype=np.float32))
            f.end_frame()
        assert f.nframes == 150
    with fl.open(tmp_gsd, 'r') as f:
        assert f.name == tmp_gsd
        assert f.mode == 'r'
        assert f.application == 'test_metadata'
        assert f.schema == 'none'
        assert f.schema_version == (1, 2)
        assert f.pgsd_version == (2, 0)
        assert f.nframes == 150
        assert f.nnames == 1
        assert f.read_chunk(149, 'data')[0] == 149.0


def test_append_reopen(tmp_gsd):
    """Re-open for appending, 1024 more frames, index relocations included (test_fl.py:130-176)."""
    with create(tmp_gsd, mode='x') as f:
        pass
    data = np.array([10], dtype=np.int64)
    nframes = 1024
    with fl.open(tmp_gsd, 'a', application='test_app', schema='none', schema_version=[1, 2]) as f:
        assert f.mode == 'a'
        for i in range(nframes):
            data[0] = i
            f.write_chunk('data1', data)
            data[0] = i * 10
            f.write_chunk('data10', data)
            f.end_frame()
    with fl.open(tmp_gsd, 'r') as f:
        assert f.nframes == nframes
        for i in range(0, nframes, 37):
            assert f.read_chunk(i, 'data1')[0] == i
            assert f.read_chunk(i, 'data10')[0] == i * 10
    # the same calls through the oracle give the same bytes
    import ctypes
    lib = S.oracle_lib()
    rc = ctypes.c_int(0)
    ref = tmp_gsd + ".ref"
    h = lib.oracle_create_and_open(ref.encode(), 1, b'test_app', b'none', lib.oracle_make_version(1, 2), 1, 1,
                                   ctypes.byref(rc))
    assert rc.value == 0 and lib.oracle_close(h) == 0
    h = lib.oracle_open(ref.encode(), 1, 1, ctypes.byref(rc))
    for i in range(nframes):
        for name, v in (('data1', i), ('data10', i * 10)):
            a = np.array([[v]], dtype=np.int64)
            assert S.oracle_write_chunk(lib, h, name, 8, [a], 1, 1, 1, [0], [1], True) == 0
        assert lib.oracle_end_frame(h) == 0
    assert lib.oracle_close(h) == 0
    with open(tmp_gsd, 'rb') as a, open(ref, 'rb') as b:
        assert a.read() == b.read()


def test_chunk_exists(tmp_gsd):
    """test_fl.py:179-254."""
    with create(tmp_gsd) as f:
        f.write_chunk('chunk1', np.array([1, 2, 3, 4], dtype=np.float32))
        f.end_frame()
        f.write_chunk('abcdefg', np.array([1, 2, 3, 4], dtype=np.float32))
        f.end_frame()
        f.write_chunk('test', np.array([1, 2, 3, 4], dtype=np.float32))
        f.end_frame()
    for reader in (lambda: fl.open(tmp_gsd, 'r'), lambda: pypgsd.PGSDFile(open(tmp_gsd, 'rb'))):
        with reader() as f:
            expected = {(0, 'chunk1'), (1, 'abcdefg'), (2, 'test')}
            for frame in range(4):
                for name in ('chunk1', 'abcdefg', 'test', 'nope'):
                    assert f.chunk_exists(frame, name) == ((frame, name) in expected)
            assert f.chunk_exists(0, 'chunk1')
            f.read_chunk(0, 'chunk1')


def test_readonly_errors(tmp_gsd):
    """Writing to a read-only file fails; closed files raise ValueError (test_fl.py:257-289)."""
    with create(tmp_gsd) as f:
        for i in range(10):
            f.write_chunk('chunk1', np.array([1, 2, 3, 4], dtype=np.float32))
            f.end_frame()
    f = fl.open(tmp_gsd, 'r')
    with pytest.raises(RuntimeError):
        f.end_frame()
    with pytest.raises(RuntimeError):
        f.write_chunk('chunk1', np.array([1, 2, 3, 4], dtype=np.float32))
    with pytest.raises(RuntimeError):
        f.flush()
    f.close()
    f.close()  # twice is fine
    for call in (lambda: f.end_frame(), lambda: f.nframes, lambda: f.read_chunk(0, 'chunk1'),
                 lambda: f.write_chunk('c', np.zeros(1)), lambda: f.flush(),
                 lambda: f.find_matching_chunk_names('')):
        with pytest.raises(ValueError):
            call()


def test_fileio_errors(tmp_gsd, tmp_path):
    """Missing file -> OSError; a file that is no GSD file -> RuntimeError (test_fl.py:292-311)."""
    with pytest.raises(OSError):
        fl.open('/this/file/does/not/exist', 'r')
    bad = tmp_path / "bad.gsd"
    bad.write_bytes(b'test' * 100)
    with pytest.raises(RuntimeError, match="Not a PGSD file"):
        fl.open(str(bad), 'r')
    with pytest.raises(RuntimeError):
        pypgsd.PGSDFile(open(str(bad), 'rb'))
    empty = tmp_path / "empty.gsd"
    empty.write_bytes(b'')
    with pytest.raises(RuntimeError):
        fl.open(str(empty), 'r')
    with create(tmp_gsd):
        pass
    with pytest.raises(FileExistsError):
        create(tmp_gsd, mode='x')
    with pytest.raises(ValueError):
        fl.open(tmp_gsd, 'wb', application='a', schema='b', schema_version=[1, 0])
    for missing in ('application', 'schema', 'schema_version'):
        kw = dict(application='a', schema='b', schema_version=[1, 0])
        kw[missing] = None
        with pytest.raises(ValueError):
            fl.open(tmp_gsd, 'w', **kw)
    with pytest.raises(RuntimeError, match="incorrect schema"):
        fl.open(tmp_gsd, 'r', schema='other')


def test_dtype_errors(tmp_gsd):
    """Unsupported dtypes and > 2 dimensions (test_fl.py:314-358)."""
    with create(tmp_gsd) as f:
        with pytest.raises(ValueError):
            f.write_chunk('chunk1', np.array([1, 2, 3, 4], dtype=np.bool_))
        with pytest.raises(ValueError):
            f.write_chunk('chunk1', np.array([1, 2, 3, 4], dtype=np.float16))
        with pytest.raises(ValueError):
            f.write_chunk('chunk1', np.array([1, 2, 3, 4], dtype=np.complex64))
        with pytest.raises(ValueError):
            f.write_chunk('chunk1', np.zeros((2, 2, 2), dtype=np.float32))
        f.end_frame()


def test_truncation_and_long_names(tmp_gsd):
    """application / schema are cut to 63 bytes, chunk names are unbounded in v2 files
    (test_fl.py:399-429)."""
    long_app, long_schema = 'a' * 100, 's' * 80
    long_name = 'n' * 200
    with fl.open(tmp_gsd, 'w', application=long_app, schema=long_schema, schema_version=[1, 0]) as f:
        f.write_chunk(long_name, np.array([1, 2, 3], dtype=np.int32))
        f.write_chunk(long_name + 'x', np.array([4], dtype=np.int32))
        f.end_frame()
    with fl.open(tmp_gsd, 'r') as f:
        assert f.application == 'a' * 63
        assert f.schema == 's' * 63
        np.testing.assert_array_equal(f.read_chunk(0, long_name), [1, 2, 3])
        np.testing.assert_array_equal(f.read_chunk(0, long_name + 'x'), [4])
    with fl.open(tmp_gsd, 'r', schema=long_schema):
        pass


def test_find_matching_chunk_names(tmp_gsd):
    """Prefix search, also for names still pending in the open frame (test_fl.py:498-555)."""
    with create(tmp_gsd) as f:
        f.write_chunk('log/A', np.zeros(2, dtype=np.float32))
        f.write_chunk('log/chunk2', np.zeros(2, dtype=np.float32))
        f.end_frame()
        f.write_chunk('data/B', np.zeros(2, dtype=np.float32))
        f.end_frame()
        assert set(f.find_matching_chunk_names('')) == {'log/A', 'log/chunk2', 'data/B'}
    for reader in (lambda: fl.open(tmp_gsd, 'r'), lambda: pypgsd.PGSDFile(open(tmp_gsd, 'rb'))):
        with reader() as f:
            assert f.find_matching_chunk_names('') == ['log/A', 'log/chunk2', 'data/B']
            assert f.find_matching_chunk_names('log/') == ['log/A', 'log/chunk2']
            assert f.find_matching_chunk_names('data/') == ['data/B']
            assert f.find_matching_chunk_names('other/') == []


def test_name_limit(tmp_gsd):
    """65535 distinct names fit, the next one is refused (test_fl.py:558-571)."""
    with create(tmp_gsd) as f:
        one = np.array([1], dtype=np.uint8)
        for i in range(65535):
            f.write_chunk(str(i), one)
        with pytest.raises(RuntimeError, match="namelist is full"):
            f.write_chunk('65535', one)
        f.end_frame()
        assert f.nnames == 65535
    with fl.open(tmp_gsd, 'r') as f:
        assert f.chunk_exists(0, '65534') and not f.chunk_exists(0, '65535')


def test_many_names_shuffled(tmp_gsd):
    """1000 names written in a different order every frame (test_fl.py:574-610)."""
    values = list(range(1000))
    rng = random.Random(5)
    with create(tmp_gsd) as f:
        for frame in range(5):
            rng.shuffle(values)
            for v in values:
                f.write_chunk(str(v), np.array([v * 13 + frame], dtype=np.int32))
            f.end_frame()
    for reader in (lambda: fl.open(tmp_gsd, 'r'), lambda: pypgsd.PGSDFile(open(tmp_gsd, 'rb'))):
        with reader() as f:
            assert f.nframes == 5
            for frame in range(5):
                for v in range(0, 1000, 7):
                    assert f.read_chunk(frame, str(v))[0] == v * 13 + frame


def test_read_reference_v1_file():
    """The reference's own fixture: GSD v1.0, 5 frames x 127 int32 chunks named "0".."126"
    holding 13 * name (test_fl.py:613-651); v1 stores 64-byte names and an index that is only
    ordered by frame."""
    path = os.path.join(S.GOLDEN, 'reference_test_gsd_v1.gsd')
    for reader in (lambda: fl.open(path, 'r'), lambda: pypgsd.PGSDFile(open(path, 'rb'))):
        with reader() as f:
            assert f.pgsd_version == (1, 0)
            assert f.nframes == 5
            for frame in range(5):
                for v in range(127):
                    data = f.read_chunk(frame, str(v))
                    assert data.dtype == np.int32 and data[0] == v * 13
            assert not f.chunk_exists(5, '0') and not f.chunk_exists(0, '127')


@pytest.mark.parametrize("reopen_mode", ["r", "a", "r+"])
def test_write_to_reference_v1_file(tmp_path, reopen_mode):
    """test_fl.py:709-785: a v1 file can be written to and stays v1 -- 256 integer names (129 of them new:
    the name list moves) plus one far too long name, which v1's 64-byte slots cut to 63 bytes.  Checked on
    the writing handle, after reopening, and with the pure-Python reader.  (The layout of such a file is
    pinned to the reference's bytes by tests/golden/scenarios/vone_append.scn.)"""
    import random
    import shutil
    path = str(tmp_path / "v1.gsd")
    shutil.copy(os.path.join(S.GOLDEN, 'reference_test_gsd_v1.gsd'), path)
    long_name = 'abcdefg' * 1000
    values = list(range(256)) + [long_name]
    names = sorted(str(v)[:63] for v in values)

    def payload(v):
        return np.array([v * 13], dtype=np.int32) if isinstance(v, int) else np.array([len(v), -7], dtype=np.int64)

    def check(f):
        assert f.pgsd_version == (1, 0)
        assert sorted(f.find_matching_chunk_names('')) == names
        order = list(values)
        random.Random(7).shuffle(order)
        for v in order:
            got = f.read_chunk(frame=5, name=str(v)[:63])
            assert got.dtype == payload(v).dtype
            np.testing.assert_array_equal(got, payload(v))
        assert f.read_chunk(frame=3, name='126')[0] == 126 * 13        # the old frames are still there
        assert not f.chunk_exists(4, '200') and f.chunk_exists(5, '200')

    with fl.open(path, 'r+') as f:
        assert f.pgsd_version == (1, 0) and f.nframes == 5
        for v in values:
            f.write_chunk(name=str(v), data=payload(v))
        f.end_frame()
        assert f.nframes == 6
        check(f)
    with fl.open(path, reopen_mode) as f:
        check(f)
    with pypgsd.PGSDFile(open(path, 'rb')) as f:
        check(f)


def test_zero_size_chunk_in_middle(tmp_gsd):
    """test_fl.py:863-893."""
    with create(tmp_gsd) as f:
        f.write_chunk('a', np.array([1, 2, 3], dtype=np.uint8))
        f.write_chunk('empty', np.array([], dtype=np.float32))
        f.write_chunk('b', np.array([[4, 5]], dtype=np.float64))
        f.end_frame()
    with fl.open(tmp_gsd, 'r') as f:
        assert f.chunk_exists(0, 'empty')
        assert f.read_chunk(0, 'empty').size == 0
        np.testing.assert_array_equal(f.read_chunk(0, 'b'), [[4, 5]])


def test_utf8_filename(tmp_path):
    """test_fl.py:898-929."""
    path = str(tmp_path / "teilchen_αβγ_粒子.gsd")
    with create(path) as f:
        f.write_chunk('x', np.arange(4, dtype=np.int16))
        f.end_frame()
    with fl.open(path, 'r') as f:
        np.testing.assert_array_equal(f.read_chunk(0, 'x'), np.arange(4))


def test_write_then_read_same_handle(tmp_gsd):
    """A writable handle can read back what it sealed, and keeps writing (test_fl.py:932-963)."""
    with create(tmp_gsd) as f:
        for i in range(3):
            f.write_chunk('d', np.array([i, i + 1], dtype=np.uint32))
            f.end_frame()
            np.testing.assert_array_equal(f.read_chunk(i, 'd'), [i, i + 1])
        f.write_chunk('pending', np.array([9], dtype=np.uint32))
        assert not f.chunk_exists(3, 'pending')   # frame 3 is not sealed yet
        f.end_frame()
        assert f.chunk_exists(3, 'pending')


def test_non_contiguous_and_array_like(tmp_gsd):
    base = np.arange(24, dtype=np.float32).reshape(6, 4)
    with create(tmp_gsd) as f:
        f.write_chunk('cols', base[:, :3])          # implicit contiguous copy, fl.pyx:571-573
        f.write_chunk('list', [[1, 2], [3, 4]])     # array-like -> int64
        f.end_frame()
    with fl.open(tmp_gsd, 'r') as f:
        np.testing.assert_array_equal(f.read_chunk(0, 'cols'), base[:, :3])
        assert f.read_chunk(0, 'list').dtype == np.int64


def test_buffer_properties_and_pickle(tmp_gsd):
    with create(tmp_gsd) as f:
        assert f.maximum_write_buffer_size == 64 * 1024 * 1024
        assert f.index_entries_to_buffer == 256 * 1024
        f.maximum_write_buffer_size = 1024
        f.index_entries_to_buffer = 10
        assert f.maximum_write_buffer_size == 1024 and f.index_entries_to_buffer == 10
        with pytest.raises(RuntimeError):
            f.maximum_write_buffer_size = 0
        with pytest.raises(pickle.PickleError):
            pickle.dumps(f)
        f.write_chunk('x', np.zeros(3))
        f.end_frame()
    with fl.open(tmp_gsd, 'r') as f:
        g = pickle.loads(pickle.dumps(f))
        assert g.nframes == 1 and g.mode == 'r'
        g.close()


def test_partial_row_read(tmp_gsd):
    """read_chunk(N, M, offset, r_all=True) returns a row slab (fl.pyx:717-874, pgsd.c:2498-2508)."""
    data = np.arange(60, dtype=np.float32).reshape(20, 3)
    with create(tmp_gsd) as f:
        f.write_chunk('p', data)
        f.end_frame()
    from pgsd import _lib
    import ctypes
    with fl.open(tmp_gsd, 'r') as f:
        h = f._h()
        e = _lib.lib.pgsd_find_chunk(h, 0, b'p')
        out = np.zeros((5, 3), dtype=np.float32)
        assert _lib.lib.pgsd_read_chunk(h, out.ctypes.data, e, 5, 3, 7, True) == 0
        np.testing.assert_array_equal(out, data[7:12])


def test_async_end_frame_on_host_data_is_the_same_file(tmp_path):
    """end_frame(wait=False) with host arrays only: nothing is in flight, the file equals the one
    written with synchronous frames."""
    a, b = str(tmp_path / "a.gsd"), str(tmp_path / "b.gsd")
    for path, wait in ((a, True), (b, False)):
        with create(path) as f:
            for i in range(50):
                f.write_chunk('x', np.arange(7, dtype=np.float32) + i)
                f.write_chunk('s', np.array([i], dtype=np.uint64), write_all=False)
                f.end_frame(wait=wait)
            f.frame_sync()
    with open(a, 'rb') as fa, open(b, 'rb') as fb:
        assert fa.read() == fb.read()


def test_open_modes(tmp_path):
    """The open() mode table (fl.pyx:301-317; the walk through the modes is test_fl.py:432-495, with the
    reference's own mode names: it has no 'wb'/'xb+' spellings): 'x' creates and refuses an existing file,
    'w' overwrites, 'a' creates a missing file and otherwise appends, 'r+' writes to an existing file, 'r'
    only reads; writable modes can read what they wrote."""
    data = np.array([1, 2, 3, 4, 5, 10012], dtype=np.int64)
    path = str(tmp_path / "modes.gsd")
    kw = dict(application='test_open', schema='none', schema_version=[1, 2])

    def one_frame(f):
        f.write_chunk(name='chunk1', data=data)
        f.end_frame()

    with fl.open(name=path, mode='x', **kw) as f:
        one_frame(f)
        np.testing.assert_array_equal(f.read_chunk(0, name='chunk1'), data)
    with pytest.raises(FileExistsError):
        fl.open(name=path, mode='x', **kw)
    with fl.open(name=path, mode='w', **kw) as f:          # starts over
        assert f.nframes == 0
        one_frame(f)
        f.read_chunk(0, name='chunk1')
    with fl.open(name=path, mode='a', **kw) as f:          # keeps frame 0, adds frame 1
        assert f.nframes == 1
        one_frame(f)
    with fl.open(name=path, mode='r', **kw) as f:
        assert f.nframes == 2
        f.read_chunk(0, name='chunk1')
        f.read_chunk(1, name='chunk1')
        with pytest.raises(RuntimeError):
            f.write_chunk(name='chunk1', data=data)
    with fl.open(name=path, mode='r+', **kw) as f:
        one_frame(f)
        for i in range(3):
            np.testing.assert_array_equal(f.read_chunk(i, name='chunk1'), data)
    other = str(tmp_path / "fresh.gsd")
    with fl.open(name=other, mode='a', **kw) as f:         # 'a' on a missing file creates it
        one_frame(f)
    with fl.open(name=other, mode='r') as f:
        assert f.nframes == 1 and f.application == 'test_open'
    for bad in ('wb', 'xb+', 'rb', 'ab', 'z', ''):
        with pytest.raises(ValueError):
            fl.open(name=path, mode=bad, **kw)
    for mode in ('r', 'r+'):
        with pytest.raises(FileNotFoundError):
            fl.open(name=str(tmp_path / "missing.gsd"), mode=mode, **kw)
    for mode in ('w', 'x'):                                # creating needs the metadata (fl.pyx:327-334)
        with pytest.raises(ValueError):
            fl.open(name=str(tmp_path / "nometa.gsd"), mode=mode)


def test_read_rows_and_local_reads(tmp_gsd):
    """`read_rows`: a row range into an array of that height (pgsd_read_chunk with all == true, pgsd.c:2498-2534);
    `local_reads`: reads on a writable handle that take no part in a collective flush (pgsd_set_local_reads)."""
    rng = np.random.default_rng(3)
    a = rng.standard_normal((1000, 3)).astype(np.float32)
    b = rng.integers(0, 9, size=1000).astype(np.uint32)
    with fl.open(tmp_gsd, 'w', application='app', schema='hoomd', schema_version=[1, 4]) as f:
        f.write_chunk('particles/position', a)
        f.write_chunk('particles/typeid', b)
        f.end_frame()
        f.write_chunk('particles/position', a + 1)
        f.end_frame()
        assert f.local_reads is False
        f.local_reads = True
        assert f.local_reads is True
        got = f.read_rows(0, 'particles/position', 250, 100)
        assert got.shape == (100, 3) and got.tobytes() == a[250:350].tobytes()
        assert f.read_rows(0, 'particles/typeid', 999, 1).tobytes() == b[999:].tobytes()
        assert f.read_rows(1, 'particles/position', 0, 1000).tobytes() == (a + 1).tobytes()
        assert f.read_rows(0, 'particles/typeid', 1000, 0).shape == (0,)
        f.local_reads = False
        assert f.read_rows(0, 'particles/position', 0, 3).tobytes() == a[:3].tobytes()
        with pytest.raises(ValueError):
            f.read_rows(0, 'particles/position', 990, 11)
        with pytest.raises(KeyError):
            f.read_rows(2, 'particles/position', 0, 1)
    with fl.open(tmp_gsd, 'r') as f:
        assert f.read_rows(0, 'particles/position', 1, 2).tobytes() == a[1:3].tobytes()
