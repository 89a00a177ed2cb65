"""The behaviours the reference's own test/test_hoomd.py asks of a hoomd-schema trajectory, restated for
`pgsd.hoomd` (Frame instead of upstream GSD's Snapshot; the PGSD-SPH particle schema; no bonds/angles/...:
PGSD's Frame keeps configuration, particles, constraints, state and log -- hoomd.py:424-467).

One test per reference test (test_hoomd.py:13-790), each over the reference's open-mode matrix
(test/conftest.py:9-10: write 'w' / read 'r', write 'x' / read 'a', write 'a' / read 'r').  The reference's
file cannot run against the reference itself (its writer is a sketch, hoomd.py:568, and the tests still
import upstream `gsd`), so it serves as the specification of the API surface SURVEY 8(b) lists."""
import collections
import pickle

import numpy as np
import pytest

import pgsd.hoomd as hoomd

Mode = collections.namedtuple("Mode", "read write")


@pytest.fixture(params=[Mode("r", "w"), Mode("a", "x"), Mode("r", "a")], ids=lambda m: "(%s,%s)" % (m.read, m.write))
def open_mode(request):
    return request.param


def stepped(i):
    f = hoomd.Frame()
    f.configuration.step = i + 1
    return f


def rich_frame():
    """Every property away from its default (the reference's make_nondefault_snapshot, SPH edition)."""
    f = hoomd.Frame()
    f.configuration.step = 10000
    f.configuration.dimensions = 3
    f.configuration.box = [4, 5, 6, 0.1, 0.2, 0.3]
    f.particles.N = 2
    f.particles.types = ['p', 'q', 'r']
    f.particles.type_shapes = [{"type": "Sphere", "diameter": 2.0}, {}, {"type": "Ellipsoid", "a": 7.0, "b": 5.0, "c": 3.0}]
    f.particles.typeid = [2, 1]
    f.particles.mass = [2, 3]
    f.particles.body = [10, 20]
    f.particles.position = [[0.1, 0.2, 0.3], [-1.0, -2.0, -3.0]]
    f.particles.velocity = [[1.1, 2.2, 3.3], [-3.3, -2.2, -1.1]]
    f.particles.slength = [0.25, 0.5]
    f.particles.density = [1000, 1001]
    f.particles.pressure = [7, 8]
    f.particles.energy = [-1, -2]
    for k in range(1, 5):
        setattr(f.particles, "auxiliary%d" % k, [[k, 0, 0], [0, k, 0]])
    f.particles.image = [[10, 20, 30], [5, 6, 7]]
    f.constraints.N = 1
    f.constraints.value = [1.1]
    f.constraints.group = [[0, 1]]
    f.log['value'] = [1, 2, 4, 10, 12, 18, 22]
    return f


PER_PARTICLE = ('typeid', 'mass', 'body', 'position', 'velocity', 'slength', 'density', 'pressure', 'energy',
                'auxiliary1', 'auxiliary2', 'auxiliary3', 'auxiliary4', 'image')


def assert_frames_equal(got, want, check_step=True):
    want.validate()
    if check_step:
        assert got.configuration.step == want.configuration.step
    assert got.configuration.dimensions == want.configuration.dimensions
    np.testing.assert_array_equal(got.configuration.box, want.configuration.box)
    assert got.particles.N == want.particles.N
    assert got.particles.types == want.particles.types
    assert got.particles.type_shapes == want.particles.type_shapes
    for name in PER_PARTICLE:
        a, b = getattr(got.particles, name), getattr(want.particles, name)
        assert a.dtype == b.dtype and a.shape == b.shape, name
        np.testing.assert_array_equal(a, b, err_msg=name)
    assert got.constraints.N == want.constraints.N
    np.testing.assert_array_equal(got.constraints.value, want.constraints.value)
    np.testing.assert_array_equal(got.constraints.group, want.constraints.group)


def test_create(tmp_path):
    with hoomd.open(name=tmp_path / "c.gsd", mode='w') as hf:
        assert hf.file.schema == 'hoomd'
        assert hf.file.schema_version >= (1, 0)


def test_append(tmp_path, open_mode):
    f = hoomd.Frame()
    f.particles.N = 10
    with hoomd.open(name=tmp_path / "a.gsd", mode=open_mode.write) as hf:
        for i in range(5):
            f.configuration.step = i + 1
            hf.append(f)
    with hoomd.open(name=tmp_path / "a.gsd", mode=open_mode.read) as hf:
        assert len(hf) == 5
        assert [s.configuration.step for s in hf] == [1, 2, 3, 4, 5]
        assert hf[3].particles.N == 10


def test_extend(tmp_path, open_mode):
    with hoomd.open(name=tmp_path / "e.gsd", mode=open_mode.write) as hf:
        hf.extend(stepped(i) for i in range(5))
    with hoomd.open(name=tmp_path / "e.gsd", mode=open_mode.read) as hf:
        assert len(hf) == 5


def test_defaults(tmp_path, open_mode):
    """Nothing but the counts set: every property reads back as its documented default (hoomd.py:167-184,
    :57-62, :381-385), per-particle ones broadcast to N rows."""
    f = hoomd.Frame()
    f.particles.N = 2
    f.constraints.N = 4
    with hoomd.open(name=tmp_path / "d.gsd", mode=open_mode.write) as hf:
        hf.append(f)
    with hoomd.open(name=tmp_path / "d.gsd", mode=open_mode.read) as hf:
        s = hf[0]
    assert s.configuration.step == 0 and s.configuration.dimensions == 3
    np.testing.assert_array_equal(s.configuration.box, np.array([1, 1, 1, 0, 0, 0], dtype=np.float32))
    assert s.particles.N == 2 and s.particles.types == ['A'] and s.particles.type_shapes == [{}]
    expect = dict(typeid=np.zeros(2, np.uint32), mass=np.ones(2, np.float32), body=np.full(2, -1, np.int32),
                  position=np.zeros((2, 3), np.float32), velocity=np.zeros((2, 3), np.float32),
                  slength=np.ones(2, np.float32), density=np.zeros(2, np.float32), pressure=np.zeros(2, np.float32),
                  energy=np.zeros(2, np.float32), image=np.zeros((2, 3), np.int32))
    for k in range(1, 5):
        expect["auxiliary%d" % k] = np.zeros((2, 3), np.float32)
    for name, want in expect.items():
        got = getattr(s.particles, name)
        assert got.dtype == want.dtype and got.shape == want.shape, name
        np.testing.assert_array_equal(got, want, err_msg=name)
    assert s.constraints.N == 4
    np.testing.assert_array_equal(s.constraints.value, np.zeros(4, np.float32))
    np.testing.assert_array_equal(s.constraints.group, np.zeros((4, 2), np.int32))
    assert len(s.state) == 0 and len(s.log) == 0


def test_fallback(tmp_path, open_mode):
    """A frame whose particle count differs from frame 0's takes defaults, not frame 0's arrays
    (hoomd.py:858-884); types and type_shapes still come from frame 0 unless written."""
    f0 = rich_frame()
    f1 = hoomd.Frame()
    f1.particles.N = 2
    f1.particles.position = [[-2, -1, 0], [1, 3.0, 0.5]]
    f1.constraints.N = None
    f2 = hoomd.Frame()
    f2.particles.N = 3
    f2.particles.types = ['q', 's']
    f2.particles.type_shapes = [{}, {"type": "Ellipsoid", "a": 7.0, "b": 5.0, "c": 3.0}]
    f2.constraints.N = 4
    with hoomd.open(name=tmp_path / "f.gsd", mode=open_mode.write) as hf:
        hf.extend([f0, f1, f2])
    with hoomd.open(name=tmp_path / "f.gsd", mode=open_mode.read) as hf:
        assert len(hf) == 3
        assert_frames_equal(hf[0], f0)
        assert 'value' in hf[0].log
        np.testing.assert_array_equal(hf[0].log['value'], f0.log['value'])

        s = hf[1]                    # same N as frame 0: only position differs
        assert s.particles.N == 2 and s.particles.types == f0.particles.types
        np.testing.assert_array_equal(s.particles.position, np.asarray(f1.particles.position, np.float32))
        for name in PER_PARTICLE:
            if name != 'position':
                np.testing.assert_array_equal(getattr(s.particles, name), getattr(f0.particles, name), err_msg=name)
        assert s.constraints.N == 1
        np.testing.assert_array_equal(s.constraints.group, f0.constraints.group)
        np.testing.assert_array_equal(s.log['value'], f0.log['value'])

        s = hf[2]                    # N changed: defaults
        assert s.particles.N == 3 and s.particles.types == ['q', 's']
        assert s.particles.type_shapes == f2.particles.type_shapes
        np.testing.assert_array_equal(s.particles.typeid, np.zeros(3, np.uint32))
        np.testing.assert_array_equal(s.particles.mass, np.ones(3, np.float32))
        np.testing.assert_array_equal(s.particles.body, np.full(3, -1, np.int32))
        np.testing.assert_array_equal(s.particles.position, np.zeros((3, 3), np.float32))
        np.testing.assert_array_equal(s.particles.density, np.zeros(3, np.float32))
        np.testing.assert_array_equal(s.particles.image, np.zeros((3, 3), np.int32))
        assert s.constraints.N == 4
        np.testing.assert_array_equal(s.constraints.value, np.zeros(4, np.float32))
        np.testing.assert_array_equal(s.constraints.group, np.zeros((4, 2), np.int32))


def test_fallback_to_frame0(tmp_path, open_mode):
    """Counts left at None: everything, counts included, is frame 0's."""
    f0 = rich_frame()
    f1 = hoomd.Frame()
    f1.configuration.step = 200000
    f1.particles.N = None
    f1.constraints.N = None
    with hoomd.open(name=tmp_path / "f0.gsd", mode=open_mode.write) as hf:
        hf.extend([f0, f1])
    with hoomd.open(name=tmp_path / "f0.gsd", mode=open_mode.read) as hf:
        assert len(hf) == 2
        s = hf[1]
        assert s.configuration.step == 200000
        assert_frames_equal(s, f0, check_step=False)
        assert 'value' in s.log
        np.testing.assert_array_equal(s.log['value'], f0.log['value'])


def test_no_fallback(tmp_path, open_mode):
    """Defaults written on purpose in a later frame are stored (they differ from frame 0) and win over it."""
    f0 = rich_frame()
    n = f0.particles.N
    f1 = hoomd.Frame()
    f1.configuration.step = 200000
    f1.configuration.dimensions = 3
    f1.configuration.box = [1, 1, 1, 0, 0, 0]
    f1.particles.N = n
    f1.particles.types = ['A']
    f1.particles.type_shapes = [{}]
    f1.particles.typeid = [0] * n
    f1.particles.mass = [1.0] * n
    f1.particles.body = [-1] * n
    f1.particles.slength = [1.0] * n
    for name in ('density', 'pressure', 'energy'):
        setattr(f1.particles, name, [0.0] * n)
    for name in ('position', 'velocity', 'auxiliary1', 'auxiliary2', 'auxiliary3', 'auxiliary4', 'image'):
        setattr(f1.particles, name, [[0, 0, 0]] * n)
    f1.constraints.N = f0.constraints.N
    f1.constraints.value = [0] * f0.constraints.N
    f1.constraints.group = [[0, 0]] * f0.constraints.N
    with hoomd.open(name=tmp_path / "nf.gsd", mode=open_mode.write) as hf:
        hf.extend([f0, f1])
    with hoomd.open(name=tmp_path / "nf.gsd", mode=open_mode.read) as hf:
        assert len(hf) == 2
        assert_frames_equal(hf[1], f1)


def test_iteration(tmp_path, open_mode):
    with hoomd.open(name=tmp_path / "i.gsd", mode=open_mode.write) as hf:
        hf.extend(stepped(i) for i in range(20))
    with hoomd.open(name=tmp_path / "i.gsd", mode=open_mode.read) as hf:
        assert hf[-1].configuration.step == 20
        assert hf[-2].configuration.step == 19
        assert hf[-3].configuration.step == 18
        assert hf[0].configuration.step == 1
        assert hf[-20].configuration.step == 1
        with pytest.raises(IndexError):
            hf[-21]
        with pytest.raises(IndexError):
            hf[20]
        assert [s.configuration.step for s in hf[5:10]] == [6, 7, 8, 9, 10]
        assert [s.configuration.step for s in hf[15:50]] == [16, 17, 18, 19, 20]
        assert [s.configuration.step for s in hf[15:-3]] == [16, 17]


def _check_sequence_protocol(seq, n):
    """len / iter / repeated passes / explicit iterators, as test_hoomd.py:574-677 spells them out."""
    assert len(seq) == n
    assert len(seq[:10]) == min(10, n)
    assert len(iter(seq)) == len(seq)
    assert len(iter(seq[:10])) == len(seq[:10])
    for _ in range(2):                                   # no iterator exhaustion between passes
        assert len(list(seq)) == len(seq)
        assert len(list(seq[:10])) == len(seq[:10])
    it = iter(seq)
    assert len(it) == len(seq)
    assert len(list(it)) == len(seq)
    assert len(list(it)) == len(seq)                     # an iterator hands out fresh passes
    it = iter(seq[:10])
    assert len(it) == 10 and len(list(it)) == 10 and len(list(it)) == 10
    with pytest.raises(IndexError):
        seq[len(seq)]
    assert seq[len(seq) - 1].configuration.step == seq[-1].configuration.step
    it = iter(seq)                                       # and plain next() walks the frames in order
    assert next(it).configuration.step == seq[0].configuration.step
    assert next(it).configuration.step == seq[1].configuration.step


def test_slicing_and_iteration(tmp_path, open_mode):
    with hoomd.open(name=tmp_path / "s.gsd", mode=open_mode.write) as hf:
        hf.extend(stepped(i) for i in range(20))
    with hoomd.open(name=tmp_path / "s.gsd", mode=open_mode.read) as hf:
        _check_sequence_protocol(hf, 20)


def test_view_slicing_and_iteration(tmp_path, open_mode):
    with hoomd.open(name=tmp_path / "v.gsd", mode=open_mode.write) as hf:
        hf.extend(stepped(i) for i in range(40))
    with hoomd.open(name=tmp_path / "v.gsd", mode=open_mode.read) as hf:
        view = hf[::2]
        _check_sequence_protocol(view, 20)
        assert len(view[::2]) == 10
        for _ in range(2):
            assert len(list(view[::2])) == 10
        assert [s.configuration.step for s in view[::2]][:3] == [1, 5, 9]
        assert view[3].configuration.step == 7


def test_state(tmp_path, open_mode):
    f0 = hoomd.Frame()
    f0.state['hpmc/sphere/radius'] = [2.0]
    f0.state['hpmc/sphere/orientable'] = [1]
    f1 = hoomd.Frame()
    f1.state['hpmc/convex_polyhedron/N'] = [3]
    f1.state['hpmc/convex_polyhedron/vertices'] = [[-1, -1, -1], [0, 1, 1], [1, 0, 0]]
    with hoomd.open(name=tmp_path / "st.gsd", mode=open_mode.write) as hf:
        hf.extend([f0, f1])
    with hoomd.open(name=tmp_path / "st.gsd", mode=open_mode.read) as hf:
        assert len(hf) == 2
        s = hf[0]
        np.testing.assert_array_equal(s.state['hpmc/sphere/radius'], f0.state['hpmc/sphere/radius'])
        np.testing.assert_array_equal(s.state['hpmc/sphere/orientable'], f0.state['hpmc/sphere/orientable'])
        s = hf[1]
        np.testing.assert_array_equal(s.state['hpmc/convex_polyhedron/N'], f1.state['hpmc/convex_polyhedron/N'])
        np.testing.assert_array_equal(s.state['hpmc/convex_polyhedron/vertices'],
                                      f1.state['hpmc/convex_polyhedron/vertices'])


def test_log(tmp_path, open_mode):
    f0 = hoomd.Frame()
    f0.log['particles/net_force'] = [[1, 2, 3], [4, 5, 6]]
    f0.log['particles/pair_lj_energy'] = [0, -5, -8, -3]
    f0.log['value/potential_energy'] = [10]
    f0.log['value/pressure'] = [-3]
    f1 = hoomd.Frame()
    f1.log['particles/pair_lj_energy'] = [1, 2, -4, -10]
    f1.log['value/pressure'] = [5]
    with hoomd.open(name=tmp_path / "l.gsd", mode=open_mode.write) as hf:
        hf.extend([f0, f1])
    with hoomd.open(name=tmp_path / "l.gsd", mode=open_mode.read) as hf:
        assert len(hf) == 2
        s = hf[0]
        for k in f0.log:
            np.testing.assert_array_equal(s.log[k], f0.log[k], err_msg=k)
        s = hf[1]
        # entries a frame does not give come from frame 0 (hoomd.py:892-900), the others are its own
        np.testing.assert_array_equal(s.log['particles/net_force'], f0.log['particles/net_force'])
        np.testing.assert_array_equal(s.log['value/potential_energy'], f0.log['value/potential_energy'])
        np.testing.assert_array_equal(s.log['particles/pair_lj_energy'], f1.log['particles/pair_lj_energy'])
        np.testing.assert_array_equal(s.log['value/pressure'], f1.log['value/pressure'])


def test_pickle(tmp_path, open_mode):
    with hoomd.open(name=tmp_path / "p.gsd", mode=open_mode.write) as traj:
        traj.extend(stepped(i) for i in range(20))
        with pytest.raises(pickle.PickleError):
            pickle.dumps(traj)
    if open_mode.read != 'r':
        return          # fl.pyx:971-978: only read-only files pickle; the reference's matrix includes 'a' here,
                        # which its own PGSDFile.__reduce__ refuses
    with hoomd.open(name=tmp_path / "p.gsd", mode=open_mode.read) as traj:
        pkl = pickle.dumps(traj)
        with pickle.loads(pkl) as hf:
            assert len(hf) == 20
            assert hf[7].configuration.step == 8


def test_no_duplicate_types(tmp_path):
    with hoomd.open(name=tmp_path / "dup.gsd", mode='w') as hf:
        f = hoomd.Frame()
        f.particles.types = ['A', 'B', 'B', 'C']
        with pytest.raises(ValueError):
            hf.append(f)
        assert len(hf) == 0
