"""pgsd.hoomd: the schema layer (host path, one rank).  Cases follow what the reference's
test_hoomd.py specifies for the parts of the schema PGSD keeps (configuration, particles incl.
the SPH fields, constraints, log); cited lines are /root/reference/pgsd/pgsd/hoomd.py."""
import numpy as np
import pytest

import pgsd.fl as fl
import pgsd.hoomd as hoomd
import pgsd.pypgsd as pypgsd


def make_frame(i, N=7):
    f = hoomd.Frame()
    f.configuration.step = 100 * i
    f.configuration.box = [10, 11, 12, 0.1, 0.2, 0.3]
    f.particles.N = N
    f.particles.types = ['fluid', 'wall']
    f.particles.typeid = np.arange(N) % 2
    f.particles.position = np.arange(3 * N).reshape(N, 3) + i
    f.particles.velocity = np.ones((N, 3)) * i
    f.particles.mass = np.full(N, 2.5)
    f.particles.slength = np.full(N, 0.1)
    f.particles.density = np.arange(N) * 0.5 + i
    f.particles.pressure = np.arange(N) * 2.0
    f.particles.energy = np.arange(N) * 3.0
    f.particles.auxiliary1 = np.full((N, 3), i)
    f.particles.image = np.zeros((N, 3)) + (i % 2)
    f.particles.type_shapes = [{'type': 'Sphere', 'diameter': 1.0}, {}]
    f.log['kinetic'] = np.array([0.5 * i])
    f.log['matrix'] = np.arange(4, dtype=np.float64).reshape(2, 2) * i
    return f


def test_append_and_read_back(tmp_gsd):
    with hoomd.open(tmp_gsd, 'w') as t:
        t.extend(make_frame(i) for i in range(4))
        assert len(t) == 4
    for reader in (lambda: hoomd.open(tmp_gsd, 'r'),
                   lambda: hoomd.HOOMDTrajectory(pypgsd.PGSDFile(open(tmp_gsd, 'rb')))):
        with reader() as t:
            assert len(t) == 4
            assert t.file.schema == 'hoomd' and t.file.schema_version == (1, 4)
            assert t.file.application.startswith('pgsd.hoomd')
            for i in (0, 3, 1):
                s = t[i]
                ref = make_frame(i)
                ref.validate()
                assert s.configuration.step == 100 * i
                assert s.configuration.dimensions == 3
                np.testing.assert_array_equal(s.configuration.box, ref.configuration.box)
                assert s.particles.N == 7
                assert s.particles.types == ['fluid', 'wall']
                assert s.particles.type_shapes == [{'type': 'Sphere', 'diameter': 1.0}, {}]
                for name in ('typeid', 'position', 'velocity', 'mass', 'slength', 'density', 'pressure',
                             'energy', 'auxiliary1', 'image'):
                    got, exp = getattr(s.particles, name), getattr(ref.particles, name)
                    assert got.dtype == exp.dtype, name
                    np.testing.assert_array_equal(got, exp, err_msg=name)
                # never written -> defaults broadcast to N rows, read-only (hoomd.py:872-881)
                np.testing.assert_array_equal(s.particles.body, np.full(7, -1, dtype=np.int32))
                np.testing.assert_array_equal(s.particles.auxiliary3, np.zeros((7, 3), dtype=np.float32))
                assert not s.particles.body.flags.writeable
                np.testing.assert_array_equal(s.log['kinetic'], [0.5 * i])
                np.testing.assert_array_equal(s.log['matrix'], np.arange(4).reshape(2, 2) * i)
            assert [s.configuration.step for s in t[1:3]] == [100, 200]
            assert t[-1].configuration.step == 300
            with pytest.raises(IndexError):
                t[4]
            with pytest.raises(TypeError):
                t['a']


def test_unchanged_chunks_are_elided(tmp_gsd):
    """Chunks equal to frame 0 or to the default are not written again (hoomd.py:654-694)."""
    with hoomd.open(tmp_gsd, 'w') as t:
        for i in range(3):
            f = make_frame(0)
            f.configuration.step = i
            f.particles.position = np.arange(21).reshape(7, 3) + i
            f.particles.body = np.full(7, -1)          # default value: never stored
            t.append(f)
    with fl.open(tmp_gsd, 'r') as f:
        assert f.chunk_exists(0, 'particles/mass') and not f.chunk_exists(1, 'particles/mass')
        assert f.chunk_exists(0, 'particles/types') and not f.chunk_exists(2, 'particles/types')
        assert not f.chunk_exists(0, 'particles/body')
        assert not f.chunk_exists(0, 'configuration/step')   # 0 == default
        assert f.chunk_exists(1, 'configuration/step') and f.chunk_exists(2, 'particles/position')
        assert not f.chunk_exists(1, 'configuration/box')
    with hoomd.open(tmp_gsd, 'r') as t:
        s = t[2]
        assert s.configuration.step == 2
        np.testing.assert_array_equal(s.particles.mass, np.full(7, 2.5, dtype=np.float32))
        np.testing.assert_array_equal(s.particles.position, np.arange(21).reshape(7, 3) + 2)


def test_upstream_hoomd_attributes_round_trip(tmp_gsd):
    """orientation / angmom / charge / diameter / moment_inertia (BASELINE config 4): documented by
    the reference (hoomd.py:133-158) but absent from its reader; written and read here."""
    N = 5
    with hoomd.open(tmp_gsd, 'w') as t:
        f = hoomd.Frame()
        f.particles.N = N
        f.particles.position = np.zeros((N, 3))
        f.particles.orientation = np.tile([0.5, 0.5, 0.5, 0.5], (N, 1))
        f.particles.angmom = np.arange(4 * N).reshape(N, 4)
        f.particles.charge = np.arange(N) - 2.0
        f.particles.diameter = np.full(N, 3.0)
        f.particles.moment_inertia = np.ones((N, 3)) * 7
        t.append(f)
    with hoomd.open(tmp_gsd, 'r') as t:
        s = t[0]
        np.testing.assert_array_equal(s.particles.orientation, np.tile([0.5] * 4, (N, 1)).astype(np.float32))
        np.testing.assert_array_equal(s.particles.angmom, np.arange(4 * N).reshape(N, 4).astype(np.float32))
        np.testing.assert_array_equal(s.particles.charge, (np.arange(N) - 2.0).astype(np.float32))
        np.testing.assert_array_equal(s.particles.diameter, np.full(N, 3.0, dtype=np.float32))
        assert s.particles.moment_inertia.shape == (N, 3)
    # a file that never stored them reads back as None for those attributes
    with hoomd.open(tmp_gsd, 'w') as t:
        t.append(make_frame(0))
    with hoomd.open(tmp_gsd, 'r') as t:
        assert t[0].particles.orientation is None and t[0].particles.charge is None


def test_validation_and_schema_errors(tmp_gsd):
    f = hoomd.Frame()
    f.particles.N = 3
    f.particles.position = np.zeros((4, 3))
    with pytest.raises(ValueError):
        f.validate()
    f = hoomd.Frame()
    f.particles.types = ['A', 'A']
    with pytest.raises(ValueError, match="unique"):
        f.validate()
    with fl.open(tmp_gsd, 'w', application='x', schema='other', schema_version=[1, 0]):
        pass
    with pytest.raises(RuntimeError):
        hoomd.open(tmp_gsd, 'r')
    with fl.open(tmp_gsd, 'w', application='x', schema='hoomd', schema_version=[2, 0]):
        pass
    with pytest.raises(RuntimeError, match="Incompatible"):
        hoomd.HOOMDTrajectory(fl.open(tmp_gsd, 'r'))


def test_two_dimensional_box_sets_dimensions(tmp_gsd):
    with hoomd.open(tmp_gsd, 'w') as t:
        f = hoomd.Frame()
        f.configuration.box = [5, 5, 0, 0, 0, 0]
        assert f.configuration.dimensions == 2
        t.append(f)
    with hoomd.open(tmp_gsd, 'r') as t:
        assert t[0].configuration.dimensions == 2 and t[0].particles.N == 0


def test_constraints_and_read_log(tmp_gsd):
    with hoomd.open(tmp_gsd, 'w') as t:
        for i in range(3):
            f = make_frame(i, N=4)
            f.constraints.N = 2
            f.constraints.value = [1.5, 2.5]
            f.constraints.group = [[0, 1], [2, 3]]
            t.append(f)
    with hoomd.open(tmp_gsd, 'r') as t:
        s = t[1]
        assert s.constraints.N == 2
        np.testing.assert_array_equal(s.constraints.value, np.array([1.5, 2.5], dtype=np.float32))
        np.testing.assert_array_equal(s.constraints.group, [[0, 1], [2, 3]])
    log = hoomd.read_log(tmp_gsd)
    np.testing.assert_array_equal(log['configuration/step'], [0, 100, 200])
    np.testing.assert_array_equal(log['log/kinetic'], [0.0, 0.5, 1.0])
    assert log['log/matrix'].shape == (3, 2, 2)
    assert set(hoomd.read_log(tmp_gsd, scalar_only=True)) == {'configuration/step', 'log/kinetic'}


def test_append_to_existing_trajectory(tmp_gsd):
    with hoomd.open(tmp_gsd, 'w') as t:
        t.append(make_frame(0))
    with hoomd.open(tmp_gsd, 'a') as t:
        assert len(t) == 1
        t.append(make_frame(1))
    with hoomd.open(tmp_gsd, 'r+') as t:
        t.append(make_frame(2))
        assert len(t) == 3
    with hoomd.open(tmp_gsd, 'r') as t:
        assert [s.configuration.step for s in t] == [0, 100, 200]
        np.testing.assert_array_equal(t[2].particles.density, np.arange(7) * 0.5 + 2)


FIELDS = {'typeid': (np.uint32, 1), 'mass': (np.float32, 1), 'body': (np.int32, 1), 'position': (np.float32, 3),
          'velocity': (np.float32, 3), 'slength': (np.float32, 1), 'density': (np.float32, 1),
          'pressure': (np.float32, 1), 'energy': (np.float32, 1), 'auxiliary1': (np.float32, 3),
          'auxiliary4': (np.float32, 3), 'image': (np.int32, 3)}


@pytest.mark.parametrize("seed", range(12))
def test_random_trajectories_read_back_what_was_meant(seed, tmp_gsd):
    """Model check of the elision rules (hoomd.py:654-694) and the reader's fall-backs
    (hoomd.py:724-902): whatever mix of set / unset / default-valued / frame-0-valued attributes
    and particle counts a trajectory has, frame i reads back as: the value set in frame i, else
    frame 0's value when the particle count is the same, else the default."""
    rng = np.random.default_rng(seed)
    defaults = hoomd.ParticleData._default_value
    n_frames = int(rng.integers(2, 7))
    N0 = int(rng.integers(1, 9))
    provided = []
    with hoomd.open(tmp_gsd, 'w') as t:
        for i in range(n_frames):
            N = N0 if rng.random() < 0.7 else int(rng.integers(1, 9))
            fr = hoomd.Frame()
            fr.particles.N = N
            rec = {'N': N}
            if rng.random() < 0.8:
                fr.configuration.step = rec['step'] = int(rng.integers(0, 1000))
            if rng.random() < 0.5:
                fr.configuration.box = rec['box'] = [float(x) for x in rng.integers(1, 9, 3)] + [0, 0, 0]
            for name, (dt, M) in FIELDS.items():
                r = rng.random()
                if r < 0.35:
                    continue                                                    # unset
                shape = (N,) if M == 1 else (N, M)
                if r < 0.5:                                                     # the default value, spelled out
                    val = np.empty(shape, dtype=dt)
                    val[...] = defaults[name]
                elif r < 0.65 and provided and name in provided[0] and provided[0]['N'] == N:
                    val = provided[0][name].copy()                              # same as frame 0
                else:
                    val = rng.integers(0, 5, size=shape).astype(dt)
                setattr(fr.particles, name, val)
                rec[name] = val
            if rng.random() < 0.4:
                fr.log['e'] = rec['log/e'] = np.array([float(rng.integers(0, 100))])
            t.append(fr)
            provided.append(rec)

    def expected(i, name):
        if name in provided[i]:
            return provided[i][name]
        dt, M = FIELDS[name]
        N = provided[i]['N']
        if i > 0 and provided[0]['N'] == N:
            return expected(0, name)
        val = np.empty((N,) if M == 1 else (N, M), dtype=dt)
        val[...] = defaults[name]
        return val

    for reader in (lambda: hoomd.open(tmp_gsd, 'r'),
                   lambda: hoomd.HOOMDTrajectory(pypgsd.PGSDFile(open(tmp_gsd, 'rb')))):
        with reader() as t:
            assert len(t) == n_frames
            for i in list(range(n_frames))[::-1] + [0]:
                s = t[i]
                assert s.particles.N == provided[i]['N'], (seed, i)
                step = provided[i].get('step', provided[0].get('step', 0))
                assert s.configuration.step == step, (seed, i)
                box = provided[i].get('box', provided[0].get('box', [1, 1, 1, 0, 0, 0]))
                np.testing.assert_array_equal(s.configuration.box, np.array(box, dtype=np.float32))
                for name in FIELDS:
                    got, exp = getattr(s.particles, name), expected(i, name)
                    assert got.dtype == exp.dtype and got.shape == exp.shape, (seed, i, name)
                    np.testing.assert_array_equal(got, exp, err_msg="seed %d frame %d %s" % (seed, i, name))
                log = provided[i].get('log/e', provided[0].get('log/e'))
                if log is None:
                    assert 'e' not in s.log
                else:
                    np.testing.assert_array_equal(s.log['e'], log)


def test_state_chunks_round_trip(tmp_gsd):
    """``Frame.state`` (hoomd.py:444,454; the writer half is commented out in the reference, :634-636)"""
    with hoomd.open(tmp_gsd, 'w') as t:
        for i in range(3):
            f = make_frame(i)
            if i != 1:
                f.state['hpmc/sphere/radius'] = np.array([0.5, 0.25 + i], dtype=np.float32)
                f.state['hpmc/integrate/d'] = np.array([0.1 * (i + 1)])
            t.append(f)
    for reader in (lambda: hoomd.open(tmp_gsd, 'r'),
                   lambda: hoomd.HOOMDTrajectory(pypgsd.PGSDFile(open(tmp_gsd, 'rb')))):
        with reader() as t:
            np.testing.assert_array_equal(t[0].state['hpmc/sphere/radius'], np.array([0.5, 0.25], dtype=np.float32))
            assert t[1].state == {}
            np.testing.assert_array_equal(t[2].state['hpmc/sphere/radius'], np.array([0.5, 2.25], dtype=np.float32))
            np.testing.assert_allclose(t[2].state['hpmc/integrate/d'], [0.3])


def test_elision_comparisons_agree_with_numpy():
    """`append` decides what to elide with `_equal` / `_equiv` (typed, cached defaults; first rows first): the
    decisions must be numpy.array_equal's / numpy.array_equiv's (hoomd.py:673-687) whatever the shapes and types."""
    import random
    import numpy
    from pgsd import hoomd as H
    rng = numpy.random.default_rng(11)
    random.seed(11)
    defaults = [0, 1, -1, 1.5, [0, 0, 0], [1, 0, 0, 0], [0, 1], [1, 1, 1], [0, 0, 0, 0, 0, 0]]
    for _ in range(4000):
        shape = tuple(int(x) for x in rng.integers(0, 5, size=rng.integers(0, 3)))
        dt = random.choice([numpy.float32, numpy.float64, numpy.int32, numpy.uint32, numpy.uint8])
        a = rng.integers(0, 2, size=shape).astype(dt)
        b = rng.integers(0, 2, size=shape).astype(dt)
        d = random.choice(defaults)
        assert H._equiv(a, d) == bool(numpy.array_equiv(a, d)), (a, d)
        assert H._equal(a, b) == bool(numpy.array_equal(a, b)), (a, b)
        assert H._equal(a, a.copy())
    big = numpy.zeros((10000, 3), dtype=numpy.float32)
    assert H._equiv(big, [0, 0, 0]) and H._equal(big, big.copy())
    big[9999, 2] = 1
    assert not H._equiv(big, [0, 0, 0]) and not H._equal(big, numpy.zeros_like(big))
    assert not H._equal(big, big[:-1])
    nan = numpy.array([numpy.nan], dtype=numpy.float32)
    assert not H._equal(nan, nan) and not H._equiv(nan, numpy.nan)      # array_equal's NaN semantics are kept


def test_bond_data_is_importable_and_validates_like_the_reference():
    """hoomd.py:273-362: the class exists in the reference although no Frame of it carries topology; code written
    against the reference must import and validate it the same way."""
    b = hoomd.BondData(3)
    assert (b.M, b.N, b.types, b.typeid, b.group) == (3, 0, None, None, None)
    assert list(b._default_value) == ['N', 'types', 'typeid', 'group'] and b._default_value['group'].shape == (3,)
    b.N = 2
    b.typeid = [0, 1]
    b.group = [0, 1, 2, 1, 2, 3]
    b.types = ['A', 'B']
    b.validate()
    assert b.typeid.dtype == np.uint32 and b.typeid.shape == (2,)
    assert b.group.dtype == np.int32 and b.group.shape == (2, 3)
    b.types = ['A', 'A']
    with pytest.raises(ValueError, match="unique"):
        b.validate()
    b.types = ['A']
    b.group = [0, 1, 2]                 # not N x M
    with pytest.raises(ValueError):
        b.validate()
    assert not hasattr(hoomd.Frame(), "bonds")          # as in the reference (hoomd.py:450-456)
