"""Files written by this package, read back with the REFERENCE's own pure-Python reader and schema
layer (imported from /root/reference, which exists in the build container only; on the GPU box the
test skips).  The reference's hoomd.py imports mpi4py unconditionally (hoomd.py:29) but never uses it
on the read path, so an empty stand-in module is registered before the import."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

REF = "/root/reference/pgsd/pgsd"

pytestmark = [pytest.mark.ref,
              pytest.mark.skipif(not os.path.exists(os.path.join(REF, "pypgsd.py")),
                                 reason="reference checkout not present")]


def load_reference_module(name):
    """Import the reference's module under a private name (keeps it apart from this repo's `pgsd`)."""
    if "mpi4py" not in sys.modules:
        stub = types.ModuleType("mpi4py")
        stub.MPI = object()
        sys.modules["mpi4py"] = stub
    spec = importlib.util.spec_from_file_location("reference_" + name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_reader_reads_our_hoomd_file(tmp_gsd):
    import pgsd.hoomd as hoomd
    rng = np.random.default_rng(0)
    N = 321
    frames = []
    with hoomd.open(tmp_gsd, 'w') as t:
        for i in range(4):
            f = hoomd.Frame()
            f.configuration.step = 1000 + i
            f.configuration.box = [8, 9, 10, 0, 0.5, 0]
            f.particles.N = N
            f.particles.types = ['water', 'wall']
            f.particles.typeid = rng.integers(0, 2, N)
            f.particles.position = rng.standard_normal((N, 3))
            f.particles.velocity = rng.standard_normal((N, 3))
            f.particles.density = rng.random(N)
            f.particles.auxiliary2 = rng.standard_normal((N, 3))
            f.particles.image = rng.integers(-1, 2, (N, 3))
            f.log['pe'] = np.array([float(i)])
            f.validate()
            frames.append(f)
            t.append(f)
    rp = load_reference_module("pypgsd")
    rh = load_reference_module("hoomd")
    traj = rh.HOOMDTrajectory(rp.PGSDFile(open(tmp_gsd, 'rb')))
    assert len(traj) == 4
    for i in (0, 2, 3):
        s = traj[i]
        exp = frames[i]
        assert s.configuration.step == 1000 + i
        np.testing.assert_array_equal(s.configuration.box, exp.configuration.box)
        assert s.particles.N == N and s.particles.types == ['water', 'wall']
        for name in ('typeid', 'position', 'velocity', 'density', 'auxiliary2', 'image'):
            np.testing.assert_array_equal(getattr(s.particles, name), getattr(exp.particles, name), err_msg=name)
        np.testing.assert_array_equal(s.particles.mass, np.ones(N, dtype=np.float32))      # default fallback
        np.testing.assert_array_equal(s.log['pe'], [float(i)])


def test_reference_reader_reads_multi_rank_golden_like_file(tmp_gsd):
    """A 3-rank file (duplicated replicated chunks, per-rank offsets) written by the product driver."""
    import product
    import scenario as S
    product.run_driver(S.scenario_path("sph_full"), tmp_gsd, 3)
    rp = load_reference_module("pypgsd")
    f = rp.PGSDFile(open(tmp_gsd, 'rb'))
    assert f.nframes == 2 and f.schema == 'hoomd'
    pos = f.read_chunk(1, 'particles/position')
    np.testing.assert_array_equal(pos, S.gen_data(9, 8, 0, 333, 3))
    assert f.read_chunk(0, 'particles/N').dtype == np.uint32
    ours = __import__("pgsd.pypgsd", fromlist=["PGSDFile"]).PGSDFile(open(tmp_gsd, 'rb'))
    for name in f.find_matching_chunk_names(''):
        for frame in range(2):
            assert f.chunk_exists(frame, name) == ours.chunk_exists(frame, name)
            if f.chunk_exists(frame, name):
                np.testing.assert_array_equal(f.read_chunk(frame, name), ours.read_chunk(frame, name))
