"""`HOOMDTrajectory.append` pinned byte for byte to the call sequence the reference sketches.

The reference's writer is disabled (hoomd.py:568); what it is meant to do survives as the
commented-out body (hoomd.py:569-642) and `_should_write` (hoomd.py:654-694).  This file holds an
independent model of that sketch -- chunk order, which chunks are replicated, dtypes, the string
encoding, the elision rules -- replays the resulting `pgsd_write_chunk` calls through the CPU ORACLE
(pinned to reference-written files by tests/test_oracle_golden.py, including the default-argument
call shape of `log/*`: tests/golden/scenarios/defaultargs.scn), and compares the oracle's file with
the file `pgsd.hoomd` writes: P = 1 in this process, P = 2 over a gloo group.

Deliberate deviations from the sketch (documented in `HOOMDTrajectory.append`) are part of the model:
`constraints/*` replicated with the constraint count, decisions agreed over the ranks,
`particles/N` compared as the global count.
"""
import ctypes
import json
import os
import sys

import numpy as np
import pytest

import scenario as S

sys.path.insert(0, os.path.join(S.ROOT, "pgsd-sph_amd"))

# the order of the reference's _default_value tables (hoomd.py:60-63, 167-184, 388-391)
CONFIGURATION = [("step", np.uint64(0)), ("dimensions", np.uint8(3)),
                 ("box", np.array([1, 1, 1, 0, 0, 0], dtype=np.float32))]
PARTICLES = [("N", np.uint32(0)), ("types", ["A"]), ("typeid", np.uint32(0)), ("mass", np.float32(1.0)),
             ("body", np.int32(-1)), ("position", np.zeros(3, np.float32)), ("velocity", np.zeros(3, np.float32)),
             ("slength", np.float32(1.0)), ("density", np.float32(0.0)), ("pressure", np.float32(0.0)),
             ("energy", np.float32(0.0)), ("auxiliary1", np.zeros(3, np.float32)),
             ("auxiliary2", np.zeros(3, np.float32)), ("auxiliary3", np.zeros(3, np.float32)),
             ("auxiliary4", np.zeros(3, np.float32)), ("image", np.zeros(3, np.int32)), ("type_shapes", [{}])]
CONSTRAINTS = [("N", np.uint32(0)), ("value", np.float32(0)), ("group", np.zeros(2, np.int32))]
DTYPES = {"typeid": np.uint32, "mass": np.float32, "body": np.int32, "position": np.float32,
          "velocity": np.float32, "slength": np.float32, "density": np.float32, "pressure": np.float32,
          "energy": np.float32, "auxiliary1": np.float32, "auxiliary2": np.float32, "auxiliary3": np.float32,
          "auxiliary4": np.float32, "image": np.int32, "value": np.float32, "group": np.int32}
REPLICATED = ("box", "N", "step", "dimensions", "types", "type_shapes")      # hoomd.py:604-627


def encode_strings(strings):
    """hoomd.py:628-630"""
    wid = max(len(w) for w in strings) + 1
    b = np.array(strings, dtype=np.dtype((bytes, wid)))
    return b.view(dtype=np.int8).reshape(len(b), wid)


def global_frames():
    """Three frames of GLOBAL data (particle arrays in global row order)."""
    frames = []
    n = 23
    f0 = {
        "configuration": {"step": 100, "dimensions": None, "box": [10, 11, 12, 0.5, 0.25, 0.125]},
        "particles": {"types": ["A", "Bb"], "typeid": S.gen_data(3, 1, 0, n, 1)[:, 0] % 2,
                      "mass": S.gen_data(9, 2, 0, n, 1)[:, 0], "body": np.full(n, -1, np.int32),    # default: elided
                      "position": S.gen_data(9, 3, 0, n, 3), "velocity": S.gen_data(9, 4, 0, n, 3),
                      "slength": S.gen_data(9, 5, 0, n, 1)[:, 0], "density": S.gen_data(9, 6, 0, n, 1)[:, 0],
                      "pressure": np.zeros(n, np.float32),                                           # default: elided
                      "energy": S.gen_data(9, 7, 0, n, 1)[:, 0], "auxiliary1": S.gen_data(9, 8, 0, n, 3),
                      "auxiliary3": S.gen_data(9, 9, 0, n, 3), "image": S.gen_data(7, 10, 0, n, 3) % 5,
                      "type_shapes": [{"type": "Sphere", "diameter": 1.0}, {"type": "Sphere", "diameter": 2.5}]},
        "constraints": {"N": 3, "value": [1.5, 2.5, 3.5], "group": [[0, 1], [2, 3], [4, 5]]},
        "log": {"energy": np.array([1.5, -2.5]), "walltime": np.array([0.125], dtype=np.float32)},
        "state": {},
        "n": n,
    }
    frames.append(f0)
    f1 = {
        "configuration": {"step": 200, "dimensions": 3, "box": [10, 11, 12, 0.5, 0.25, 0.125]},     # box == frame 0
        "particles": {"types": ["A", "Bb"], "typeid": f0["particles"]["typeid"],                      # == frame 0
                      "mass": f0["particles"]["mass"], "body": np.full(n, -1, np.int32),
                      "position": S.gen_data(9, 13, 0, n, 3), "velocity": S.gen_data(9, 14, 0, n, 3),
                      "density": f0["particles"]["density"], "pressure": S.gen_data(9, 16, 0, n, 1)[:, 0],
                      "image": f0["particles"]["image"]},
        "constraints": {"N": 3, "value": [1.5, 2.5, 3.5], "group": [[0, 1], [2, 3], [4, 5]]},        # == frame 0
        "log": {"energy": np.array([3.5, -4.5]), "walltime": np.array([0.25], dtype=np.float32)},
        "state": {"hpmc/integrate/d": np.array([0.1, 0.2])},
        "n": n,
    }
    frames.append(f1)
    m = 17                                                                                           # N changes
    f2 = {
        "configuration": {"step": 300, "dimensions": 2, "box": [10, 11, 0, 0, 0, 0]},
        "particles": {"types": ["A", "Bb", "C"], "typeid": S.gen_data(3, 21, 0, m, 1)[:, 0] % 3,
                      "position": S.gen_data(9, 23, 0, m, 3), "velocity": np.zeros((m, 3), np.float32),
                      "mass": np.ones(m, np.float32)},
        "constraints": {"N": 0},
        "log": {"energy": np.array([5.5, -6.5])},
        "state": {},
        "n": m,
    }
    frames.append(f2)
    return frames


def partition(n, P):
    base = [n // P + (1 if r < n % P else 0) for r in range(P)]
    if P > 1:                      # uneven on purpose
        base[0] += base[-1] // 2
        base[-1] -= base[-1] // 2
    return base


def build_frame(hoomd, g, counts, rank, explicit_part_dist):
    """The pgsd.hoomd Frame rank `rank` appends for global frame `g`."""
    row0 = sum(counts[:rank])
    n = counts[rank]
    fr = hoomd.Frame()
    c = g["configuration"]
    fr.configuration.step = c["step"]
    if c["dimensions"] is not None:
        fr.configuration.dimensions = c["dimensions"]
    fr.configuration.box = c["box"]
    fr.particles.N = n
    for name, value in g["particles"].items():
        if name in ("types", "type_shapes"):
            setattr(fr.particles, name, value)
        else:
            setattr(fr.particles, name, np.asarray(value)[row0:row0 + n])
    for name, value in g["constraints"].items():
        setattr(fr.constraints, name, value)
    fr.log = dict(g["log"])
    fr.state = dict(g["state"])
    if explicit_part_dist:
        fr.part_dist = np.array(counts, dtype=np.uint64)
    return fr


# ------------------------------------------------------------------ the model of the sketch
class Model:
    """Replays hoomd.py:569-642 + 654-694 for P ranks at once and drives the oracle."""

    def __init__(self, path, P):
        self.lib = S.oracle_lib()
        self.P = P
        rc = ctypes.c_int(0)
        self.h = self.lib.oracle_create_and_open(path.encode(), P, b"pgsd.hoomd 3.2.0", b"hoomd",
                                                 self.lib.oracle_make_version(1, 4), 1, 0, ctypes.byref(rc))
        assert rc.value == 0
        self.initial = None          # frame 0 as a READER sees it (global, defaults filled in)
        self.frame0_chunks = set()
        self.part_off = False

    def small(self, name, arr):
        """write_all=False, offset=None: every rank passes the same value (fl.pyx:592-598)"""
        a = np.ascontiguousarray(arr)
        a2 = a.reshape(a.shape[0], 1) if a.ndim == 1 else a
        t = S.TYPE_IDS[{"uint8": "u8", "uint32": "u32", "uint64": "u64", "int8": "i8", "int32": "i32",
                        "float32": "f32", "float64": "f64"}[str(a2.dtype)]]
        M = a2.shape[1]
        assert S.oracle_write_chunk(self.lib, self.h, name, t, [a2] * self.P, M, a2.shape[0], M, [0] * self.P,
                                    [a2.shape[0] * M] * self.P, False) == 0

    def default_args(self, name, arr):
        """write_chunk(name, data): write_all=True, offset=None (fl.pyx:526, 592-598)"""
        a = np.ascontiguousarray(arr)
        a2 = a.reshape(a.shape[0], 1) if a.ndim == 1 else a
        t = S.TYPE_IDS[{"float32": "f32", "float64": "f64", "int32": "i32", "uint32": "u32"}[str(a2.dtype)]]
        M = a2.shape[1]
        assert S.oracle_write_chunk(self.lib, self.h, name, t, [a2] * self.P, M, a2.shape[0], M, [0] * self.P,
                                    [a2.shape[0] * M] * self.P, True) == 0

    def partitioned(self, name, per_rank, counts):
        """write_all=True, offset=part_dist (hoomd.py:597-600, fl.pyx:594-598)"""
        arrs = [a.reshape(a.shape[0], 1) if a.ndim == 1 else a for a in per_rank]
        M = arrs[0].shape[1]
        t = S.TYPE_IDS[{"float32": "f32", "int32": "i32", "uint32": "u32"}[str(arrs[0].dtype)]]
        Ng = sum(counts)
        row0 = [sum(counts[:r]) for r in range(self.P)]
        assert S.oracle_write_chunk(self.lib, self.h, name, t, arrs, M, Ng, M, [x * M for x in row0], [Ng * M] * self.P,
                                    True) == 0

    def should_write(self, path, name, data, default, frame_index, rows=None):
        """hoomd.py:654-694 for ONE rank's value `data` (None = not set).  rows: with several ranks a per-particle
        array is compared with THAT RANK'S rows of frame 0 (`HOOMDTrajectory._host_row_votes`) -- while the
        partition is frame 0's; against the whole of frame 0, as the sketch compares, it could never be equal."""
        if data is None:
            return False
        if self.initial is not None:
            init = self.initial[path].get(name)
            if rows is not None and init is not None:
                init = init[rows] if (not self.part_off and (path + "/" + name) in self.frame0_chunks) else None
            if init is not None and np.array_equal(init, data):
                return False
        if name in ("types", "type_shapes"):
            matches_default = data == default
        else:
            matches_default = np.array_equiv(data, default)
        if matches_default and (path + "/" + name) not in self.frame0_chunks:
            return False
        return True

    def append(self, g, counts, frame_index, device=False):
        """device=True: the per-particle attributes are GPU-resident.  They are decided by the same rule as host
        arrays (`HOOMDTrajectory._device_votes`: numpy's equality on the GPU, against this rank's rows of frame 0 or
        against the default value), so the model does not tell the two apart."""
        P = self.P
        if frame_index == 0:
            self.counts0, self.part_off = list(counts), False
        elif list(counts) != self.counts0:
            self.part_off = True            # particles moved between ranks / their number changed: frame 0's rows are
                                            # other particles' from now on (several ranks: no more comparisons)
        n_global = sum(counts)
        row0 = [sum(counts[:r]) for r in range(P)]
        written = []
        for path, table in (("configuration", CONFIGURATION), ("particles", PARTICLES), ("constraints", CONSTRAINTS)):
            for name, default in table:
                # each rank's local view of the value
                local = []
                for r in range(P):
                    v = g[path].get(name)
                    if path == "particles" and name == "N":
                        v = n_global
                    elif path == "configuration" and name == "dimensions" and v is None:
                        v = 2 if g["configuration"]["box"][2] == 0 else 3          # hoomd.py:89-99
                    elif path == "particles" and v is not None and name not in ("types", "type_shapes"):
                        v = np.ascontiguousarray(np.asarray(v)[row0[r]:row0[r] + counts[r]], dtype=DTYPES[name])
                    elif path == "constraints" and v is not None and name != "N":
                        v = np.ascontiguousarray(v, dtype=DTYPES[name])
                    elif path == "configuration" and name == "box":
                        v = np.ascontiguousarray(v, dtype=np.float32)
                    local.append(v)
                if not any(self.should_write(path, name, local[r], default, frame_index,
                                               slice(row0[r], row0[r] + counts[r])
                                               if P > 1 and path == "particles" and name not in REPLICATED else None)
                             for r in range(P)):
                    continue
                chunk = path + "/" + name
                written.append(chunk)
                if path == "particles" and name not in REPLICATED:
                    self.partitioned(chunk, local, counts)
                elif name == "N":
                    count = n_global if path == "particles" else int(g["constraints"]["N"])
                    self.small(chunk, np.array([count], dtype=np.uint32))
                elif name == "step":
                    self.small(chunk, np.array([local[0]], dtype=np.uint64))
                elif name == "dimensions":
                    self.small(chunk, np.array([local[0]], dtype=np.uint8))
                elif name == "types":
                    self.small(chunk, encode_strings(local[0]))
                elif name == "type_shapes":
                    self.small(chunk, encode_strings([json.dumps(d) for d in local[0]]))
                else:   # box, constraints/value, constraints/group
                    self.small(chunk, local[0])
        for name, data in g["state"].items():
            self.default_args("state/" + name, np.ascontiguousarray(data))
            written.append("state/" + name)
        for name, data in g["log"].items():
            self.default_args("log/" + name, data)
            written.append("log/" + name)
        assert self.lib.oracle_end_frame(self.h) == 0
        if frame_index == 0:
            self.frame0_chunks = set(written)
            # frame 0 as the reader returns it: what was written, defaults (broadcast to N rows) elsewhere
            self.initial = {"configuration": {}, "particles": {}, "constraints": {}}
            for name, default in CONFIGURATION:
                v = g["configuration"].get(name)
                if name == "dimensions" and v is None:
                    v = 2 if g["configuration"]["box"][2] == 0 else 3
                if name == "box":
                    v = np.ascontiguousarray(v, dtype=np.float32)
                self.initial["configuration"][name] = v if "configuration/" + name in written else default
            for path, table, n_rows in (("particles", PARTICLES, n_global),
                                        ("constraints", CONSTRAINTS, int(g["constraints"].get("N", 0)))):
                for name, default in table:
                    v = g[path].get(name)
                    if name == "N":
                        self.initial[path][name] = n_rows if path + "/N" in written else 0
                    elif name in ("types", "type_shapes"):
                        self.initial[path][name] = v if path + "/" + name in written else default
                    elif path + "/" + name in written:
                        self.initial[path][name] = np.ascontiguousarray(v, dtype=DTYPES[name])
                    else:
                        d = np.asarray(default)
                        self.initial[path][name] = np.broadcast_to(d, (n_rows,) + d.shape).copy()
        return written

    def close(self):
        assert self.lib.oracle_close(self.h) == 0


def expected_file(path, P, device=False, frames=None):
    m = Model(path, P)
    chunks = []
    for k, g in enumerate(frames if frames is not None else global_frames()):
        chunks.append(m.append(g, g.get("counts") or partition(g["n"], P), k, device=device))
    m.close()
    return chunks


def test_model_elides_what_the_sketch_elides(tmp_path):
    """The model itself: defaults skipped in frame 0, frame-0 matches skipped later, new N written."""
    chunks = expected_file(str(tmp_path / "o.gsd"), 1)
    assert "particles/body" not in chunks[0] and "particles/pressure" not in chunks[0]
    assert "configuration/dimensions" not in chunks[0]          # 3 is the default
    assert chunks[0][:3] == ["configuration/step", "configuration/box", "particles/N"]
    assert chunks[0][-2:] == ["log/energy", "log/walltime"]
    assert "particles/typeid" not in chunks[1] and "configuration/box" not in chunks[1]
    assert "particles/pressure" in chunks[1] and "constraints/value" not in chunks[1]
    assert "state/hpmc/integrate/d" in chunks[1]
    assert "particles/N" in chunks[2] and "configuration/dimensions" in chunks[2] and "constraints/N" in chunks[2]


def test_append_single_rank_matches_oracle_file(tmp_path):
    import pgsd.hoomd as hoomd
    ref, mine = str(tmp_path / "ref.gsd"), str(tmp_path / "mine.gsd")
    expected_file(ref, 1)
    with hoomd.open(mine, "w") as t:
        for g in global_frames():
            t.append(build_frame(hoomd, g, [g["n"]], 0, False))
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()


def _worker(rank, world, port, path):
    sys.path.insert(0, os.path.join(S.ROOT, "pgsd-sph_amd"))
    sys.path.insert(0, os.path.join(S.ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pgsd.dist as pdist
    import pgsd.hoomd as hoomd
    import test_hoomd_append_oracle as me
    assert pdist.init_from_torch() == "torch-gloo"
    t = hoomd.open(path, "w")
    for k, g in enumerate(me.global_frames()):
        # frame 1 hands over part_dist itself, the others let append() gather it
        t.append(me.build_frame(hoomd, g, me.partition(g["n"], world), rank, explicit_part_dist=(k == 1)))
    t.close()
    pdist.finalize()
    dist.destroy_process_group()


@pytest.mark.parametrize("P", [2, 3])
def test_append_multi_rank_matches_oracle_file(P, tmp_path):
    torch = pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from test_multirank import free_port
    ref, mine = str(tmp_path / "ref.gsd"), str(tmp_path / "mine.gsd")
    expected_file(ref, P)
    mp.spawn(_worker, args=(P, free_port(), mine), nprocs=P, join=True)
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()
    # and the reader gives the global frames back
    import pgsd.hoomd as hoomd
    with hoomd.open(mine, "r") as t:
        frames = global_frames()
        assert len(t) == 3
        np.testing.assert_array_equal(t[1].particles.position, frames[1]["particles"]["position"])
        np.testing.assert_array_equal(t[1].particles.typeid, frames[0]["particles"]["typeid"])
        assert t[2].particles.N == 17 and t[2].particles.types == ["A", "Bb", "C"]
        np.testing.assert_array_equal(t[1].log["energy"], [3.5, -4.5])


def _lopsided_worker(rank, world, port, path):
    sys.path.insert(0, os.path.join(S.ROOT, "pgsd-sph_amd"))
    sys.path.insert(0, os.path.join(S.ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pgsd.dist as pdist
    import pgsd.hoomd as hoomd
    assert pdist.init_from_torch() == "torch-gloo"
    counts = [0, 6]                         # rank 0 holds no particles: it has no rows of frame 0 to read
    t = hoomd.open(path, "w")
    for k in range(4):
        fr = hoomd.Frame()
        fr.configuration.step = k
        fr.particles.N = counts[rank]
        fr.particles.position = np.full((counts[rank], 3), k + 1, np.float32)
        fr.particles.density = np.full(counts[rank], 7.0 if k != 2 else 8.0, np.float32)
        t.append(fr, wait=(k % 2 == 0))
    t.close()
    pdist.finalize()
    dist.destroy_process_group()


def test_a_rank_without_rows_is_not_waited_for_by_the_others_reads(tmp_path):
    """Rank 1 compares its arrays with its rows of frame 0 and reads them from the file when it first does; rank 0
    holds no particles and reads nothing.  Those reads are LOCAL (`PGSDFile.local_reads`): as the collective flush a
    read on a writable handle is by default (pgsd.c:2436-2537), rank 1 would wait in one rank 0 never enters."""
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from test_multirank import free_port
    import pgsd.hoomd as hoomd
    path = str(tmp_path / "t.gsd")
    mp.spawn(_lopsided_worker, args=(2, free_port(), path), nprocs=2, join=True)
    with hoomd.open(path, "r") as t:
        f = t.file
        assert [f.chunk_exists(k, "particles/density") for k in range(4)] == [True, False, True, False]
        assert all(f.chunk_exists(k, "particles/position") for k in range(4))
        np.testing.assert_array_equal(t[2].particles.density, [8] * 6)
        np.testing.assert_array_equal(t[3].particles.density, [7] * 6)
        np.testing.assert_array_equal(t[3].particles.position, np.full((6, 3), 4, np.float32))


# ------------------------------------------------------------------ random trajectories on several ranks
def random_frames(seed, P):
    """Global frames whose per-particle arrays never change, always change or change once; the partition changes once
    in some trajectories (the same particles, other counts per rank: the comparisons with frame 0 end there)."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(P, 40))
    nframes = int(rng.integers(3, 6))
    names = {"position": (np.float32, 3), "velocity": (np.float32, 3), "typeid": (np.uint32, 1), "mass": (np.float32, 1),
             "density": (np.float32, 1), "image": (np.int32, 3)}
    picked = [nm for nm in names if rng.random() < 0.8] or ["position"]
    behaviour = {nm: rng.choice(["static", "moving", "once"]) for nm in picked}
    once_at = {nm: int(rng.integers(1, nframes)) for nm in picked}
    repartition_at = int(rng.integers(1, nframes)) if rng.random() < 0.4 else None

    def values(nm, k):
        dt, M = names[nm]
        shape = (n, M) if M > 1 else (n,)
        v = rng.integers(1, 50, size=shape) + 100 * k       # never a default, never an earlier frame's values
        return v.astype(dt)

    base = {nm: values(nm, 0) for nm in picked}
    changed = {}
    counts = partition(n, P)
    frames = []
    for k in range(nframes):
        if repartition_at is not None and k == repartition_at:
            counts = list(reversed(counts)) if counts != list(reversed(counts)) else [counts[0] - 1] + counts[1:-1] + [counts[-1] + 1]
        particles = {}
        for nm in picked:
            b = behaviour[nm]
            if k == 0 or b == "static":
                particles[nm] = changed.get(nm, base[nm])
            elif b == "moving":
                particles[nm] = values(nm, k)
            else:
                if k == once_at[nm]:
                    changed[nm] = values(nm, k)
                particles[nm] = changed.get(nm, base[nm])
        frames.append({"configuration": {"step": k, "dimensions": None, "box": [5, 5, 5, 0, 0, 0]},
                       "particles": particles, "constraints": {}, "log": {}, "state": {}, "n": n,
                       "counts": list(counts), "explicit": bool(rng.random() < 0.5)})
    return frames


def _random_worker(rank, world, port, path, seed):
    sys.path.insert(0, os.path.join(S.ROOT, "pgsd-sph_amd"))
    sys.path.insert(0, os.path.join(S.ROOT, "tests"))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import pgsd.dist as pdist
    import pgsd.hoomd as hoomd
    import test_hoomd_append_oracle as me
    assert pdist.init_from_torch() == "torch-gloo"
    t = hoomd.open(path, "w")
    for k, g in enumerate(me.random_frames(seed, world)):
        t.append(me.build_frame(hoomd, g, g["counts"], rank, explicit_part_dist=g["explicit"]), wait=(k % 2 == 0))
    t.close()
    pdist.finalize()
    dist.destroy_process_group()


@pytest.mark.parametrize("seed,P", [(1, 2), (3, 3), (4, 3), (6, 3), (8, 2), (9, 3), (13, 2), (23, 3)])
def test_random_multi_rank_trajectories_match_the_model(seed, P, tmp_path):
    """Host arrays on several ranks: static arrays are elided as on one rank (each rank against its own rows of frame
    0), until the partition changes; the file is the model's, byte for byte."""
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    from test_multirank import free_port
    ref, mine = str(tmp_path / "ref.gsd"), str(tmp_path / "mine.gsd")
    frames = random_frames(seed, P)
    written = expected_file(ref, P, frames=frames)
    mp.spawn(_random_worker, args=(P, free_port(), mine, seed), nprocs=P, join=True)
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read(), written


def test_four_ranks_as_threads_append_host_arrays(tmp_path):
    """`HOOMDTrajectory.append` on per-handle communicators (`pgsd.dist.create_shm` + `pgsd.fl.open(comm=)`): four
    ranks as four threads of this process, host arrays, every C call made without the GIL -- the model's file."""
    import threading
    import uuid
    import pgsd.dist as pdist
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    P, seed = 4, 9
    ref, mine = str(tmp_path / "ref.gsd"), str(tmp_path / "mine.gsd")
    frames = random_frames(seed, P)
    expected_file(ref, P, frames=frames)
    shm = "pgsdthr_%s" % uuid.uuid4().hex[:10]
    errors = []

    def rank_main(rank):
        try:
            comm = pdist.create_shm(shm, rank, P)
            f = fl.open(mine, "w", application="pgsd.hoomd 3.2.0", schema="hoomd", schema_version=[1, 4], comm=comm)
            t = hoomd.HOOMDTrajectory(f)
            for k, g in enumerate(frames):
                t.append(build_frame(hoomd, g, g["counts"], rank, explicit_part_dist=g["explicit"]))
            t.close()
            pdist.release(comm)
        except Exception:  # pragma: no cover
            import traceback
            errors.append((rank, traceback.format_exc()))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not errors and not any(th.is_alive() for th in threads), errors
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()
