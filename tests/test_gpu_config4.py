"""BASELINE config 4 on the GPU: the full PGSD-SPH particle schema (hoomd.py:167-184, 15 chunks,
112 B/particle) and its union with the upstream HOOMD attributes (pgsd.tex:508-521: charge,
diameter, moment_inertia, orientation, angmom; 20 chunks, 164 B/particle) written from
HOOMD-SPH-style device arrays -- Scalar4 position (w = type id bits) / velocity (w = mass) /
(density, pressure, energy, slength) / auxiliary1-4, int4 image, scalar arrays -- through the fused
device path, compared with
  * the file the REFERENCE wrote for the same closed-form values (tests/golden/files/sph_full.p1.gsd),
  * the CPU oracle's file on random values (1 rank and 2 ranks sharing the GPU),
  * the CPU oracle's file at 10 M particles (sha256),
and `HOOMDTrajectory.append` with GPU-resident attributes compared with the ORACLE's file of the
call sequence the reference sketches (model in tests/test_hoomd_append_oracle.py)."""
import hashlib
import multiprocessing as mp
import os
import uuid

import numpy as np
import pytest

import scenario as S
from test_gpu_file import _oracle_frames, dev

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

# chunk -> (pgsd type id, columns), in the order of hoomd.py:167-184 / the union's tail
SPH = [("typeid", 3, 1), ("mass", 9, 1), ("body", 7, 1), ("position", 9, 3), ("velocity", 9, 3),
       ("slength", 9, 1), ("density", 9, 1), ("pressure", 9, 1), ("energy", 9, 1),
       ("auxiliary1", 9, 3), ("auxiliary2", 9, 3), ("auxiliary3", 9, 3), ("auxiliary4", 9, 3), ("image", 7, 3)]
UPSTREAM = [("charge", 9, 1), ("diameter", 9, 1), ("moment_inertia", 9, 3), ("orientation", 9, 4), ("angmom", 9, 4)]


def hoomd_sph_device_arrays(values, fl):
    """values: {chunk: (N, M) numpy array of the chunk's dtype}.  Lays them out the way HOOMD-SPH keeps
    them on the device and returns [(chunk name, DeviceField)] in schema order + the keep-alive list."""
    N = values["position"].shape[0]
    f32 = np.float32

    def four(xyz, w):
        a = np.zeros((N, 4), dtype=xyz.dtype)
        a[:, :xyz.shape[1]] = xyz
        if w is not None:
            a[:, 3] = w
        return a
    pos4 = dev(four(values["position"], values["typeid"][:, 0].view(f32)))
    vel4 = dev(four(values["velocity"], values["mass"][:, 0]))
    dpe4 = dev(np.stack([values["density"][:, 0], values["pressure"][:, 0], values["energy"][:, 0],
                         values["slength"][:, 0]], axis=1).astype(f32))
    aux = [dev(four(values["auxiliary%d" % k], None)) for k in (1, 2, 3, 4)]
    img4 = dev(four(values["image"], None))
    body = dev(values["body"][:, 0])
    D = fl.DeviceField.from_tensor
    fields = {
        "typeid": D(pos4, columns=(3, 4), out_dtype=np.uint32, bitcast=True),
        "mass": D(vel4, columns=(3, 4)), "body": D(body),
        "position": D(pos4, columns=(0, 3)), "velocity": D(vel4, columns=(0, 3)),
        "slength": D(dpe4, columns=(3, 4)), "density": D(dpe4, columns=(0, 1)),
        "pressure": D(dpe4, columns=(1, 2)), "energy": D(dpe4, columns=(2, 3)),
        "image": D(img4, columns=(0, 3)),
    }
    for k in (1, 2, 3, 4):
        fields["auxiliary%d" % k] = D(aux[k - 1], columns=(0, 3))
    keep = [pos4, vel4, dpe4, img4, body] + aux
    if "charge" in values:
        charge, diameter = dev(values["charge"][:, 0]), dev(values["diameter"][:, 0])
        inertia, orient, angmom = dev(values["moment_inertia"]), dev(values["orientation"]), dev(values["angmom"])
        fields.update(charge=D(charge), diameter=D(diameter), moment_inertia=D(inertia), orientation=D(orient),
                      angmom=D(angmom))
        keep += [charge, diameter, inertia, orient, angmom]
    return fields, keep


def test_device_arrays_reproduce_the_reference_written_sph_full_golden(tmp_gsd):
    """Chunk for chunk the sequence of tests/golden/scenarios/sph_full.scn at P = 1 (the chunk order of
    hoomd.py:582-632), the 14 per-particle chunks of frame 0 packed by ONE fused launch out of Scalar4 /
    int4 / scalar device arrays: byte-identical to the file the compiled reference wrote."""
    import pgsd.fl as fl
    N = 333
    part = np.array([N])
    f = fl.open(tmp_gsd, "w", application="pgsd.hoomd_3.2.0", schema="hoomd", schema_version=[1, 4])
    f.configure_device(profile=True)

    def small(name, t, M, n, seed):
        f.write_chunk(name, S.gen_data(t, seed, 0, n, M), write_all=False)

    seed = 7
    small("configuration/step", 4, 1, 1, seed)
    small("configuration/dimensions", 1, 1, 1, seed)
    small("configuration/box", 9, 1, 6, seed)
    small("particles/N", 3, 1, 1, seed)
    small("particles/types", 5, 2, 2, seed)
    values = {name: S.gen_data(t, seed, 0, N, M) for name, t, M in SPH}
    fields, keep = hoomd_sph_device_arrays(values, fl)
    f.write_chunks([("particles/" + name, fields[name]) for name, _, _ in SPH], offset=part)
    small("particles/type_shapes", 5, 3, 2, seed)
    small("log/energy", 10, 1, 1, seed)
    f.end_frame()
    assert f.device_stats()["pack_launches"] == 1          # hoomd_pvi + dpe + aux + image + body: 8 arrays, 1 launch
    seed = 8
    small("configuration/step", 4, 1, 1, seed)
    names = ("position", "velocity", "density", "pressure", "energy")
    spec = {name: (t, M) for name, t, M in SPH}
    values8 = {name: S.gen_data(t, seed, 0, N, M) for name, t, M in SPH}
    fields8, keep8 = hoomd_sph_device_arrays(values8, fl)
    f.write_chunks([("particles/" + name, fields8[name]) for name in names], offset=part)
    small("log/energy", 10, 1, 1, seed)
    f.end_frame()
    f.close()
    assert spec["position"] == (9, 3)
    with open(tmp_gsd, "rb") as a, open(os.path.join(S.GOLDEN, "sph_full.p1.gsd"), "rb") as b:
        assert a.read() == b.read()


def random_values(rng, N, union):
    values = {}
    for name, t, M in SPH + (UPSTREAM if union else []):
        dt = S.NP_TYPES[t]
        if name == "typeid":
            values[name] = rng.integers(0, 5, size=(N, M)).astype(dt)
        elif np.dtype(dt).kind == "f":
            values[name] = (rng.standard_normal((N, M)) * 30).astype(dt)
        else:
            values[name] = rng.integers(-4, 5, size=(N, M)).astype(dt)
    return values


def oracle_chunks(values, spec, counts):
    row0 = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int)
    return [("particles/" + name, t, M, True,
             [np.ascontiguousarray(values[name][int(row0[r]):int(row0[r]) + counts[r]]) for r in range(len(counts))])
            for name, t, M in spec]


@pytest.mark.parametrize("union", [False, True])
@pytest.mark.parametrize("N", [1, 10007])
def test_full_schema_from_device_arrays_matches_oracle(N, union, tmp_path):
    """15-chunk SPH set / 20-chunk union set, two frames, one fused write per frame."""
    import pgsd.fl as fl
    rng = np.random.default_rng(N + int(union))
    spec = SPH + (UPSTREAM if union else [])
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    f = fl.open(mine, "w", application="app", schema="hoomd", schema_version=[1, 4])
    f.configure_device(slab_bytes=128 * 1024, n_slabs=3)
    frames = []
    for frame in range(2):
        values = random_values(rng, N, union)
        fields, keep = hoomd_sph_device_arrays(values, fl)
        step = np.array([[frame]], dtype=np.uint64)
        f.write_chunk("configuration/step", step, write_all=False)
        f.write_chunks([("particles/" + name, fields[name]) for name, _, _ in spec], offset=np.array([N]))
        f.end_frame()
        frames.append([("configuration/step", 4, 1, False, [step])] + oracle_chunks(values, spec, [N]))
    f.close()
    _oracle_frames(ref, 1, frames)
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()


def _union_rank(rank, P, shm, path, counts, q):
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "pgsd-sph_amd"))
        sys.path.insert(0, os.path.join(root, "tests"))
        import torch as _t
        import pgsd.fl as fl
        from pgsd import _lib
        import test_gpu_config4 as me
        __import__("pgsd.dist").dist.init_shm(shm, rank, P)
        _t.cuda.set_device(0)
        row0 = int(sum(counts[:rank]))
        n = counts[rank]
        f = fl.open(path, "w", application="app", schema="hoomd", schema_version=[1, 4])
        rng = np.random.default_rng(77)
        for frame in range(2):
            values = me.random_values(rng, sum(counts), True)          # same global values on every rank
            local = {k: np.ascontiguousarray(v[row0:row0 + n]) for k, v in values.items()}
            f.write_chunk("configuration/step", np.array([frame], dtype=np.uint64), write_all=False)
            if n > 0:
                fields, keep = me.hoomd_sph_device_arrays(local, fl)
                f.write_chunks([("particles/" + name, fields[name]) for name, _, _ in me.SPH + me.UPSTREAM],
                               offset=np.array(counts), rank=rank)
            else:
                # a rank without particles still takes part in every collective chunk write
                z = _t.zeros((0, 4), device="cuda")
                f.write_chunks([("particles/" + name, fl.DeviceField(z.data_ptr() or 16, S.NP_TYPES[t], 0, M))
                                for name, t, M in me.SPH + me.UPSTREAM], offset=np.array(counts), rank=rank)
            f.end_frame()
        f.close()
        _lib.lib.pgsd_comm_finalize()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("counts", [[4001, 3000], [0, 1234, 5]])
def test_union_schema_multi_rank_matches_oracle(counts, tmp_path):
    """The 20-chunk union set from 2-3 ranks sharing cuda:0 (shm communicator): == the oracle's P-rank file."""
    P = len(counts)
    mine, ref = str(tmp_path / "mine.gsd"), str(tmp_path / "ref.gsd")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shm = "pgsdgpu_%s" % uuid.uuid4().hex[:10]
    procs = [ctx.Process(target=_union_rank, args=(r, P, shm, mine, counts, q)) for r in range(P)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(msg == "ok" for _, msg in results), results
    rng = np.random.default_rng(77)
    frames = []
    for frame in range(2):
        values = random_values(rng, sum(counts), True)
        frames.append([("configuration/step", 4, 1, False, [np.array([[frame]], dtype=np.uint64)] * P)]
                      + oracle_chunks(values, SPH + UPSTREAM, counts))
    _oracle_frames(ref, P, frames)
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()


@pytest.mark.parametrize("union", [False, True])
def test_config4_full_size_frame_matches_oracle_file(union, tmp_path):
    """Config 4 at its real size: 10 M particles, 1.12 GB (SPH set) / 1.64 GB (union) per frame, written
    from device arrays; sha256 of the file == sha256 of the oracle's file from host copies."""
    import pgsd.fl as fl
    N = 10_000_000
    spec = SPH + (UPSTREAM if union else [])
    d = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    mine = os.path.join(d, "pgsd_c4_mine_%d.gsd" % os.getpid())
    ref = os.path.join(d, "pgsd_c4_ref_%d.gsd" % os.getpid())
    try:
        g = torch.Generator(device="cuda").manual_seed(99)
        pos4 = (torch.rand((N, 4), generator=g, device="cuda") - 0.5) * 100.0
        pos4[:, 3] = torch.randint(0, 7, (N,), generator=g, device="cuda", dtype=torch.int32).view(torch.float32)
        vel4 = torch.randn((N, 4), generator=g, device="cuda")
        dpe4 = torch.rand((N, 4), generator=g, device="cuda")
        aux = [torch.randn((N, 4), generator=g, device="cuda") for _ in range(4)]
        img4 = torch.randint(-2, 3, (N, 4), generator=g, device="cuda", dtype=torch.int32)
        body = torch.randint(-1, 50, (N,), generator=g, device="cuda", dtype=torch.int32)
        D = fl.DeviceField.from_tensor
        fields = {"typeid": D(pos4, columns=(3, 4), out_dtype=np.uint32, bitcast=True), "mass": D(vel4, columns=(3, 4)),
                  "body": D(body), "position": D(pos4, columns=(0, 3)), "velocity": D(vel4, columns=(0, 3)),
                  "slength": D(dpe4, columns=(3, 4)), "density": D(dpe4, columns=(0, 1)),
                  "pressure": D(dpe4, columns=(1, 2)), "energy": D(dpe4, columns=(2, 3)),
                  "image": D(img4, columns=(0, 3))}
        host = {"typeid": pos4[:, 3:4].contiguous().view(torch.int32).cpu().numpy().view(np.uint32),
                "mass": vel4[:, 3:4].cpu().numpy(), "body": body.cpu().numpy().reshape(-1, 1),
                "position": pos4[:, :3].cpu().numpy(), "velocity": vel4[:, :3].cpu().numpy(),
                "slength": dpe4[:, 3:4].cpu().numpy(), "density": dpe4[:, 0:1].cpu().numpy(),
                "pressure": dpe4[:, 1:2].cpu().numpy(), "energy": dpe4[:, 2:3].cpu().numpy(),
                "image": img4[:, :3].cpu().numpy()}
        for k in range(4):
            fields["auxiliary%d" % (k + 1)] = D(aux[k], columns=(0, 3))
            host["auxiliary%d" % (k + 1)] = aux[k][:, :3].cpu().numpy()
        if union:
            extra = {"charge": torch.randn((N,), generator=g, device="cuda"),
                     "diameter": torch.rand((N,), generator=g, device="cuda"),
                     "moment_inertia": torch.rand((N, 3), generator=g, device="cuda"),
                     "orientation": torch.randn((N, 4), generator=g, device="cuda"),
                     "angmom": torch.randn((N, 4), generator=g, device="cuda")}
            for name, t in extra.items():
                fields[name] = D(t)
                host[name] = t.cpu().numpy().reshape(N, -1)
        f = fl.open(mine, "w", application="app", schema="hoomd", schema_version=[1, 4])
        step = np.array([[5]], dtype=np.uint64)
        f.write_chunk("configuration/step", step, write_all=False)
        f.write_chunks([("particles/" + name, fields[name]) for name, _, _ in spec], offset=np.array([N]))
        f.end_frame()
        f.close()
        _oracle_frames(ref, 1, [[("configuration/step", 4, 1, False, [step])]
                                + [("particles/" + name, t, M, True, [np.ascontiguousarray(host[name])])
                                   for name, t, M in spec]])
        assert os.path.getsize(mine) == os.path.getsize(ref) > N * (164 if union else 112)

        def digest(p):
            h = hashlib.sha256()
            with open(p, "rb") as fh:
                for block in iter(lambda: fh.read(1 << 24), b""):
                    h.update(block)
            return h.hexdigest()
        assert digest(mine) == digest(ref)
    finally:
        for p in (mine, ref):
            if os.path.exists(p):
                os.unlink(p)


# ---------------------------------------------------------------- HOOMDTrajectory.append vs the oracle
def _device_frame(hoomd, fl, g, counts, rank):
    """The frame of tests/test_hoomd_append_oracle.py with every per-particle attribute on the GPU:
    position + type id and velocity + mass as Scalar4 arrays where both are set, plain tensors otherwise."""
    import test_hoomd_append_oracle as A
    fr = A.build_frame(hoomd, g, counts, rank, False)
    p = fr.particles
    n = p.N
    keep = []
    if p.position is not None and p.typeid is not None:
        pos4 = np.zeros((n, 4), np.float32)
        pos4[:, :3] = np.asarray(p.position, np.float32).reshape(n, 3)
        pos4[:, 3] = np.asarray(p.typeid, np.uint32).view(np.float32)
        t = dev(pos4)
        keep.append(t)
        p.position = fl.DeviceField.from_tensor(t, columns=(0, 3))
        p.typeid = fl.DeviceField.from_tensor(t, columns=(3, 4), out_dtype=np.uint32, bitcast=True)
    if p.velocity is not None and p.mass is not None:
        vel4 = np.zeros((n, 4), np.float64)                       # a double-precision build: f64 -> f32 on the way out
        vel4[:, :3] = np.asarray(p.velocity, np.float32).reshape(n, 3)
        vel4[:, 3] = np.asarray(p.mass, np.float32)
        t = dev(vel4)
        keep.append(t)
        p.velocity = fl.DeviceField.from_tensor(t, columns=(0, 3), out_dtype=np.float32)
        p.mass = fl.DeviceField.from_tensor(t, columns=(3, 4), out_dtype=np.float32)
    for name, dt in A.DTYPES.items():
        v = getattr(p, name, None)
        if name in ("value", "group") or v is None or isinstance(v, fl.DeviceField):
            continue
        t = dev(np.ascontiguousarray(v, dtype=dt))
        if dt == np.uint32:
            t = dev(np.ascontiguousarray(v, dtype=dt).view(np.int32))
            setattr(p, name, fl.DeviceField.from_tensor(t, out_dtype=np.uint32))
        else:
            setattr(p, name, t)
        keep.append(t)
    fr._keep = keep
    return fr


def test_hoomd_append_device_fields_match_oracle_file(tmp_path):
    """`HOOMDTrajectory.append` with GPU-resident attributes == the oracle's file of the sketched call
    sequence (hoomd.py:597-640), not merely == the product's own host path."""
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    import test_hoomd_append_oracle as A
    ref, mine = str(tmp_path / "ref.gsd"), str(tmp_path / "mine.gsd")
    A.expected_file(ref, 1, device=True)
    with hoomd.open(mine, "w") as t:
        for g in A.global_frames():
            t.append(_device_frame(hoomd, fl, g, [g["n"]], 0))
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()


def _append_rank(rank, P, shm, path, q):
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, os.path.join(root, "pgsd-sph_amd"))
        sys.path.insert(0, os.path.join(root, "tests"))
        import torch as _t
        import pgsd.fl as fl
        import pgsd.hoomd as hoomd
        from pgsd import _lib
        import test_gpu_config4 as me
        import test_hoomd_append_oracle as A
        __import__("pgsd.dist").dist.init_shm(shm, rank, P)
        _t.cuda.set_device(0)
        t = hoomd.open(path, "w")
        for g in A.global_frames():
            t.append(me._device_frame(hoomd, fl, g, A.partition(g["n"], P), rank))
        t.close()
        _lib.lib.pgsd_comm_finalize()
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
        raise


def test_hoomd_append_device_fields_two_ranks_match_oracle_file(tmp_path):
    import test_hoomd_append_oracle as A
    P = 2
    ref, mine = str(tmp_path / "ref.gsd"), str(tmp_path / "mine.gsd")
    A.expected_file(ref, P, device=True)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    shm = "pgsdgpu_%s" % uuid.uuid4().hex[:10]
    procs = [ctx.Process(target=_append_rank, args=(r, P, shm, mine, q)) for r in range(P)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert all(msg == "ok" for _, msg in results), results
    with open(mine, "rb") as a, open(ref, "rb") as b:
        assert a.read() == b.read()
