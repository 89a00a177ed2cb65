"""Randomised trajectories through `HOOMDTrajectory.append` with every per-particle attribute either in HBM or in host
memory: arrays that never change, that always change, that change once and stay changed, a particle count that changes
once.  The file must be the one the SAME frames give when every array is a host array (the reference's elision rules,
hoomd.py:654-694, decided by numpy there and by `compare_bytes_kernel` here) -- byte for byte."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SEEDS = int(os.environ.get("PGSD_FUZZ_SEEDS", "24"))

SPEC = {"position": (np.float32, 3), "velocity": (np.float32, 3), "typeid": (np.uint32, 1), "mass": (np.float32, 1),
        "body": (np.int32, 1), "density": (np.float32, 1), "image": (np.int32, 3), "auxiliary2": (np.float32, 3)}


def _values(rng, name, n):
    dt, M = SPEC[name]
    shape = (n, M) if M > 1 else (n,)
    if dt == np.float32:
        return (rng.standard_normal(shape) + 3.0).astype(np.float32)        # never the default
    if dt == np.uint32:
        return rng.integers(1, 5, size=shape).astype(np.uint32)
    return rng.integers(1, 4, size=shape).astype(np.int32)


def _to_device(fl, name, a, keep):
    dt, M = SPEC[name]
    if M == 3 and a.dtype == np.float32:                                    # Scalar4 rows, as HOOMD keeps them
        a4 = np.zeros((a.shape[0], 4), np.float32)
        a4[:, :3] = a
        t = torch.from_numpy(a4).cuda()
        keep.append(t)
        return fl.DeviceField.from_tensor(t, columns=(0, 3))
    if dt == np.uint32:
        t = torch.from_numpy(a.view(np.int32)).cuda()
        keep.append(t)
        return fl.DeviceField.from_tensor(t, out_dtype=np.uint32)
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    keep.append(t)
    return t


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_trajectories_match_the_host_path(seed, tmp_path):
    _run(seed, tmp_path, exact=False)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_trajectories_exact_mode(seed, tmp_path):
    """`device_elision = 'exact'`: arrays that return to frame 0's values for a frame or two ("blip") are in the mix;
    the file is the host path's all the same."""
    _run(seed, tmp_path, exact=True)


def _run(seed, tmp_path, exact):
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    rng = np.random.default_rng((9100 if not exact else 19100) + seed)
    n0 = int(rng.choice([1, 7, 300, 2000, 5000, 70_000]))
    nframes = int(rng.integers(2, 7))
    resize_at = int(rng.integers(1, nframes)) if rng.random() < 0.25 else None
    names = [nm for nm in SPEC if rng.random() < 0.7] or ["position"]
    behaviour = {nm: rng.choice(["static", "moving", "once"] + (["blip"] if exact else [])) for nm in names}
    once_at = {nm: int(rng.integers(1, nframes)) for nm in names}
    on_gpu = {nm: bool(rng.random() < 0.7) for nm in names}
    # the frames, as host arrays
    frames = []
    base = {nm: _values(rng, nm, n0) for nm in names}
    changed = {}
    n = n0
    for k in range(nframes):
        if resize_at is not None and k == resize_at:
            n = max(1, n0 // 2 + 1)
            base = {nm: _values(rng, nm, n) + SPEC[nm][0](1000) for nm in names}       # (never frame 0's values)
            changed = {}
        cur = {}
        for nm in names:
            b = behaviour[nm]
            if k == 0 or b == "static":
                cur[nm] = changed.get(nm, base[nm])
            elif b == "moving":
                # never frame 0's values again: an array that differed once is written from then on by the device
                # path even if it returns to frame 0's values (the host path would elide it there) -- the one place
                # where the two files may differ, exercised by tests/test_gpu_elision.py, kept out of this comparison
                cur[nm] = _values(rng, nm, n) + SPEC[nm][0](10 * k)
            elif b == "blip":                   # other values in ONE frame, frame 0's again afterwards
                cur[nm] = _values(rng, nm, n) + SPEC[nm][0](700) if k == once_at[nm] else changed.get(nm, base[nm])
            else:
                if k == once_at[nm]:
                    changed[nm] = _values(rng, nm, n) + SPEC[nm][0](500)
                cur[nm] = changed.get(nm, base[nm])
        frames.append((n, cur))
    a, b = str(tmp_path / "mixed.gsd"), str(tmp_path / "host.gsd")
    keep = []
    for path, device in ((a, True), (b, False)):
        with hoomd.open(path, "w") as t:
            if exact:
                t.device_elision = 'exact'
            for k, (n, cur) in enumerate(frames):
                fr = hoomd.Frame()
                fr.configuration.step = k
                fr.particles.N = n
                for nm, v in cur.items():
                    setattr(fr.particles, nm, _to_device(fl, nm, v, keep) if device and on_gpu[nm] else v)
                t.append(fr, wait=bool(rng.random() < 0.7) if device else True)
            if device:
                t.file.frame_sync()
    with open(a, "rb") as fa, open(b, "rb") as fb:
        assert fa.read() == fb.read(), (seed, exact, n0, nframes, resize_at, behaviour, on_gpu)
