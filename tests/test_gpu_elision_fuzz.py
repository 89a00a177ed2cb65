"""Randomised trajectories through `HOOMDTrajectory.append` with every per-particle attribute either in HBM or in host
memory: arrays that never change, that always change, that change once and stay changed, that change for one frame and
return to frame 0's values, that hold the schema's default value (in frame 0, later, or both), NaNs, zeros of either
sign, a particle count that changes -- and may change back.  The file must be the one the SAME frames give when every
array is a host array (the reference's elision rules, hoomd.py:654-694, decided by numpy there and by
`compare_bytes_kernel` here) -- byte for byte, in the DEFAULT mode of the device path."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

SEEDS = int(os.environ.get("PGSD_FUZZ_SEEDS", "24"))

SPEC = {"position": (np.float32, 3), "velocity": (np.float32, 3), "typeid": (np.uint32, 1), "mass": (np.float32, 1),
        "body": (np.int32, 1), "density": (np.float32, 1), "image": (np.int32, 3), "auxiliary2": (np.float32, 3)}


DEFAULTS = {"position": 0.0, "velocity": 0.0, "typeid": 0, "mass": 1.0, "body": -1, "density": 0.0, "image": 0,
            "auxiliary2": 0.0}


def _values(rng, name, n, special=False):
    dt, M = SPEC[name]
    shape = (n, M) if M > 1 else (n,)
    if dt == np.float32:
        a = (rng.standard_normal(shape) + 3.0).astype(np.float32)
        if special and n > 0:
            # values whose bytes and whose VALUES tell different stories: NaNs (never equal), zeros of either sign
            flat = a.reshape(-1)
            for v in (np.nan, 0.0, -0.0):
                flat[rng.integers(0, flat.size, size=max(1, flat.size // 50))] = np.float32(v)
        return a
    if dt == np.uint32:
        return rng.integers(1, 5, size=shape).astype(np.uint32)
    return rng.integers(1, 4, size=shape).astype(np.int32)


def _default(name, n, negative_zero=False):
    dt, M = SPEC[name]
    a = np.full((n, M) if M > 1 else (n,), DEFAULTS[name], dtype=dt)
    if negative_zero and dt == np.float32 and DEFAULTS[name] == 0.0 and n > 0:
        a.reshape(-1)[::3] = np.float32(-0.0)            # equal to the default by value, not by bytes
    return a


def _flip_zero_signs(a):
    """The same VALUES with every zero's sign flipped: equal for numpy, different bytes."""
    if a.dtype != np.float32:
        return a
    b = a.copy()
    z = b == 0
    b[z] = -b[z]
    return b


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
    _run(seed, tmp_path, special=False)


@pytest.mark.parametrize("seed", range(SEEDS))
def test_random_trajectories_with_nans_signed_zeros_and_defaults(seed, tmp_path):
    """... with NaNs and zeros of either sign among the values, static arrays re-submitted with their zeros' signs
    flipped, and arrays that hold the default value in frame 0 and / or later."""
    _run(seed, tmp_path, special=True)


def _run(seed, tmp_path, special):
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    rng = np.random.default_rng((9100 if not special else 19100) + seed)
    n0 = int(rng.choice([1, 7, 300, 2000, 5000, 70_000]))
    nframes = int(rng.integers(2, 8))
    resize_at = int(rng.integers(1, nframes)) if rng.random() < 0.3 else None
    back_at = int(rng.integers(resize_at + 1, nframes + 1)) if resize_at is not None and rng.random() < 0.5 else None
    names = [nm for nm in SPEC if rng.random() < 0.7] or ["position"]
    kinds = ["static", "moving", "once", "blip"] + (["default", "default0", "default_later"] if special else [])
    behaviour = {nm: str(rng.choice(kinds)) for nm in names}
    once_at = {nm: int(rng.integers(1, nframes)) for nm in names}
    on_gpu = {nm: bool(rng.random() < 0.7) for nm in names}
    # the frames, as host arrays
    frames = []
    base0 = {nm: _values(rng, nm, n0, special) for nm in names}
    base, changed = base0, {}
    n = n0
    for k in range(nframes):
        if resize_at is not None and k == resize_at:
            n = max(1, n0 // 2 + 1)
            base = {nm: _values(rng, nm, n, special) + SPEC[nm][0](1000) for nm in names}   # (never frame 0's values)
            changed = {}
        if back_at is not None and k == back_at:
            n, base, changed = n0, base0, {}            # frame 0's particle count -- and values -- again
        cur = {}
        for nm in names:
            b = behaviour[nm]
            if b == "default":                          # the default value in every frame (zeros of either sign)
                cur[nm] = _default(nm, n, negative_zero=special and k % 2 == 1)
            elif b == "default0" and k == 0:            # the default in frame 0 (not written), values later
                cur[nm] = _default(nm, n)
            elif b == "default_later" and k > 0 and k % 2 == 0:     # frame 0 holds the chunk, defaults come later
                cur[nm] = _default(nm, n)
            elif k == 0 or b in ("static", "default0", "default_later"):
                cur[nm] = changed.get(nm, base[nm])
                if special and k % 2 == 1:
                    cur[nm] = _flip_zero_signs(cur[nm])
            elif b == "moving":
                cur[nm] = _values(rng, nm, n, special) + SPEC[nm][0](10 * k)
            elif b == "blip":                   # other values in ONE frame, frame 0's again afterwards
                cur[nm] = _values(rng, nm, n, special) + SPEC[nm][0](700) if k == once_at[nm] else changed.get(nm, base[nm])
            else:
                if k == once_at[nm]:
                    changed[nm] = _values(rng, nm, n, special) + SPEC[nm][0](500)
                cur[nm] = changed.get(nm, base[nm])
        frames.append((n, cur))
    a, b = str(tmp_path / "mixed.gsd"), str(tmp_path / "host.gsd")
    keep = []
    for path, device in ((a, True), (b, False)):
        with hoomd.open(path, "w") as t:
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
        assert fa.read() == fb.read(), (seed, special, n0, nframes, resize_at, back_at, behaviour, on_gpu)
