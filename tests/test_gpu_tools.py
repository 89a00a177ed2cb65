"""The user-facing tools on a DEVICE-written trajectory: a three-frame SPH
trajectory is written from arrays in HBM (HOOMD-SPH Scalar4 layout, one fused pack launch per frame), then
`python -m pgsd info`, `pgsd.hoomd.read_log` and the VTK exporter run on the file and their NUMBERS are compared
with the source tensors -- points and every per-particle attribute through a `.vtu` reader written from the VTK
file-format description (no code shared with `pgsd.vtu`), the logged series through `read_log`.

(Interop with upstream `gsd` / OVITO cannot be shown in this image -- `import gsd` fails and there is no network;
it rests on format identity with the reference-written goldens and on the reference's own pure-Python reader
reading these files, tests/test_reference_reader.py.)"""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_tools import PKG_DIR, _vtk_spec_reader

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

N = 20_011


def _run(*argv):
    env = dict(os.environ, PYTHONPATH=PKG_DIR + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable, "-m", "pgsd"] + list(argv), env=env, capture_output=True, text=True, timeout=300)


def _write_device_trajectory(path):
    import pgsd.fl as fl
    import pgsd.hoomd as hoomd
    g = torch.Generator(device="cuda").manual_seed(77)
    frames = []
    with hoomd.open(path, "w") as t:
        for i in range(3):
            pos4 = (torch.rand((N, 4), generator=g, device="cuda") - 0.5) * 20.0
            vel4 = torch.randn((N, 4), generator=g, device="cuda")
            tid = torch.randint(0, 3, (N,), generator=g, device="cuda", dtype=torch.int32)
            pos4[:, 3] = tid.view(torch.float32)                 # HOOMD keeps the type id in position.w
            vel4[:, 3] = torch.rand((N,), generator=g, device="cuda") + 0.5       # ... and the mass in velocity.w
            dpe = torch.rand((N, 4), generator=g, device="cuda")  # density, pressure, energy, slength
            aux1 = torch.randn((N, 4), generator=g, device="cuda")
            img = torch.randint(-2, 3, (N, 4), generator=g, device="cuda", dtype=torch.int32)
            fr = hoomd.Frame()
            fr.configuration.step = 250 * i
            fr.configuration.box = [20, 20, 20, 0, 0, 0]
            fr.particles.N = N
            fr.particles.types = ["Fluid", "Solid", "Wall"]
            fr.particles.typeid = fl.DeviceField.from_tensor(pos4, columns=(3, 4), out_dtype=np.uint32, bitcast=True)
            fr.particles.mass = fl.DeviceField.from_tensor(vel4, columns=(3, 4))
            fr.particles.position = fl.DeviceField.from_tensor(pos4, columns=(0, 3))
            fr.particles.velocity = fl.DeviceField.from_tensor(vel4, columns=(0, 3))
            fr.particles.slength = fl.DeviceField.from_tensor(dpe, columns=(3, 4))
            fr.particles.density = fl.DeviceField.from_tensor(dpe, columns=(0, 1))
            fr.particles.pressure = fl.DeviceField.from_tensor(dpe, columns=(1, 2))
            fr.particles.energy = fl.DeviceField.from_tensor(dpe, columns=(2, 3))
            fr.particles.auxiliary1 = fl.DeviceField.from_tensor(aux1, columns=(0, 3))
            fr.particles.image = fl.DeviceField.from_tensor(img, columns=(0, 3))
            fr.log["kinetic_energy"] = np.array([0.5 * float((vel4[:, :3] ** 2).sum())])
            fr.log["momentum"] = vel4[:, :3].sum(dim=0).cpu().numpy().astype(np.float64)
            t.append(fr)
            frames.append(dict(step=250 * i, pos4=pos4.cpu().numpy(), vel4=vel4.cpu().numpy(), tid=tid.cpu().numpy(),
                               dpe=dpe.cpu().numpy(), aux1=aux1.cpu().numpy(), img=img.cpu().numpy(),
                               ke=fr.log["kinetic_energy"][0], mom=fr.log["momentum"]))
    return frames


def test_cli_read_log_and_vtu_on_a_device_written_trajectory(tmp_path):
    import pgsd.hoomd as hoomd
    import pgsd.vtu as vtu
    path = str(tmp_path / "sph.gsd")
    frames = _write_device_trajectory(path)

    r = _run("info", path)
    assert r.returncode == 0, r.stderr
    assert "frames:          3" in r.stdout and "schema:          hoomd 1.4" in r.stdout
    for chunk, shape in (("particles/position", "%dx3" % N), ("particles/typeid", "%d" % N), ("particles/image", "%dx3" % N),
                         ("particles/auxiliary1", "%dx3" % N), ("log/momentum", "3")):
        line = [ln for ln in r.stdout.splitlines() if chunk + " " in ln]
        assert line and line[0].split()[-1] == shape, (chunk, line)

    log = hoomd.read_log(path)
    assert log["configuration/step"].tolist() == [0, 250, 500]
    np.testing.assert_array_equal(log["log/kinetic_energy"], np.array([f["ke"] for f in frames]))
    np.testing.assert_array_equal(log["log/momentum"], np.stack([f["mom"] for f in frames]))

    files = vtu.pgsd2vtu(path, out_dir=str(tmp_path / "vtk"))
    assert [os.path.basename(f) for f in files] == ["sph_000000000.vtu", "sph_000000250.vtu", "sph_000000500.vtu"]
    for f, name in zip(frames, files):
        got = _vtk_spec_reader(name)
        assert got["_npoints"] == got["_ncells"] == N
        np.testing.assert_array_equal(got["points"], f["pos4"][:, :3])
        np.testing.assert_array_equal(got["velocity"], f["vel4"][:, :3])
        np.testing.assert_array_equal(got["typeid"], f["tid"].view(np.uint32))
        np.testing.assert_array_equal(got["mass"], f["vel4"][:, 3])
        np.testing.assert_array_equal(got["density"], f["dpe"][:, 0])
        np.testing.assert_array_equal(got["pressure"], f["dpe"][:, 1])
        np.testing.assert_array_equal(got["energy"], f["dpe"][:, 2])
        np.testing.assert_array_equal(got["slength"], f["dpe"][:, 3])
        np.testing.assert_array_equal(got["auxiliary1"], f["aux1"][:, :3])
        np.testing.assert_array_equal(got["image"], f["img"][:, :3])
        assert int(got["step"][0]) == f["step"] and got["box"].tolist() == [20, 20, 20, 0, 0, 0]
    r = _run("vtu", path, "-o", str(tmp_path / "vtk2"))
    assert r.returncode == 0 and len(r.stdout.split()) == 3
