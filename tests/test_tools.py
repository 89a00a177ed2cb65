"""Tooling around the file layer: command line and VTK export."""
import os
import subprocess
import sys

import numpy as np
import pytest

PKG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pgsd-sph_amd")


def _run(*argv, stdin=None):
    env = dict(os.environ, PYTHONPATH=PKG_DIR + os.pathsep + os.environ.get("PYTHONPATH", ""))
    return subprocess.run([sys.executable, "-m", "pgsd"] + list(argv), env=env, input=stdin,
                          capture_output=True, text=True, timeout=300)


def _write_traj(path, N=1000, frames=3):
    import pgsd.hoomd as hoomd
    rng = np.random.default_rng(4)
    out = []
    with hoomd.open(path, 'w') as t:
        for i in range(frames):
            fr = hoomd.Frame()
            fr.configuration.step = 100 * i
            fr.configuration.box = [10, 10, 10, 0, 0, 0]
            fr.particles.N = N
            fr.particles.types = ['F', 'S']
            fr.particles.position = rng.random((N, 3), dtype=np.float32)
            fr.particles.velocity = rng.random((N, 3), dtype=np.float32)
            fr.particles.typeid = rng.integers(0, 2, size=N).astype(np.uint32)
            fr.particles.density = rng.random(N, dtype=np.float32)
            t.append(fr)
            out.append(fr)
    return out


def test_cli_version_and_usage():
    r = _run("--version")
    assert r.returncode == 0 and r.stdout.strip() == "pgsd 3.2.0"
    r = _run()
    assert r.returncode == 2 and "usage" in r.stderr.lower() + r.stdout.lower()


def test_cli_info_and_read(tmp_path):
    path = str(tmp_path / "t.gsd")
    _write_traj(path)
    r = _run("info", path)
    assert r.returncode == 0, r.stderr
    assert "frames:          3" in r.stdout and "schema:          hoomd 1.4" in r.stdout
    assert "particles/position" in r.stdout and "1000x3" in r.stdout
    # the interactive command with a scripted stdin
    r = _run("read", path, stdin="print('N =', traj[1].particles.N, handle.nframes)\n")
    assert r.returncode == 0, r.stderr
    assert "N = 1000 3" in r.stdout and "Number of frames: 3" in r.stderr + r.stdout
    r = _run("read", "-s", "none", path, stdin="print(handle.schema)\n")
    assert r.returncode == 0 and "hoomd" in r.stdout
    r = _run("read", str(tmp_path / "missing.gsd"), stdin="")
    assert r.returncode == 1 and r.stderr.startswith("Error:")


def test_vtu_export_round_trip(tmp_path):
    import pgsd.vtu as vtu
    path = str(tmp_path / "run.gsd")
    frames = _write_traj(path, N=321, frames=2)
    r = _run("vtu", path, "-o", str(tmp_path / "vtk"))
    assert r.returncode == 0, r.stderr
    files = r.stdout.split()
    assert [os.path.basename(f) for f in files] == ["run_000000000.vtu", "run_000000100.vtu"]
    assert os.path.exists(str(tmp_path / "vtk" / "run.pvd"))
    for fr, name in zip(frames, files):
        got = vtu.read_vtu_arrays(name)
        assert got["points"].tobytes() == fr.particles.position.tobytes()
        assert got["velocity"].tobytes() == fr.particles.velocity.tobytes()
        assert (got["typeid"] == fr.particles.typeid).all() and got["typeid"].dtype == np.uint32
        assert got["density"].tobytes() == fr.particles.density.tobytes()
        assert (got["connectivity"] == np.arange(321)).all() and (got["types"] == 1).all()
        assert int(got["step"][0]) == fr.configuration.step
        assert got["box"].tolist() == [10, 10, 10, 0, 0, 0]
        import xml.dom.minidom
        head = open(name, 'rb').read().split(b'<AppendedData')[0] + b'</VTKFile>'
        xml.dom.minidom.parseString(head)      # the XML part is well formed
    with pytest.raises(ValueError):
        vtu.write_vtu(str(tmp_path / "bad.vtu"), np.zeros((3, 2)))


def _vtk_spec_reader(path):
    """A reader written from the VTK XML file-format description, independent of pgsd.vtu: the XML part is
    parsed with ElementTree; appended raw data starts behind the '_' that follows the <AppendedData> tag, every
    DataArray's `offset` counts from there, each block is a header_type (UInt64) byte count + the bytes."""
    import struct
    import xml.etree.ElementTree as ET
    raw = open(path, "rb").read()
    at = raw.index(b"<AppendedData")
    start = raw.index(b"_", raw.index(b">", at)) + 1
    root = ET.fromstring(raw[:at] + b"</VTKFile>")
    assert root.attrib["type"] == "UnstructuredGrid" and root.attrib["byte_order"] == "LittleEndian"
    assert root.attrib["header_type"] == "UInt64"
    np_of = {"Float32": "<f4", "Float64": "<f8", "Int32": "<i4", "UInt32": "<u4", "Int64": "<i8", "UInt64": "<u8",
             "UInt8": "u1", "Int8": "i1"}
    piece = root.find("UnstructuredGrid/Piece")
    arrays = {"_npoints": int(piece.attrib["NumberOfPoints"]), "_ncells": int(piece.attrib["NumberOfCells"])}
    for da in root.iter("DataArray"):
        assert da.attrib["format"] == "appended"
        off = start + int(da.attrib["offset"])
        nbytes = struct.unpack_from("<Q", raw, off)[0]
        a = np.frombuffer(raw, dtype=np_of[da.attrib["type"]], count=nbytes // np.dtype(np_of[da.attrib["type"]]).itemsize,
                          offset=off + 8)
        nc = int(da.attrib["NumberOfComponents"])
        arrays[da.attrib["Name"]] = a.reshape(-1, nc) if nc > 1 else a
    return arrays


def test_vtu_files_hold_the_frames_values_for_an_independent_reader(tmp_path):
    """The exported numbers, read back by a parser that shares no code with the exporter: points, every
    per-particle field, one VTK_VERTEX cell per particle (connectivity i, offsets i+1, type 1)."""
    import pgsd.vtu as vtu
    path = str(tmp_path / "run.gsd")
    frames = _write_traj(path, N=257, frames=2)
    files = vtu.pgsd2vtu(path, out_dir=str(tmp_path / "vtk"))
    for fr, name in zip(frames, files):
        got = _vtk_spec_reader(name)
        assert got["_npoints"] == got["_ncells"] == 257
        np.testing.assert_array_equal(got["points"], fr.particles.position)
        np.testing.assert_array_equal(got["velocity"], fr.particles.velocity)
        np.testing.assert_array_equal(got["typeid"], fr.particles.typeid)
        np.testing.assert_array_equal(got["density"], fr.particles.density)
        np.testing.assert_array_equal(got["connectivity"], np.arange(257))
        np.testing.assert_array_equal(got["offsets"], np.arange(1, 258))
        assert (got["types"] == 1).all()                      # VTK_VERTEX
        assert int(got["step"][0]) == fr.configuration.step
