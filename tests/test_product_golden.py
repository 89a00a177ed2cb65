"""Host file layer of the product vs the reference: the SAME driver source that produced the
goldens when linked against the reference's pgsd.c is linked against libpgsd_amd.so and run
with one process per rank; the files must be byte-identical."""
import os

import pytest

import product
import scenario as S


def _fails_on_purpose(golden):
    """errors.scn: calls that return error codes (the reference's trace holds `rc line=...` records); the driver
    finishes the script and exits with 1."""
    return any(ln.startswith("rc line=") for ln in S.read_log(golden[:-4] + ".log"))


@pytest.mark.parametrize("name,P", S.golden_cases())
def test_product_matches_reference_file(name, P, tmp_gsd):
    golden = os.path.join(S.GOLDEN, "%s.p%d.gsd" % (name, P))
    log = product.run_driver(S.scenario_path(name), tmp_gsd, P, allow_fail=_fails_on_purpose(golden))
    with open(tmp_gsd, "rb") as f, open(golden, "rb") as g:
        mine, ref = f.read(), g.read()
    assert len(mine) == len(ref)
    assert mine == ref
    assert log == S.read_log(golden[:-4] + ".log")


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("name,P", S.golden_cases())
def test_product_with_batched_frame_exchange_matches_reference_file(name, P, mode, tmp_gsd, tmp_path):
    """pgsd_set_frame_exchange(1): small replicated chunks are queued and ONE allgather per frame places them
    (the device chunks of the GPU tests likewise) -- the file and the state trace must not change by a byte.
    mode 2 adds pgsd_set_deferred_rows(1): per-particle host chunks wait in the queue too (their rows borrowed
    until the exchange), so every frame of every golden is placed by ONE exchange.  mode 3: declared partition
    (pgsd_set_partition before every chunk, per-particle chunks written with PGSD_PARTITION_AUTO): chunks are placed
    with NO exchange at all wherever the declaration can express their sizes."""
    scn = product.batched_script(S.scenario_path(name), str(tmp_path / "batched.scn"), mode)
    golden = os.path.join(S.GOLDEN, "%s.p%d.gsd" % (name, P))
    log = product.run_driver(scn, tmp_gsd, P, allow_fail=_fails_on_purpose(golden))
    with open(tmp_gsd, "rb") as f, open(golden, "rb") as g:
        mine, ref = f.read(), g.read()
    assert len(mine) == len(ref)
    assert mine == ref
    # `batch` lines shift the line numbers the driver prints: compare the trace without them
    import re
    # (and `batch` on a handle whose create failed on purpose, errors.scn, is itself refused)
    strip = lambda lines: [re.sub(r"line=\d+ ", "", ln) for ln in lines if not ln.startswith("rc ") or "cmd=batch" not in ln]
    assert strip(log) == strip(S.read_log(golden[:-4] + ".log"))


MPIEXEC = "/opt/conda/bin/mpiexec"
MPI_DRIVER = os.path.join(product.TBUILD, "scenario_driver_mpi")


@pytest.fixture(scope="module")
def mpi_driver():
    import subprocess
    if not (os.path.exists(MPIEXEC) and os.path.exists("/opt/conda/include/mpi.h")):
        pytest.skip("no MPI installation in this image")
    product.build()
    r = product.locked_make(["-C", product.TESTS, "mpi"], check=False, capture_output=True)
    if r.returncode != 0:
        pytest.skip("cannot link against MPI: " + r.stderr.decode()[-300:])
    return MPI_DRIVER


@pytest.mark.parametrize("name,P", [("posvelid", 8), ("sph_full", 4), ("index_expand", 2), ("names_reloc", 5),
                                    ("maxbuf", 4), ("reopen", 2), ("benchlike", 8)])
def test_product_under_mpiexec_matches_reference_file(mpi_driver, name, P, tmp_gsd):
    """Drop-in under the reference's own launcher: the driver source + libpgsd_amd.so, started with
    `mpiexec -n P`, the library's collectives forwarded to MPI through the communicator vtable
    (INTEGRATION.md section 1)."""
    import subprocess
    out = subprocess.run([MPIEXEC, "-n", str(P), mpi_driver, S.scenario_path(name), tmp_gsd],
                         capture_output=True, timeout=180)
    assert out.returncode == 0, out.stderr.decode()[-500:]
    golden = os.path.join(S.GOLDEN, "%s.p%d.gsd" % (name, P))
    with open(tmp_gsd, "rb") as f, open(golden, "rb") as g:
        assert f.read() == g.read()
    assert [ln for ln in out.stdout.decode().splitlines() if ln.strip()] == S.read_log(golden[:-4] + ".log")


@pytest.fixture(scope="module")
def mpiio_plugin(mpi_driver):
    r = product.locked_make(["-C", product.CSRC, "mpiio"], check=False, capture_output=True)
    plugin = os.path.join(product.ROOT, "pgsd-sph_amd", "pgsd", "libpgsd_amd_mpiio.so")
    if r.returncode != 0 or not os.path.exists(plugin):
        pytest.skip("cannot build the MPI-IO back end: " + r.stderr.decode()[-300:])
    return plugin


@pytest.mark.parametrize("name,P", S.golden_cases())
def test_mpiio_back_end_under_mpiexec_matches_reference_file(mpi_driver, mpiio_plugin, name, P, tmp_gsd):
    """PGSD_IO=mpiio: the bytes reach the file through the reference's own calls --
    MPI_File_write_at / MPI_File_read_at / MPI_File_set_size / MPI_File_get_size at the offsets of pgsd.c:2229, 1154,
    2032, 1289-1306, 1015, 2534 -- via the plugin libpgsd_amd_mpiio.so; the library's collectives go to MPI through
    the communicator vtable.  All 48 reference-written goldens are reproduced under `mpiexec -n P`, reads and their
    hashes included; the default (POSIX pwrite) is unchanged (every other test of this file)."""
    import subprocess
    golden = os.path.join(S.GOLDEN, "%s.p%d.gsd" % (name, P))
    out = subprocess.run([MPIEXEC, "-n", str(P), mpi_driver, S.scenario_path(name), tmp_gsd],
                         capture_output=True, timeout=300, env=dict(os.environ, PGSD_IO="mpiio"))
    assert out.returncode == (1 if _fails_on_purpose(golden) else 0), out.stderr.decode()[-500:]
    with open(tmp_gsd, "rb") as f, open(golden, "rb") as g:
        assert f.read() == g.read()
    assert [ln for ln in out.stdout.decode().splitlines() if ln.strip()] == S.read_log(golden[:-4] + ".log")


def test_mpiio_back_end_fails_loudly_without_its_plugin_or_without_mpi(mpi_driver, mpiio_plugin, tmp_gsd):
    """No silent fall-back to pwrite: PGSD_IO=mpiio with a plugin that cannot be loaded, or in a process that never
    initialised MPI (the plain driver), fails the create with the reason in the error string."""
    import subprocess
    scn = S.scenario_path("posvelid")
    out = subprocess.run([MPIEXEC, "-n", "1", mpi_driver, scn, tmp_gsd], capture_output=True, timeout=120,
                         env=dict(os.environ, PGSD_IO="mpiio", PGSD_MPIIO_LIBRARY="/nonexistent/libpgsd_amd_mpiio.so"))
    assert out.returncode != 0 and b"rc=-1" in out.stdout          # PGSD_ERROR_IO from create, then everything fails
    assert not os.path.exists(tmp_gsd) or os.path.getsize(tmp_gsd) == 0
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import pgsd.fl as fl\n"
            "from pgsd import _lib\n"
            "try:\n"
            "    fl.open(%r, 'w', application='a', schema='s', schema_version=[1, 0])\n"
            "    print('OPENED')\n"
            "except OSError as e:\n"
            "    print('REFUSED', _lib.last_error())\n" % (os.path.join(product.ROOT, "pgsd-sph_amd"), tmp_gsd))
    import sys
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, PGSD_IO="mpiio"))
    assert "REFUSED" in p.stdout and "MPI is not initialised" in p.stdout, (p.stdout, p.stderr[-1000:])


@pytest.mark.parametrize("name", ["readback", "idxbuf", "vone_append"])
def test_local_reads_on_one_rank_change_nothing(name, tmp_gsd, tmp_path):
    """pgsd_set_local_reads(1): a read drains this rank's own copies instead of running the reference's collective
    flush (pgsd.c:2436-2537).  On one rank -- every row is this rank's -- the reads of the goldens that read return
    the same bytes and the file stays the reference's."""
    import re
    scn = product.local_reads_script(S.scenario_path(name), str(tmp_path / "local.scn"))
    golden = os.path.join(S.GOLDEN, "%s.p1.gsd" % name)
    log = product.run_driver(scn, tmp_gsd, 1, allow_fail=_fails_on_purpose(golden))
    with open(tmp_gsd, "rb") as f, open(golden, "rb") as g:
        assert f.read() == g.read()
    strip = lambda ls: [re.sub(r"line=\d+ ", "", ln) for ln in ls if "cmd=localreads" not in ln]
    assert strip(log) == strip(S.read_log(golden[:-4] + ".log"))


@pytest.mark.parametrize("name,P", [c for c in S.golden_cases() if c[1] >= 4])
def test_ranks_as_threads_match_reference_file(name, P, tmp_gsd):
    """The same replay with the P ranks as P THREADS of one driver process, each on a communicator and a handle of
    its own (pgsd_comm_create_shm + pgsd_create_and_open_on / pgsd_open_on; PGSD_DRIVER_THREADS) -- how one process
    drives several GPUs, and how the GPU tests put eight ranks on a box that admits six GPU processes."""
    golden = os.path.join(S.GOLDEN, "%s.p%d.gsd" % (name, P))
    log = product.run_driver(S.scenario_path(name), tmp_gsd, P, allow_fail=_fails_on_purpose(golden), threads=True)
    with open(tmp_gsd, "rb") as f, open(golden, "rb") as g:
        assert f.read() == g.read()
    assert log == S.read_log(golden[:-4] + ".log")


@pytest.mark.parametrize("mode", [0, 1, 3])
@pytest.mark.parametrize("name,P", [c for c in S.golden_cases() if c[0] in ("index_expand", "idxbuf", "names_reloc", "benchlike", "reopen")])
def test_computed_end_of_file_equals_fstat_at_every_index_relocation(name, P, mode, tmp_gsd, tmp_path, monkeypatch):
    """pgsd_expand_file_index places the new block at the file's true end.  The product computes that end from the ranks'
    placements instead of draining and asking fstat; PGSD_CHECK_EOF=1 does both, compares them -- and the
    in-memory index mirror with the block on disk -- and fails the flush with PGSD_ERROR_FILE_CORRUPT on a difference.
    Per-chunk exchange, batched exchange and declared partitions; the file must still be the reference's."""
    monkeypatch.setenv("PGSD_CHECK_EOF", "1")
    scn = S.scenario_path(name) if mode == 0 else product.batched_script(S.scenario_path(name), str(tmp_path / "b.scn"), mode)
    golden = os.path.join(S.GOLDEN, "%s.p%d.gsd" % (name, P))
    log = product.run_driver(scn, tmp_gsd, P, allow_fail=_fails_on_purpose(golden))
    assert not any("rc=-5" in ln for ln in log), [ln for ln in log if "rc=" in ln][:5]
    with open(tmp_gsd, "rb") as f, open(golden, "rb") as g:
        assert f.read() == g.read()
