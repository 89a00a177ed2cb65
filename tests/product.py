"""Helpers to build and drive the product (libpgsd_amd.so) from the tests."""
import os
import subprocess
import uuid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pgsd-sph_amd", "csrc")
LIB = os.path.join(ROOT, "pgsd-sph_amd", "pgsd", "libpgsd_amd.so")
TESTS = os.path.join(ROOT, "tests")
TBUILD = os.path.join(TESTS, "build")           # tests/Makefile: drivers, the stand-in librccl, sanitizer builds
DRIVER = os.path.join(TBUILD, "scenario_driver")
DEVICE_DRIVER = os.path.join(TBUILD, "scenario_driver_device")      # rows of every chunk write from HBM

_built = False


def locked_make(args, check=True, **kw):
    """`make <args>` under ONE lock for the whole test session: pytest-xdist workers (and the fixtures of different
    test modules) would otherwise relink a library or a driver that another worker is loading or running
    ("Text file busy", half-written objects)."""
    import fcntl
    import tempfile
    with open(os.path.join(tempfile.gettempdir(), "pgsd_amd_tests_make_%d.lock" % os.getuid()), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        r = subprocess.run(["make"] + list(args), **kw)
    if check and r.returncode != 0:
        raise subprocess.CalledProcessError(r.returncode, ["make"] + list(args))
    return r


def build():
    """make -C pgsd-sph_amd/csrc, then make -C tests (no-ops when up to date)."""
    global _built
    if not _built:
        locked_make(["-C", CSRC, "-j8"], stdout=subprocess.DEVNULL)
        locked_make(["-C", TESTS, "-j8"], stdout=subprocess.DEVNULL)
        _built = True
    return LIB


def run_driver(script, out_path, P, timeout=120, allow_fail=False, driver=None, threads=False):
    """Replay a scenario through the product's C ABI with P processes (shm communicator) -- or, `threads`, with the P
    ranks as P threads of ONE process, each with a communicator of its own (PGSD_DRIVER_THREADS).

    Returns rank 0's stdout lines."""
    build()
    name = "pgsdtest_%s" % uuid.uuid4().hex[:12]
    procs = []
    if threads and P > 1:
        env = dict(os.environ, PGSD_DRIVER_THREADS=str(P), PGSD_SHM_NAME=name)
        procs.append(subprocess.Popen([driver or DRIVER, script, out_path], env=env, stdout=subprocess.PIPE))
    for r in range(0 if procs else P):
        env = dict(os.environ, PGSD_RANK=str(r), PGSD_NRANKS=str(P), PGSD_SHM_NAME=name)
        procs.append(subprocess.Popen([driver or DRIVER, script, out_path], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = None
    try:
        out, _ = procs[0].communicate(timeout=timeout)
        for p in procs[1:]:
            p.wait(timeout=timeout)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        try:
            os.unlink("/dev/shm/" + name)
        except OSError:
            pass
    for r, p in enumerate(procs):
        if p.returncode != 0 and not (allow_fail and p.returncode == 1):
            raise RuntimeError("scenario driver rank %d exited with %s" % (r, p.returncode))
    return [ln for ln in out.decode().splitlines() if ln.strip()]


def batched_script(path, out_path, mode=1):
    """The same scenario with the frame exchange batched (pgsd_set_frame_exchange) on every handle; mode 2: and the
    rows of per-particle host chunks deferred to the frame's exchange (pgsd_set_deferred_rows)."""
    lines = []
    with open(path) as f:
        for line in f:
            tok = line.split("#", 1)[0].split()
            if tok and tok[0] == "prefill" and not os.path.isabs(tok[1]):   # looked up beside the ORIGINAL script
                line = "prefill %s\n" % os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(path)), "..", "files", tok[1]))
            lines.append(line)
            if tok and (tok[0] == "create" or (tok[0] == "open" and tok[1] != "ro")):
                lines.append("batch %d\n" % mode)
    with open(out_path, "w") as f:
        f.writelines(lines)
    return out_path


def local_reads_script(path, out_path):
    """The same scenario with pgsd_set_local_reads(1) on every handle (`localreads 1` behind create / open)."""
    lines = []
    with open(path) as f:
        for line in f:
            tok = line.split("#", 1)[0].split()
            if tok and tok[0] == "prefill" and not os.path.isabs(tok[1]):   # looked up beside the ORIGINAL script
                line = "prefill %s\n" % os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(path)), "..", "files", tok[1]))
            lines.append(line)
            if tok and tok[0] in ("create", "open"):
                lines.append("localreads 1\n")
    with open(out_path, "w") as f:
        f.writelines(lines)
    return out_path


def device_script(path, out_path, mode=1, batch=0, async_seal=False):
    """The same scenario with the rows of every chunk write in HBM (`device <mode>` behind create / open; the
    scenario_driver_device build): 1 = dense device arrays, 2 = rows inside wider arrays.  `batch` as in
    :func:`batched_script`; `async_seal`: frames sealed with pgsd_end_frame_async."""
    lines = []
    with open(path) as f:
        for line in f:
            tok = line.split("#", 1)[0].split()
            if tok and tok[0] == "prefill" and not os.path.isabs(tok[1]):   # looked up beside the ORIGINAL script
                line = "prefill %s\n" % os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(path)), "..", "files", tok[1]))
            lines.append(line)
            if tok and tok[0] in ("create", "open"):
                lines.append("device %d\n" % mode)
                if async_seal:
                    lines.append("async 1\n")
                if batch and tok[0] == "create" or (tok[0] == "open" and tok[1] != "ro" and batch):
                    lines.append("batch %d\n" % batch)
    with open(out_path, "w") as f:
        f.writelines(lines)
    return out_path
