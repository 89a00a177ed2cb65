"""Randomised parity: seeded random scenario scripts replayed through
  (1) the compiled reference under mpiexec (only where oracle/_ref exists: the build container
      and the GPU box image, which ships MPICH) -- marked `ref`,
  (2) the CPU oracle,
  (3) the product (same driver source linked against libpgsd_amd.so, one process per rank).
All files and state traces must be identical.  This widens the pinning of the oracle beyond
the 34 committed goldens and checks the product on call sequences nobody hand-picked."""
import os
import random
import subprocess

import pytest

import product
import scenario as S

REF_DRIVER = os.path.join(S.ROOT, "oracle", "_ref", "ref_driver")
MPIEXEC = "/opt/conda/bin/mpiexec"
TYPES = ["u8", "u16", "u32", "u64", "i8", "i16", "i32", "i64", "f32", "f64"]
ESZ = {"u8": 1, "i8": 1, "u16": 2, "i16": 2, "u32": 4, "i32": 4, "f32": 4, "u64": 8, "i64": 8, "f64": 8}
NAMES = ["configuration/step", "particles/N", "particles/position", "particles/velocity", "particles/typeid",
         "particles/image", "log/e", "a", "b/c", "x" * 70, "particles/a_rather_long_auxiliary_chunk_name_number_1",
         "q/0", "q/1", "q/2"]


def make_script(seed, P):
    rng = random.Random(seed)
    lines = ["# fuzz seed %d P %d" % (seed, P),
             "create fuzz_app_%d hoomd %d %d %s 0" % (seed, rng.randint(0, 3), rng.randint(0, 9),
                                                      rng.choice(["rw", "rw", "append"]))]
    maxbuf = 64 * 1024 * 1024
    if rng.random() < 0.35:
        maxbuf = rng.choice([48, 64, 200, 1024, 4096])
        lines.append("maxbuf %d" % maxbuf)
    rng_idx = random.Random(seed * 7919 + 13)      # a stream of its own: the scripts of earlier rounds keep their lines
    if rng_idx.random() < 0.3:
        lines.append("idxbuf %d" % rng_idx.choice([1, 2, 5, 20]))   # end_frame leaves buffered-only frames unflushed
    names = rng.sample(NAMES, rng.randint(3, len(NAMES)))
    n_frames = rng.randint(2, 14)
    cur, last_nonempty = 0, -1      # the file's frame counter: a re-open continues behind the last frame that has entries
    for frame in range(n_frames):
        lines.append("seed %d" % rng.randint(0, 10 ** 6))
        used = set()
        hole = False
        shapes = {}             # name -> (rows of the whole chunk, M, element size) for the read-backs below
        dups = set()
        for _ in range(rng.randint(0, 9)):
            name = rng.choice(names)
            if name in used and rng.random() < 0.9:
                continue
            if name in used:
                dups.add(name)      # written twice in one frame: which entry a read finds is the sort's business
            used.add(name)
            t = rng.choice(TYPES)
            M = rng.randint(1, 4)
            shape = rng.random()
            if shape < 0.12:
                # fl.pyx's default arguments (write_chunk(name, data): all=1, offset=0, N_global=N,
                # global_size=N*M on every rank, fl.pyx:526, 592-598, 640-652): replicated rows on the
                # direct path, the file advances by the SUM of the ranks' sizes (pgsd.c:2240-2246)
                n = rng.randint(1, 30)
                gs = rng.choice([n * M, n * M, 0, 1, 10 ** 9])
                lines.append("samechunk %s %s %d %d %d %d 0 %d 1" % (name, t, n, M, n, M, gs))
                shapes[name] = (n, M, ESZ[t])
                hole = True
            elif shape < 0.22:
                # a proper partition with a global_size that is not the sum (dead argument, pgsd.c:2147-2151)
                if rng.random() < 0.4:
                    dist = "list:" + ",".join(str(rng.choice([0, 0, 1, 3, 17, 40])) for _ in range(P))
                else:
                    dist = "even:%d" % rng.randint(0, 120)
                lines.append("chunkgs %s %s %d 1 %s %d" % (name, t, M, dist, rng.choice([0, 1, 7, 10 ** 12])))
                shapes[name] = (sum(S.dist_counts(dist, P)), M, ESZ[t])
            elif shape < 0.60:
                if rng.random() < 0.3:
                    dist = "list:" + ",".join(str(rng.choice([0, 0, 1, 3, 17, 40])) for _ in range(P))
                else:
                    dist = "even:%d" % rng.randint(0, 120)
                lines.append("chunk %s %s %d 1 %s" % (name, t, M, dist))
                shapes[name] = (sum(S.dist_counts(dist, P)), M, ESZ[t])
            else:
                lo = 0 if P == 1 else 1    # zero-size replicated chunks dead-lock the reference at P > 1
                n = rng.randint(lo, 30)
                if P > 1:
                    # A replicated chunk >= the buffer limit takes the direct path where only rank 0
                    # writes but file_size advances by P copies (pgsd.c:2228-2249).  If nothing is
                    # written behind it the file ends short of its own index, the reference then
                    # calls it corrupt on re-open -- on rank 0 only -- and dead-locks (found by this
                    # fuzzer, seed 114).  The golden `maxbuf` scenario covers the hole itself.
                    esz = {"8": 1, "16": 2, "32": 4, "64": 8}[t[1:]]
                    n = max(lo, min(n, (maxbuf - 1) // (M * esz)))
                lines.append("chunk %s %s %d 0 same:%d" % (name, t, M, n))
                shapes[name] = (n, M, ESZ[t])
            if rng.random() < 0.05:
                lines.append("flush")
        if hole and P > 1:
            # replicated rows written with all=1 leave the file shorter than file_size (P-1 copies'
            # worth of hole); data behind the hole keeps the file as long as its index says, which
            # the reference needs on re-open (see the seed-114 note above)
            lines.append("chunk fuzz/tail u32 1 1 even:%d" % (P + rng.randint(0, 5)))
        lines.append("end_frame")
        if used:
            last_nonempty = cur
        this_frame, cur = cur, cur + 1
        # read some of the frame back: whole chunks and row slabs (pgsd_find_chunk + pgsd_read_chunk, collective
        # in the reference); a chunk without rows answers PGSD_ERROR_FILE_CORRUPT in all three implementations
        for name in dups:
            shapes.pop(name, None)
        if shapes and rng.random() < 0.35:
            for name in rng.sample(sorted(shapes), min(len(shapes), rng.randint(1, 2))):
                rows, M, esz = shapes[name]
                if rng.random() < 0.5 or rows == 0:
                    lines.append("read %d %s 0 %d 0 0 %d %d %d" % (this_frame, name, M, max(rows, 1), M, esz))
                else:
                    n = rng.randint(1, rows)
                    off = rng.randint(0, rows - n)
                    lines.append("read %d %s %d %d %d 1 %d %d %d" % (this_frame, name, n, M, off, n, M, esz))
        r = rng.random()
        if r < 0.25:
            lines.append("dump")
        elif r < 0.35 and frame > 0:
            lines.append("find %d %s" % (rng.randint(0, frame), rng.choice(names)))
        elif r < 0.42:
            lines += ["close", "open %s" % rng.choice(["rw", "append"]), "dump"]
            cur = last_nonempty + 1
            maxbuf = 64 * 1024 * 1024          # a fresh handle starts from the default
            if rng.random() < 0.5:
                maxbuf = rng.choice([64, 4096])
                lines.append("maxbuf %d" % maxbuf)
            if rng_idx.random() < 0.3:
                lines.append("idxbuf %d" % rng_idx.choice([1, 3, 50]))
    lines += ["dump", "close", "open ro", "dump", "names", "find 0 %s" % names[0], "close"]
    return "\n".join(lines) + "\n"


RELOC_TYPES = ["u8", "u16", "u32", "u64", "i8", "i16", "i32", "i64", "f32", "f64"]


def make_relocation_script(seed, P):
    """Relocation-heavy scripts (make_script above rarely fills the 128-entry index): many chunks per frame, many frames,
    index_entries_to_buffer, flushes in the middle of frames, close + re-open, zero-row ranks -- one to three index
    relocations per script.  tools/fuzz_relocation.py replays them through the compiled reference as well."""
    rng = random.Random(seed * 1000003 + P)
    lines = ["create reloc_%d hoomd 1 4 %s 0" % (seed, rng.choice(["rw", "append"]))]
    if rng.random() < 0.4:
        lines.append("maxbuf %d" % rng.choice([16, 64, 4096]))
    if rng.random() < 0.5:
        lines.append("idxbuf %d" % rng.choice([1, 3, 40, 200]))
    n_names = rng.randint(5, 60)
    shape = {}
    for k in range(n_names):
        shape[k] = (rng.choice(RELOC_TYPES), rng.randint(1, 4), rng.random() < 0.6)
    frames = rng.randint(3, 40)
    budget = rng.randint(140, 700)                # index entries of the script: one to three relocations
    per_frame = max(1, min(n_names, budget // frames))
    for f in range(frames):
        lines.append("seed %d" % rng.randint(0, 10 ** 6))
        for k in rng.sample(range(n_names), per_frame):
            t, M, part = shape[k]
            if part:
                dist = ("even:%d" % rng.randint(P, 90)) if rng.random() < 0.8 else \
                    "list:" + ",".join(str(rng.choice([0, 0, 2, 11, 31])) for _ in range(P))
                lines.append("chunk r/%d %s %d 1 %s" % (k, t, M, dist))
            else:
                lines.append("chunk r/%d %s %d 0 same:%d" % (k, t, M, rng.randint(1, 5)))
            if rng.random() < 0.02:
                lines.append("flush")
        lines.append("end_frame")
        r = rng.random()
        if r < 0.06:
            lines += ["close", "open %s" % rng.choice(["rw", "append"]), "dump"]
        elif r < 0.12:
            lines.append("dump")
    lines += ["dump", "close"]
    return "\n".join(lines) + "\n"



@pytest.mark.parametrize("check_eof", ["", "1"])
@pytest.mark.parametrize("mode", [0, 1, 3])
@pytest.mark.parametrize("seed,P", [(s, P) for s in range(500, 506) for P in (1, 2, 3) if (s + P) % 2 == 0] + [(506, 5)])
def test_relocation_heavy_scenarios_equal_the_oracle(seed, P, mode, check_eof, tmp_path, monkeypatch):
    """One to three index relocations per script, every placement mode, with and without the PGSD_CHECK_EOF cross-check
    of the computed end of file: file and state trace are the oracle's."""
    import re
    if check_eof:
        monkeypatch.setenv("PGSD_CHECK_EOF", "1")
    scn = tmp_path / "reloc.scn"
    scn.write_text(make_relocation_script(seed, P))
    o_path, p_path = str(tmp_path / "oracle.gsd"), str(tmp_path / "product.gsd")
    o_log = S.run_oracle(str(scn), o_path, P)
    assert not [ln for ln in o_log if ln.startswith("rc ")], o_log
    assert any("index_allocated=" in ln and "index_allocated=128 " not in ln for ln in o_log)     # it did relocate
    s2 = str(scn) if mode == 0 else product.batched_script(str(scn), str(tmp_path / "b.scn"), mode)
    p_log = product.run_driver(s2, p_path, P)
    with open(o_path, "rb") as a, open(p_path, "rb") as b:
        assert a.read() == b.read()
    strip = lambda lines: [re.sub(r"line=\d+ ", "", ln) for ln in lines]
    assert strip(p_log) == strip(o_log)


def adversarial_script(kind, P):
    """Two hand-made call sequences around the index: 300 chunks in ONE frame (the block doubles twice inside one flush),
    empty frames behind it, a sparse frame, re-open for append; and duplicate names inside frames, empty frames, tiny
    write and index buffers, mid-frame flushes, two re-opens over 50 frames."""
    L = ["create adv hoomd 1 4 rw 0"]
    if kind == "big_frame":
        L.append("seed 5")
        for k in range(300):
            L.append("chunk r/%d %s %d %d %s" % (k, ["f32", "u8", "i64", "f64"][k % 4], 1 + k % 3, 1 if k % 2 else 0,
                                                 ("even:%d" % (P + k % 40)) if k % 2 else "same:%d" % (1 + k % 4)))
        L += ["end_frame", "dump", "end_frame", "end_frame", "seed 6"]
        L += ["chunk r/%d f32 2 1 even:%d" % (k, P + 5) for k in range(0, 300, 7)]
        L += ["end_frame", "dump", "close", "open append", "dump", "seed 7", "chunk r/3 u8 1 1 even:%d" % (P + 3),
              "end_frame", "dump", "close"]
    else:
        L += ["maxbuf 16", "idxbuf 2"]
        for f in range(50):
            L.append("seed %d" % f)
            if f % 5 == 4:
                L.append("end_frame")
                continue
            for k in (0, 1, 1, 2, 0):
                L.append("chunk d/%d u32 1 %d %s" % (k, k % 2, ("even:%d" % (P + 2)) if k % 2 else "same:3"))
            if f % 7 == 0:
                L.append("flush")
            L.append("end_frame")
            if f in (20, 41):
                L += ["close", "open rw", "dump"]
        L += ["dump", "close"]
    return "\n".join(L) + "\n"


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("kind,P", [(k, P) for k in ("big_frame", "dups_and_empty") for P in (1, 2, 5)])
def test_adversarial_index_scenarios_equal_the_oracle_and_the_reference(kind, P, mode, tmp_path, monkeypatch):
    """... through the product in every placement mode with the PGSD_CHECK_EOF cross-check on, against the oracle -- and,
    where the compiled reference is present (build container), against its file as well."""
    import re
    monkeypatch.setenv("PGSD_CHECK_EOF", "1")
    scn = tmp_path / "adv.scn"
    scn.write_text(adversarial_script(kind, P))
    o_path, p_path = str(tmp_path / "oracle.gsd"), str(tmp_path / "product.gsd")
    o_log = S.run_oracle(str(scn), o_path, P)
    assert not [ln for ln in o_log if ln.startswith("rc ")], o_log
    s2 = str(scn) if mode == 0 else product.batched_script(str(scn), str(tmp_path / "b.scn"), mode)
    p_log = product.run_driver(s2, p_path, P)
    with open(o_path, "rb") as a, open(p_path, "rb") as b:
        want = a.read()
        assert want == b.read()
    strip = lambda lines: [re.sub(r"line=\d+ ", "", ln) for ln in lines]
    assert strip(p_log) == strip(o_log)
    if mode == 0 and have_ref():
        r_path = str(tmp_path / "ref.gsd")
        out = subprocess.run([MPIEXEC, "-n", str(P), REF_DRIVER, str(scn), r_path], capture_output=True, timeout=120)
        assert out.returncode == 0, out.stderr.decode()[-500:]
        with open(r_path, "rb") as r:
            assert r.read() == want


def have_ref():
    return os.path.exists(REF_DRIVER) and os.path.exists(MPIEXEC)


CASES = [(seed, P) for seed in range(24) for P in (1, 2, 3) if (seed + P) % 3 != 1 or P == 1]


@pytest.mark.parametrize("seed,P", CASES)
def test_product_equals_oracle_on_random_scenarios(seed, P, tmp_path):
    scn = tmp_path / "fuzz.scn"
    scn.write_text(make_script(seed, P))
    o_path, p_path = str(tmp_path / "oracle.gsd"), str(tmp_path / "product.gsd")
    o_log = S.run_oracle(str(scn), o_path, P)
    assert not [ln for ln in o_log if ln.startswith("rc ")], o_log
    p_log = product.run_driver(str(scn), p_path, P)
    with open(o_path, "rb") as a, open(p_path, "rb") as b:
        assert a.read() == b.read()
    assert p_log == o_log


@pytest.mark.parametrize("mode", [1, 2, 3])
@pytest.mark.parametrize("seed,P", [(seed, P) for seed in range(200, 216) for P in (1, 2, 3)])
def test_product_with_batched_frame_exchange_equals_oracle_on_random_scenarios(seed, P, mode, tmp_path):
    """The same random call sequences with pgsd_set_frame_exchange(1) on every handle: replicated small
    chunks wait in the queue, per-particle host chunks resolve it, flushes / re-opens / buffer-limit changes
    hit it in every state -- files and state traces identical to the oracle's."""
    import re
    scn = tmp_path / "fuzz.scn"
    scn.write_text(make_script(seed, P))
    o_path, p_path = str(tmp_path / "oracle.gsd"), str(tmp_path / "product.gsd")
    o_log = S.run_oracle(str(scn), o_path, P)
    assert not [ln for ln in o_log if ln.startswith("rc ")], o_log
    # mode 2: per-particle host chunks are deferred to the frame's exchange as well (pgsd_set_deferred_rows);
    # mode 3: declared partition (pgsd_set_partition): no exchange wherever the declaration covers a chunk
    p_log = product.run_driver(product.batched_script(str(scn), str(tmp_path / "batched.scn"), mode), p_path, P)
    with open(o_path, "rb") as a, open(p_path, "rb") as b:
        assert a.read() == b.read()
    strip = lambda lines: [re.sub(r"line=\d+ ", "", ln) for ln in lines]
    assert strip(p_log) == strip(o_log)


@pytest.mark.ref
@pytest.mark.skipif(not have_ref(), reason="compiled reference (oracle/_ref) or MPICH not present")
@pytest.mark.parametrize("seed,P", [(s, P) for s in range(100, 116) for P in (1, 2, 4)])
def test_oracle_equals_compiled_reference_on_random_scenarios(seed, P, tmp_path):
    scn = tmp_path / "fuzz.scn"
    scn.write_text(make_script(seed, P))
    r_path, o_path = str(tmp_path / "ref.gsd"), str(tmp_path / "oracle.gsd")
    out = subprocess.run([MPIEXEC, "-n", str(P), REF_DRIVER, str(scn), r_path], capture_output=True, timeout=120)
    assert out.returncode == 0, out.stderr.decode()[-500:]
    r_log = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    o_log = S.run_oracle(str(scn), o_path, P)
    with open(r_path, "rb") as a, open(o_path, "rb") as b:
        assert a.read() == b.read()
    assert o_log == r_log


def test_file_the_reference_leaves_short_is_reported_not_hung(tmp_path):
    """Seed 114 of the fuzzer: at P=2 a replicated chunk on the direct path is the last data of
    the session, so the file ends before the offsets its index records.  On re-open the reference
    flags the file corrupt on rank 0 only and the other ranks wait forever (pgsd.c:1616-1639).
    The product writes the same bytes, and on re-open EVERY rank returns PGSD_ERROR_FILE_CORRUPT
    (-5): same verdict as the reference's rank 0 and the oracle, no dead-lock."""
    scn = tmp_path / "short.scn"
    scn.write_text("create app hoomd 1 4 rw 0\nmaxbuf 64\nseed 1\n"
                   "chunk a u16 3 0 same:17\nchunk b f32 2 1 list:0,0\nend_frame\nclose\nopen rw\n")
    o_path, p_path = str(tmp_path / "o.gsd"), str(tmp_path / "p.gsd")
    lib = S.oracle_lib()
    import ctypes
    o_log = []
    try:
        o_log = S.run_oracle(str(scn), o_path, 2)
    except ValueError:
        pass
    rc = ctypes.c_int(0)
    assert not lib.oracle_open(o_path.encode(), 2, 1, ctypes.byref(rc)) and rc.value == -5
    p_log = product.run_driver(str(scn), p_path, 2, allow_fail=True)
    assert p_log == ["rc line=8 cmd=open rc=-5"]
    with open(o_path, "rb") as a, open(p_path, "rb") as b:
        assert a.read() == b.read()


def long_run_script(n_frames):
    """thousands of short frames: the on-disk index is relocated and doubled again and again
    (pgsd.c:965-1091), names keep arriving, buffered and direct chunks alternate"""
    lines = ["create long_run hoomd 1 4 rw 0"]
    for i in range(n_frames):
        lines += ["seed %d" % (i + 1),
                  "chunk configuration/step u64 1 0 same:1",
                  "chunk particles/N u32 1 0 same:1",
                  "chunk particles/position f32 3 1 even:%d" % (37 + i % 5)]
        if i % 3 == 0:
            lines.append("chunk log/extra%d f64 2 0 same:1" % (i % 50))
        lines.append("end_frame")
        if i in (n_frames // 3, 2 * n_frames // 3):
            lines += ["close", "open append" if i < n_frames // 2 else "open rw"]
    lines += ["dump", "close", "open ro", "dump", "find %d particles/position" % (n_frames - 1), "close"]
    return "\n".join(lines) + "\n"


@pytest.mark.parametrize("P", [1, 2])
def test_long_run_product_equals_oracle(P, tmp_path):
    scn = tmp_path / "long.scn"
    scn.write_text(long_run_script(12000))
    o_path, p_path = str(tmp_path / "oracle.gsd"), str(tmp_path / "product.gsd")
    o_log = S.run_oracle(str(scn), o_path, P)
    p_log = product.run_driver(str(scn), p_path, P, timeout=600)
    with open(o_path, "rb") as a, open(p_path, "rb") as b:
        assert a.read() == b.read()
    assert p_log == o_log


@pytest.mark.ref
@pytest.mark.skipif(not have_ref(), reason="compiled reference (oracle/_ref) or MPICH not present")
@pytest.mark.parametrize("P", [1, 2])
def test_long_run_oracle_equals_compiled_reference(P, tmp_path):
    scn = tmp_path / "long.scn"
    scn.write_text(long_run_script(3000))
    r_path, o_path = str(tmp_path / "ref.gsd"), str(tmp_path / "oracle.gsd")
    out = subprocess.run([MPIEXEC, "-n", str(P), REF_DRIVER, str(scn), r_path], capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr.decode()[-500:]
    r_log = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
    o_log = S.run_oracle(str(scn), o_path, P)
    with open(r_path, "rb") as a, open(o_path, "rb") as b:
        assert a.read() == b.read()
    assert o_log == r_log
