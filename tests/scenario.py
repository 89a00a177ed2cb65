"""Scenario scripts shared by every checker of the host file layer.

The same `.scn` script (grammar: tests/drivers/scenario_driver.c) is replayed through
  * the compiled reference (oracle/_ref/ref_driver, build container only) -> goldens,
  * the plain-C oracle (oracle/libpgsd_oracle.so) via :func:`run_oracle`,
  * the product (`pgsd.fl` over libpgsd_amd.so) via :func:`run_product_rank`.

Inputs are closed form (`mix64`), so nothing but the script and a seed is stored.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "files")
SCENARIOS = os.path.join(ROOT, "tests", "golden", "scenarios")

TYPE_IDS = {"u8": 1, "u16": 2, "u32": 3, "u64": 4, "i8": 5, "i16": 6, "i32": 7, "i64": 8,
            "f32": 9, "f64": 10}
NP_TYPES = {1: np.uint8, 2: np.uint16, 3: np.uint32, 4: np.uint64, 5: np.int8, 6: np.int16,
            7: np.int32, 8: np.int64, 9: np.float32, 10: np.float64}
FLAGS = {"rw": 1, "ro": 2, "append": 3}


def mix64(seed, gid, c, M):
    """numpy twin of mix64() in tests/drivers/scenario_driver.c (uint64 wrap-around)."""
    with np.errstate(over="ignore"):
        gid = np.asarray(gid, dtype=np.uint64)
        c = np.asarray(c, dtype=np.uint64)
        u = (gid * np.uint64(M) + c) * np.uint64(0x9E3779B97F4A7C15) \
            + np.uint64(seed) * np.uint64(0xD1B54A32D192ED03) + np.uint64(0x632BE59BD9B4E019)
        u ^= u >> np.uint64(29)
        u *= np.uint64(0xBF58476D1CE4E5B9)
        u ^= u >> np.uint64(32)
    return u


def gen_data(type_id, seed, gid0, N, M):
    """Rows [gid0, gid0+N) x M of the closed-form test data, dtype per pgsd type id."""
    dt = NP_TYPES[type_id]
    if N == 0:
        return np.zeros((0, M), dtype=dt)
    gid = (np.arange(N, dtype=np.uint64) + np.uint64(gid0))[:, None]
    c = np.arange(M, dtype=np.uint64)[None, :]
    u = mix64(seed, gid, c, M)
    if type_id <= 8:
        return u.astype(dt)  # wraps like the C casts
    d = ((u % np.uint64(2000001)).astype(np.float64) - 1000000.0) * 0.001
    return np.ascontiguousarray(d.astype(dt))


def dist_counts(dist, P):
    kind, _, arg = dist.partition(":")
    if kind == "even":
        n = int(arg)
        return [n // P + (1 if r < n % P else 0) for r in range(P)]
    if kind == "same":
        return [int(arg)] * P
    if kind == "list":
        vals = [int(v) for v in arg.split(",")]
        return [vals[r % len(vals)] for r in range(P)]
    raise ValueError(dist)


def parse(path):
    cmds = []
    with open(path) as f:
        for lineno, line in enumerate(f, 1):
            line = line.split("#", 1)[0]
            tok = line.split()
            if tok:
                cmds.append((lineno, tok))
    return cmds


def scenario_path(name):
    return os.path.join(SCENARIOS, name + ".scn")


def golden_cases():
    """[(scenario, P)] for every committed golden file."""
    import re
    out = []
    for fn in sorted(os.listdir(GOLDEN)):
        m = re.match(r"^([a-z_]+)\.p(\d+)\.gsd$", fn)
        if m and os.path.exists(scenario_path(m.group(1))):
            out.append((m.group(1), int(m.group(2))))
    return out


# ----------------------------------------------------------------------------- oracle
_oracle = None


class OracleHeader(ctypes.Structure):
    _fields_ = [("magic", ctypes.c_uint64), ("index_location", ctypes.c_uint64),
                ("index_allocated_entries", ctypes.c_uint64), ("namelist_location", ctypes.c_uint64),
                ("namelist_allocated_entries", ctypes.c_uint64), ("schema_version", ctypes.c_uint32),
                ("pgsd_version", ctypes.c_uint32), ("application", ctypes.c_char * 64),
                ("schema", ctypes.c_char * 64), ("reserved", ctypes.c_char * 80)]


class OracleIndexEntry(ctypes.Structure):
    _fields_ = [("frame", ctypes.c_uint64), ("N", ctypes.c_uint64), ("location", ctypes.c_int64),
                ("M", ctypes.c_uint32), ("id", ctypes.c_uint16), ("type", ctypes.c_uint8),
                ("flags", ctypes.c_uint8)]


def oracle_lib():
    """Load (building if needed) oracle/libpgsd_oracle.so."""
    global _oracle
    if _oracle is not None:
        return _oracle
    so = os.path.join(ROOT, "oracle", "libpgsd_oracle.so")
    src = os.path.join(ROOT, "oracle", "pgsd_oracle.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        from product import locked_make
        locked_make(["-C", os.path.join(ROOT, "oracle"), "oracle"], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(so)
    vp, u64, u32, i32 = ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int
    lib.oracle_create_and_open.restype = vp
    lib.oracle_create_and_open.argtypes = [ctypes.c_char_p, i32, ctypes.c_char_p, ctypes.c_char_p,
                                           u32, i32, i32, ctypes.POINTER(i32)]
    lib.oracle_open.restype = vp
    lib.oracle_open.argtypes = [ctypes.c_char_p, i32, i32, ctypes.POINTER(i32)]
    for fn in ("oracle_close", "oracle_end_frame", "oracle_flush"):
        getattr(lib, fn).restype = i32
        getattr(lib, fn).argtypes = [vp]
    lib.oracle_write_chunk.restype = i32
    lib.oracle_write_chunk.argtypes = [vp, ctypes.c_char_p, i32, ctypes.POINTER(u64), u32, u64, u32,
                                       ctypes.POINTER(u64), ctypes.POINTER(u64), ctypes.c_bool,
                                       ctypes.c_uint8, ctypes.POINTER(vp)]
    lib.oracle_find_chunk.restype = ctypes.POINTER(OracleIndexEntry)
    lib.oracle_find_chunk.argtypes = [vp, u64, ctypes.c_char_p]
    lib.oracle_read_chunk.restype = i32
    lib.oracle_read_chunk.argtypes = [vp, vp, ctypes.POINTER(OracleIndexEntry), u64, u32, u32,
                                      ctypes.c_bool]
    lib.oracle_find_matching_chunk_name.restype = vp
    lib.oracle_find_matching_chunk_name.argtypes = [vp, ctypes.c_char_p, vp]
    for fn in ("oracle_get_nframes", "oracle_get_nnames", "oracle_get_maximum_write_buffer_size",
               "oracle_get_index_entries_to_buffer"):
        getattr(lib, fn).restype = u64
        getattr(lib, fn).argtypes = [vp]
    lib.oracle_get_file_size.restype = ctypes.c_longlong
    lib.oracle_get_file_size.argtypes = [vp]
    lib.oracle_get_header.restype = ctypes.POINTER(OracleHeader)
    lib.oracle_get_header.argtypes = [vp]
    lib.oracle_set_maximum_write_buffer_size.restype = i32
    lib.oracle_set_maximum_write_buffer_size.argtypes = [vp, u64]
    lib.oracle_set_index_entries_to_buffer.restype = i32
    lib.oracle_set_index_entries_to_buffer.argtypes = [vp, u64]
    lib.oracle_make_version.restype = u32
    lib.oracle_make_version.argtypes = [ctypes.c_uint, ctypes.c_uint]
    lib.oracle_pack_rows.restype = i32
    lib.oracle_pack_rows.argtypes = [vp, i32, vp, i32, u64, u32, u32, u32, vp, i32]
    _oracle = lib
    return lib


def oracle_write_chunk(lib, h, name, type_id, arrays, M, N_global, M_global, offsets, global_sizes,
                       all_):
    """One collective write_chunk through the oracle; `arrays` is a per-rank list of numpy arrays."""
    P = len(arrays)
    u64 = ctypes.c_uint64
    Ns = (u64 * P)(*[a.shape[0] for a in arrays])
    offs = (u64 * P)(*offsets)
    gss = (u64 * P)(*global_sizes)
    keep = [np.ascontiguousarray(a) for a in arrays]
    ptrs = (ctypes.c_void_p * P)(*[a.ctypes.data if a.size else None for a in keep])
    return lib.oracle_write_chunk(h, name.encode(), type_id, Ns, M, N_global, M_global, offs, gss,
                                  bool(all_), 0, ptrs)


def _dump_line(lineno, nf, nn, fs, hdr):
    return ("dump line=%d nframes=%d nnames=%d file_size=%d index_location=%d index_allocated=%d"
            " namelist_location=%d namelist_allocated=%d"
            % (lineno, nf, nn, fs, hdr.index_location, hdr.index_allocated_entries,
               hdr.namelist_location, hdr.namelist_allocated_entries))


def run_oracle(script, out_path, P):
    """Replay `script` through the oracle for P simulated ranks. Returns the log lines."""
    lib = oracle_lib()
    log = []
    h = None
    seed = 1
    rc_ref = ctypes.c_int(0)
    for lineno, tok in parse(script):
        cmd = tok[0]
        rc = 0
        if cmd == "create":
            if os.path.exists(out_path) and not int(tok[6]):
                pass
            h = lib.oracle_create_and_open(out_path.encode(), P, tok[1].encode(), tok[2].encode(),
                                           lib.oracle_make_version(int(tok[3]), int(tok[4])),
                                           FLAGS[tok[5]], int(tok[6]), ctypes.byref(rc_ref))
            rc = rc_ref.value
        elif cmd == "prefill":
            import shutil
            src = tok[1] if os.path.isabs(tok[1]) else os.path.join(os.path.dirname(os.path.abspath(script)), "..", "files", tok[1])
            shutil.copyfile(src, out_path)
        elif cmd == "open":
            h = lib.oracle_open(out_path.encode(), P, FLAGS[tok[1]], ctypes.byref(rc_ref))
            rc = rc_ref.value
        elif cmd == "seed":
            seed = int(tok[1])
        elif cmd == "batch":
            pass    # product only (pgsd_set_frame_exchange): the layout must not depend on it
        elif cmd == "chunk":
            name, t, M, all_, dist = tok[1], TYPE_IDS[tok[2]], int(tok[3]), int(tok[4]), tok[5]
            counts = dist_counts(dist, P)
            if all_:
                Ng = sum(counts)
                row0 = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int)
                arrays = [gen_data(t, seed, int(row0[r]), counts[r], M) for r in range(P)]
                rc = oracle_write_chunk(lib, h, name, t, arrays, M, Ng, M,
                                        [int(row0[r]) * M for r in range(P)], [Ng * M] * P, True)
            else:
                arrays = [gen_data(t, seed, 0, counts[r], M) for r in range(P)]
                # each rank passes N_global = its own N (fl.pyx:594); entry uses root's
                rc = oracle_write_chunk(lib, h, name, t, arrays, M, counts[0], M, [0] * P,
                                        [counts[r] * M for r in range(P)], False)
        elif cmd == "chunkgs":
            name, t, M, all_, dist, gs = tok[1], TYPE_IDS[tok[2]], int(tok[3]), int(tok[4]), tok[5], int(tok[6])
            counts = dist_counts(dist, P)
            Ng = sum(counts)
            row0 = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(int)
            arrays = [gen_data(t, seed, int(row0[r]), counts[r], M) for r in range(P)]
            rc = oracle_write_chunk(lib, h, name, t, arrays, M, Ng, M, [int(row0[r]) * M for r in range(P)],
                                    [gs] * P, all_)
        elif cmd in ("rawchunk", "samechunk"):
            name, t = tok[1], TYPE_IDS[tok[2]]
            N, M, Ng, Mg, off, gs, all_ = (int(x) for x in tok[3:10])
            arrays = [gen_data(t, seed if cmd == "samechunk" else seed + r, 0, N, M) for r in range(P)]
            rc = oracle_write_chunk(lib, h, name, t, arrays, M, Ng, Mg, [off] * P, [gs] * P, all_)
        elif cmd == "end_frame":
            rc = lib.oracle_end_frame(h)
        elif cmd == "flush":
            rc = lib.oracle_flush(h)
        elif cmd == "close":
            rc = lib.oracle_close(h)
            h = None
        elif cmd == "maxbuf":
            rc = lib.oracle_set_maximum_write_buffer_size(h, int(tok[1]))
        elif cmd == "idxbuf":
            rc = lib.oracle_set_index_entries_to_buffer(h, int(tok[1]))
        elif cmd == "dump":
            log.append(_dump_line(lineno, lib.oracle_get_nframes(h), lib.oracle_get_nnames(h),
                                  lib.oracle_get_file_size(h), lib.oracle_get_header(h).contents))
        elif cmd == "find":
            e = lib.oracle_find_chunk(h, int(tok[1]), tok[2].encode())
            if e:
                e = e.contents
                log.append("find line=%d frame=%s name=%s N=%d M=%d type=%d location=%d id=%d"
                           % (lineno, tok[1], tok[2], e.N, e.M, e.type, e.location, e.id))
            else:
                log.append("find line=%d frame=%s name=%s NOTFOUND" % (lineno, tok[1], tok[2]))
        elif cmd == "read":
            frame, name = int(tok[1]), tok[2]
            N, M, off, all_ = int(tok[3]), int(tok[4]), int(tok[5]), int(tok[6])
            nbytes = int(tok[7]) * int(tok[8]) * int(tok[9])
            buf = np.zeros(max(nbytes, 1), dtype=np.uint8)
            e = lib.oracle_find_chunk(h, frame, name.encode())
            rrc = lib.oracle_read_chunk(h, buf.ctypes.data, e, N, M, off, bool(all_)) if e else -2
            got = 0
            if rrc == 0 and e:
                esz = np.dtype(NP_TYPES[e.contents.type]).itemsize
                got = min((N * M if all_ else e.contents.N * e.contents.M) * esz, nbytes)
            hv = 0xcbf29ce484222325
            for byte in buf[:got].tolist():
                hv = ((hv ^ byte) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
            log.append("read line=%d frame=%s name=%s N=%s M=%s offset=%s all=%d rc=%d bytes=%d fnv=%016x"
                       % (lineno, tok[1], name, tok[3], tok[4], tok[5], all_, rrc, got, hv))
        elif cmd == "names":
            prefix = tok[1] if len(tok) > 1 else ""
            found = []
            p = lib.oracle_find_matching_chunk_name(h, prefix.encode(), None)
            while p:
                found.append(ctypes.string_at(p).decode())
                p = lib.oracle_find_matching_chunk_name(h, prefix.encode(), p)
            log.append("names line=%d prefix=%s:%s" % (lineno, prefix, "".join(" " + n for n in found)))
        else:
            raise ValueError("line %d: bad command %r" % (lineno, tok))
        if rc != 0:
            log.append("rc line=%d cmd=%s rc=%d" % (lineno, cmd, rc))
    return log


def read_log(path):
    with open(path) as f:
        return [ln.rstrip("\n") for ln in f if ln.strip()]
