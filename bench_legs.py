#!/usr/bin/env python3
"""bench_legs.py -- the other BASELINE configurations next to bench.py's headline (its `legs` array).

bench.py times BASELINE config 3's shape (10 M particles, position + velocity + typeid, pack + write).  The legs put
the remaining configurations into the same driver-run JSON line, each with the kernel's average duration, its
algorithmic bytes and the fraction of the HBM peak, as the reference's harnesses print every leg they run
(pgsd/scripts/benchmark-write.cc:144-172, benchmark-read.cc:128-146):

  config2          2^20 particles, pack kernel only, SURVEY 8(d)'s protocol: 200 launches after 20 warm-ups over 11
                   buffer sets (> 256 MiB between two uses of a set), dispatch-stamped; the same WITHOUT rotation (what
                   a simulation that has just written the arrays sees: Infinity-Cache resident); the double4 variant
  config4_sph      the 14 per-particle chunks of the PGSD-SPH schema (112 B/particle), 3 timed frames to --dir
  config4_union    ... plus the upstream HOOMD attributes (164 B/particle)
  config5_read     one 80 M-particle frame, rows [35 M, 45 M) read into Scalar4 arrays on the GPU (file -> HBM GB/s,
                   bit-exact against the source rows) and the fused unpack kernel alone
  gather_uniform   position + typeid + velocity of 10 M particles in tag order through a uniformly random reverse-tag
                   array (one 64-byte sector per 16-byte row: the adversarial case)
  gather_hilbert   the same through a REALISTIC permutation: particles created in lattice order (tag = lattice index)
                   and kept in memory along a 3-D Hilbert curve (what HOOMD's SFC sorter leaves a dump writer)

`traffic` of a leg = HBM bytes per launch from two rocprofv3 counter passes (FETCH_SIZE, WRITE_SIZE; separate passes,
KiB units, FETCH_SIZE doubled on gfx950 for wide coalesced reads: MI355X_MICROARCH.md, HBM) over a child run of this
file that launches every leg's kernels a few times, the legs separated by a marker kernel.
"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))

HBM_PEAK_GBS = 8000.0
F32, U32, I32, F64 = 9, 3, 7, 10          # enum pgsd_type (include/pgsd.h)


# ------------------------------------------------------------------------------------------ workloads
def make_fields(schema, N, g, torch, np, fl, separate_id=False):
    """HBM-resident particle arrays of one rank and the chunks they feed.
    -> (fields [(chunk name, DeviceField)], payload B/particle, algorithmic B/particle, layout text, keepalive)"""
    pos = (torch.rand((N, 4), generator=g, device="cuda") - 0.5) * 100.0
    vel = torch.randn((N, 4), generator=g, device="cuda")
    tid = torch.randperm(N, generator=g, device="cuda").to(torch.int32)
    keep = [pos, vel, tid]
    # HOOMD's own device layout (ParticleData: Scalar4 pos = x, y, z, __int_as_scalar(type); Scalar4 vel = vx, vy,
    # vz, mass).  separate_id keeps the id in its own uint32 array as SURVEY.md 8(d) sketches; 56 algorithmic bytes
    # per particle either way.
    if separate_id:
        id_field = fl.DeviceField.from_tensor(tid, out_dtype=np.uint32)
        layout = "float4 pos, float4 vel, separate uint32 id array"
    else:
        pos[:, 3] = tid.view(torch.float32)
        id_field = fl.DeviceField.from_tensor(pos, columns=(3, 4), out_dtype=np.uint32, bitcast=True)
        layout = "HOOMD Scalar4 arrays: pos=(x,y,z,typeid bits), vel=(vx,vy,vz,mass)"
    fields = [("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
              ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3))),
              ("particles/typeid", id_field)]
    payload_bpp, algo_bpp = 28, 56
    if schema in ("sph", "union"):
        # hoomd.py:167-184: typeid, mass, body, position, velocity, slength, density, pressure, energy,
        # auxiliary1-4, image -- from HOOMD-SPH-style device arrays (Scalar4 / int3-as-int4 / int)
        dpe = torch.rand((N, 4), generator=g, device="cuda")                 # density, pressure, energy, slength
        aux = [torch.randn((N, 4), generator=g, device="cuda") for _ in range(4)]
        img = torch.randint(-2, 3, (N, 4), generator=g, device="cuda", dtype=torch.int32)
        body = torch.full((N,), -1, device="cuda", dtype=torch.int32)
        keep += [dpe, img, body] + aux
        fields = [("particles/typeid", id_field),
                  ("particles/mass", fl.DeviceField.from_tensor(vel, columns=(3, 4))),
                  ("particles/body", fl.DeviceField.from_tensor(body)),
                  ("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
                  ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3))),
                  ("particles/slength", fl.DeviceField.from_tensor(dpe, columns=(3, 4))),
                  ("particles/density", fl.DeviceField.from_tensor(dpe, columns=(0, 1))),
                  ("particles/pressure", fl.DeviceField.from_tensor(dpe, columns=(1, 2))),
                  ("particles/energy", fl.DeviceField.from_tensor(dpe, columns=(2, 3)))]
        fields += [("particles/auxiliary%d" % (k + 1), fl.DeviceField.from_tensor(a, columns=(0, 3)))
                   for k, a in enumerate(aux)]
        fields.append(("particles/image", fl.DeviceField.from_tensor(img, columns=(0, 3))))
        payload_bpp, algo_bpp = 112, 224
        layout = "HOOMD-SPH device arrays (Scalar4 pos/vel/dpe/aux1-4, int4 image, int body), full SPH schema"
    if schema == "union":
        # BASELINE config 4: the SPH set plus the upstream HOOMD particle attributes (pgsd.tex:508-521):
        # charge, diameter (Scalar arrays), moment_inertia (Scalar3), orientation, angmom (Scalar4)
        charge = torch.randn((N,), generator=g, device="cuda")
        diameter = torch.rand((N,), generator=g, device="cuda")
        inertia = torch.rand((N, 3), generator=g, device="cuda")
        orient = torch.randn((N, 4), generator=g, device="cuda")
        angmom = torch.randn((N, 4), generator=g, device="cuda")
        keep += [charge, diameter, inertia, orient, angmom]
        fields += [("particles/charge", fl.DeviceField.from_tensor(charge)),
                   ("particles/diameter", fl.DeviceField.from_tensor(diameter)),
                   ("particles/moment_inertia", fl.DeviceField.from_tensor(inertia)),
                   ("particles/orientation", fl.DeviceField.from_tensor(orient)),
                   ("particles/angmom", fl.DeviceField.from_tensor(angmom))]
        payload_bpp, algo_bpp = 164, 328
        layout = "HOOMD-SPH device arrays + upstream HOOMD attributes (charge, diameter, moment_inertia, orientation, angmom)"
    return fields, payload_bpp, algo_bpp, layout, keep


def hilbert_order(N, torch, device="cuda"):
    """Reverse-tag array of N particles created in lattice order (tag t sits at lattice site (t % n, t / n % n,
    t / n^2), n = ceil(cbrt N)) and kept in memory along the 3-D Hilbert curve through their sites (Skilling's
    transform, AIP Conf. Proc. 707, 2004): order[t] = memory row of tag t.  int32 GPU tensor."""
    n = 1
    while n * n * n < N:
        n += 1
    bits = max(1, (n - 1).bit_length())
    t = torch.arange(N, device=device, dtype=torch.int64)
    X = [t % n, (t // n) % n, t // (n * n)]
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(3):
            cond = (X[i] & Q) != 0
            tt = (X[0] ^ X[i]) & P
            x0 = torch.where(cond, X[0] ^ P, X[0] ^ tt)
            if i != 0:
                X[i] = torch.where(cond, X[i], X[i] ^ tt)
            X[0] = x0
        Q >>= 1
    X[1] = X[1] ^ X[0]
    X[2] = X[2] ^ X[1]
    tt = torch.zeros_like(t)
    Q = M
    while Q > 1:
        tt = torch.where((X[2] & Q) != 0, tt ^ (Q - 1), tt)
        Q >>= 1
    X = [x ^ tt for x in X]
    key = torch.zeros_like(t)
    for b in range(bits - 1, -1, -1):
        for i in range(3):
            key = (key << 1) | ((X[i] >> b) & 1)
    tag_of = torch.argsort(key)                       # memory row i holds tag tag_of[i]
    order = torch.empty(N, device=device, dtype=torch.int64)
    order[tag_of] = t
    return order.to(torch.int32)


def pack_jobs(fields, N, torch, np, _lib, order=None):
    """ctypes pgsd_pack_job array for `fields` into fresh dense chunk buffers -> (array, keepalive)."""
    from pgsd.fl import _pgsd_type
    arr = (_lib.PackJob * len(fields))()
    keep = []
    for i, (_name, f) in enumerate(fields):
        dt = np.dtype(f.out_dtype)
        dst = torch.empty((N * f.M * dt.itemsize,), dtype=torch.uint8, device="cuda")
        keep.append(dst)
        arr[i].dst = dst.data_ptr()
        arr[i].dst_type = _pgsd_type(dt)
        arr[i].M = f.M
        arr[i].src = f._desc()
        if order is not None:
            arr[i].src.order = order.data_ptr()
    return arr, keep


def timed_pack(arr_sets, N, launches, warmup, _lib):
    """Dispatch-stamped kernel time (what rocprofv3 reports per kernel) of `launches` pack launches, rotating over
    the job arrays of arr_sets.  -> list of ms"""
    ms = ctypes.c_float(0)
    out = []
    for i in range(warmup + launches):
        arr = arr_sets[i % len(arr_sets)]
        rc = _lib.lib.pgsd_pack_fields(len(arr), arr, N, None, ctypes.byref(ms))
        if rc != 0:
            raise RuntimeError("pgsd_pack_fields: %d %s" % (rc, _lib.last_error()))
        if i >= warmup:
            out.append(ms.value)
    return out


def leg_entry(name, kernel, ms_list, algo_bytes, **more):
    import statistics
    avg = sum(ms_list) / len(ms_list)
    # compact on purpose: the driver keeps a bounded tail of stdout; what each leg IS stands in this file's docstring
    d = {"name": name, "kernel": kernel, "launches": len(ms_list), "avg_us": round(avg * 1e3, 2),
         "median_us": round(statistics.median(ms_list) * 1e3, 2), "algorithmic_bytes": int(algo_bytes),
         "frac": round(algo_bytes / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": None}
    d.update(more)
    return d


# ------------------------------------------------------------------------------------------ the legs
def leg_config2(torch, np, fl, _lib):
    N = 1 << 20
    g = torch.Generator(device="cuda").manual_seed(1234)
    n_sets = 11                                       # 11 x 64 MiB moved per launch: > 256 MiB between two uses
    sets, keep = [], []
    for _ in range(n_sets):
        fields, _, _, _, k = make_fields("pvi", N, g, torch, np, fl, separate_id=True)
        arr, k2 = pack_jobs(fields, N, torch, np, _lib)
        sets.append(arr)
        keep += [k, k2]
    rot = timed_pack(sets, N, 200, 20, _lib)
    unrot = timed_pack(sets[:1], N, 200, 20, _lib)
    # the layout HOOMD itself keeps (type id in position.w: two source arrays instead of three), rotated likewise
    hsets = []
    for _ in range(n_sets):
        fields, _, _, _, k = make_fields("pvi", N, g, torch, np, fl)
        arr, k2 = pack_jobs(fields, N, torch, np, _lib)
        hsets.append(arr)
        keep += [k, k2]
    hoomd = timed_pack(hsets, N, 200, 20, _lib)
    # the double4 variant (Scalar = double builds): f64 -> f32 in registers, 68 + 28 B/particle moved
    dsets = []
    for _ in range(7):
        pos = torch.randn((N, 4), generator=g, device="cuda", dtype=torch.float64)
        vel = torch.randn((N, 4), generator=g, device="cuda", dtype=torch.float64)
        tid = torch.randperm(N, generator=g, device="cuda").to(torch.int32)
        f = [("p", fl.DeviceField.from_tensor(pos, columns=(0, 3), out_dtype=np.float32)),
             ("v", fl.DeviceField.from_tensor(vel, columns=(0, 3), out_dtype=np.float32)),
             ("i", fl.DeviceField.from_tensor(tid, out_dtype=np.uint32))]
        arr, k2 = pack_jobs(f, N, torch, np, _lib)
        dsets.append(arr)
        keep += [pos, vel, tid, k2]
    dbl = timed_pack(dsets, N, 100, 10, _lib)
    import statistics
    def us(x):
        return round(sum(x) / len(x) * 1e3, 2)

    # protocol: 200 launches after 20 warm-ups, 11 buffer sets rotated (704 MiB: no byte re-read from the
    # 256 MiB Infinity Cache), dispatch-stamped.  unrotated: the same on ONE set (Infinity-Cache resident: what a simulation
    # that has just written the arrays sees).  hoomd_layout: type id kept in position.w (two source arrays, 60 B moved per
    # particle).  double4: double4 pos / vel written as float32 chunks (80 algorithmic B/particle), 7 sets rotated.
    e = leg_entry("config2", "pack_rows_kernel", rot, 56 * N, rows=N, buffer_sets=n_sets, target_frac=0.70,
                  unrotated_us=us(unrot), hoomd_layout_us=us(hoomd),
                  hoomd_layout_frac=round(56 * N / (us(hoomd) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                  double4_us=us(dbl), double4_frac=round(80 * N / (us(dbl) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4))
    del keep
    return e


def run_frames(schema, N, steps, warmup, out_dir, torch, np, fl, tag):
    """`steps` timed frames of one rank through the file path (the bench's step, single rank) -> dict."""
    g = torch.Generator(device="cuda").manual_seed(1234)
    fields, payload_bpp, algo_bpp, layout, keep = make_fields(schema, N, g, torch, np, fl)
    path = os.path.join(out_dir, "pgsd_bench_leg_%s_%d.gsd" % (tag, os.getpid()))
    f = fl.open(path, "w", application="pgsd_amd bench", schema="hoomd", schema_version=[1, 4])
    try:
        f.configure_device(profile=True)
        f.frame_exchange = True

        def step(i):
            f.write_chunk("configuration/step", np.array([i], dtype=np.uint64), write_all=False)
            f.write_chunks(fields, offset="auto")
            f.end_frame()

        for i in range(warmup):
            step(i)
        f.device_stats(reset=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(warmup + i)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        stats = f.device_stats()
    finally:
        f.close()
        try:
            os.unlink(path)
        except OSError:
            pass
    del keep
    return {"dt": dt, "pack_ms_per_frame": stats["pack_ms"] / max(steps, 1),
            "kernels_per_frame": stats["pack_launches"] / max(steps, 1), "payload_bpp": payload_bpp, "algo_bpp": algo_bpp,
            "layout": layout, "d2h_GBps": stats["d2h_bytes"] / max(stats["d2h_ms"], 1e-9) / 1e6}


def leg_config4(schema, N, out_dir, fstype, torch, np, fl):
    r = run_frames(schema, N, 3, 1, out_dir, torch, np, fl, schema)
    # (BASELINE config 4 names an NVMe target: these boxes have none, the file goes where the headline's goes)
    e = leg_entry("config4_" + schema, "pack_rows_kernel", [r["pack_ms_per_frame"]], r["algo_bpp"] * N,
                  rows=N, value_GBps=round(3 * N * r["payload_bpp"] / r["dt"] / 1e9, 3), frames=3,
                  ms_per_frame=round(r["dt"] / 3 * 1e3, 2), target_fstype=fstype)
    e["launches"] = 3
    return e


def leg_config5(out_dir, torch, np, fl, _lib, N_file=80_000_000, row0=35_000_000, n_read=10_000_000):
    g = torch.Generator(device="cuda").manual_seed(5)
    pos = torch.randn((N_file, 4), generator=g, device="cuda")
    vel = torch.randn((N_file, 4), generator=g, device="cuda")
    path = os.path.join(out_dir, "pgsd_bench_leg_read_%d.gsd" % os.getpid())
    f = fl.open(path, "w", application="pgsd_amd bench", schema="hoomd", schema_version=[1, 4])
    try:
        f.write_chunks([("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
                        ("particles/typeid", fl.DeviceField.from_tensor(pos, columns=(3, 4), out_dtype=np.uint32, bitcast=True)),
                        ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3)))], offset=np.array([N_file]))
        f.end_frame()
        f.close()
        r = fl.open(path, "r")
        pos4 = torch.zeros((n_read, 4), dtype=torch.float32, device="cuda")
        vel4 = torch.zeros((n_read, 4), dtype=torch.float32, device="cuda")
        times = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r.read_chunk_device(0, "particles/position", out=pos4, N=n_read, offset=row0, columns=(0, 3), wait=False)
            r.read_chunk_device(0, "particles/typeid", out=pos4, N=n_read, offset=row0, columns=(3, 4), bitcast=True, wait=False)
            r.read_chunk_device(0, "particles/velocity", out=vel4, N=n_read, offset=row0, columns=(0, 3), wait=False, fill=1.0)
            r.wait_read()
            times.append(time.perf_counter() - t0)
        r.close()
        exact = bool(torch.equal(pos4.view(torch.int32), pos[row0:row0 + n_read].view(torch.int32))
                     and torch.equal(vel4[:, :3].contiguous().view(torch.int32),
                                     vel[row0:row0 + n_read, :3].contiguous().view(torch.int32))
                     and bool((vel4[:, 3] == 1.0).all()))
    finally:
        try:
            os.unlink(path)
        except OSError:
            pass
    del pos, vel
    ms, keep = unpack_launches(n_read, 20, 3, torch, _lib)
    best = min(times[1:])
    # one frame of N_file particles (position, typeid, velocity), rows [row0, row0 + n_read) -> Scalar4 pos (x, y, z, typeid
    # bits) and Scalar4 vel (vx, vy, vz, 1.0) on the GPU; unpack kernel: stream events around 20 back-to-back launches
    e = leg_entry("config5_read", "unpack_rows_kernel", ms, 56 * n_read, rows=n_read, file_rows=N_file,
                  file_to_hbm_GBps=round(28 * n_read / best / 1e9, 2), bit_exact=exact)
    del keep
    return e


def unpack_jobs(N, torch, _lib, g):
    cpos = torch.randn((N, 3), generator=g, device="cuda")
    cvel = torch.randn((N, 3), generator=g, device="cuda")
    ctid = torch.randint(0, 8, (N,), generator=g, device="cuda", dtype=torch.int32)
    pos4 = torch.empty((N, 4), dtype=torch.float32, device="cuda")
    vel4 = torch.empty((N, 4), dtype=torch.float32, device="cuda")
    jobs = (_lib.UnpackJob * 3)()
    for i, (src, st, M, dst, c0, bitcast, fill) in enumerate(((cpos, F32, 3, pos4, 0, 0, None), (ctid, U32, 1, pos4, 3, 1, None),
                                                               (cvel, F32, 3, vel4, 0, 0, 0x3F800000))):
        jobs[i].src = src.data_ptr()
        jobs[i].src_type = st
        jobs[i].M = M
        jobs[i].dst.dst = dst.data_ptr()
        jobs[i].dst.dst_type = F32
        jobs[i].dst.dst_stride = 4
        jobs[i].dst.dst_col0 = c0
        jobs[i].dst.bitcast = bitcast
        if fill is not None:
            jobs[i].dst.fill_rest = 1
            jobs[i].dst.fill_bits = fill
    return jobs, [cpos, cvel, ctid, pos4, vel4]


def unpack_launches(N, launches, warmup, torch, _lib):
    g = torch.Generator(device="cuda").manual_seed(6)
    sets = [unpack_jobs(N, torch, _lib, g) for _ in range(2)]
    stream = torch.cuda.current_stream().cuda_stream

    def go(k):
        rc = _lib.lib.pgsd_unpack_fields(3, sets[k % 2][0], N, ctypes.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("pgsd_unpack_fields: %d %s" % (rc, _lib.last_error()))

    for k in range(warmup):
        go(k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(launches):
        go(k)
    e1.record()
    torch.cuda.synchronize()
    return [e0.elapsed_time(e1) / launches] * launches, sets


def gather_setup(N, perm, torch, np, fl):
    g = torch.Generator(device="cuda").manual_seed(77)
    pos = (torch.rand((N, 4), generator=g, device="cuda") - 0.5) * 100.0
    vel = torch.randn((N, 4), generator=g, device="cuda")
    pos[:, 3] = torch.randint(0, 8, (N,), generator=g, device="cuda", dtype=torch.int32).view(torch.float32)
    order = torch.randperm(N, generator=g, device="cuda").to(torch.int32) if perm == "uniform" else hilbert_order(N, torch)
    fields = [("particles/position", fl.DeviceField.from_tensor(pos, columns=(0, 3))),
              ("particles/typeid", fl.DeviceField.from_tensor(pos, columns=(3, 4), out_dtype=np.uint32, bitcast=True)),
              ("particles/velocity", fl.DeviceField.from_tensor(vel, columns=(0, 3)))]
    return fields, order, [pos, vel]


def leg_gather(perm, N, torch, np, fl, _lib):
    fields, order, keep = gather_setup(N, perm, torch, np, fl)
    arr, k2 = pack_jobs(fields, N, torch, np, _lib, order=order)
    ms = timed_pack([arr], N, 20, 3, _lib)
    # position + typeid + velocity in TAG order, chunk[t] = src[order[t]]; algorithmic = 28 B of source columns + 4 B of
    # index + 28 B of chunks per particle.  uniform: adversarial (every 16-byte row in a 64-byte sector of its own);
    # hilbert: lattice-order tags over Hilbert-curve memory order (what HOOMD's SFC sorter leaves a dump writer)
    e = leg_entry("gather_" + perm, "pack_tiles_kernel", ms, 60 * N, rows=N)
    del keep, k2
    return e


# ------------------------------------------------------------------------------------------ counter passes
LEG_ORDER = ["config2", "config4_sph", "config4_union", "config5_read", "gather_uniform", "gather_hilbert"]
MARKER = "select_count_kernel"        # pgsd_select_rows: no leg uses it; one call separates two legs in the trace


def pmc_child(out_path, N, reps=3):
    """Run under rocprofv3 (--pmc passes of add_traffic; --kernel-trace --stats of tools/profile_legs.sh): every leg's
    kernels `reps` times, a marker launch in front of each leg."""
    import numpy as np
    import torch
    import pgsd.fl as fl
    from pgsd import _lib
    torch.cuda.set_device(0)
    flags = torch.ones(4096, dtype=torch.uint8, device="cuda")
    launches = {}
    g = torch.Generator(device="cuda").manual_seed(1234)

    def marker():
        torch.cuda.synchronize()
        fl.select_rows(flags)
        torch.cuda.synchronize()

    for leg in LEG_ORDER:
        if leg == "config2":
            n = 1 << 20
            keep, k2, sets = [], [], []
            for _ in range(11):                # rotated like the leg itself: no launch re-reads the Infinity Cache
                fields, _, _, _, k = make_fields("pvi", n, g, torch, np, fl, separate_id=True)
                arr, kk = pack_jobs(fields, n, torch, np, _lib)
                sets.append(arr)
                keep.append(k)
                k2.append(kk)
            run = lambda: timed_pack(sets, n, reps, 0, _lib)
        elif leg in ("config4_sph", "config4_union"):
            fields, _, _, _, keep = make_fields(leg.split("_")[1], N, g, torch, np, fl)
            arr, k2 = pack_jobs(fields, N, torch, np, _lib)
            run = lambda: timed_pack([arr], N, reps, 0, _lib)
        elif leg == "config5_read":
            run = lambda: unpack_launches(N, reps, 0, torch, _lib)
            keep = k2 = None
        else:
            fields, order, keep = gather_setup(N, leg.split("_")[1], torch, np, fl)
            arr, k2 = pack_jobs(fields, N, torch, np, _lib, order=order)
            run = lambda: timed_pack([arr], N, reps, 0, _lib)
        marker()
        run()
        torch.cuda.synchronize()
        launches[leg] = reps
        del keep, k2
    marker()
    with open(out_path, "w") as fh:
        json.dump({"launches": launches, "order": LEG_ORDER}, fh)


def pmc_pass(counter, N, timeout):
    """-> {leg: counter value (KiB) summed over the leg's pgsd kernels, per launch} or None."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    d = tempfile.mkdtemp(prefix="pgsd_legs_pmc_", dir="/tmp")
    try:
        info = os.path.join(d, "child.json")
        # the program itself follows `--` (no env/bash hop: the profiler's library has the GPU initialised)
        cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
               sys.executable, os.path.abspath(__file__), "--pmc-child", info, "--particles", str(N)]
        env = {k: v for k, v in os.environ.items() if k != "PGSD_IO"}
        r = subprocess.run(cmd, timeout=timeout, capture_output=True, cwd="/tmp", env=dict(env, TMPDIR="/tmp"))
        if r.returncode != 0 or not os.path.exists(info):
            print("bench_legs: %s pass failed (rc %d): %s" % (counter, r.returncode, r.stderr.decode()[-400:]), file=sys.stderr)
            return None
        with open(info) as fh:
            child = json.load(fh)
        rows = []
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            with open(path, newline="") as fh:
                for row in csv.DictReader(fh):
                    if row.get("Counter_Name") == counter:
                        rows.append((int(row.get("Dispatch_Id", "0")), row.get("Kernel_Name", ""), float(row["Counter_Value"])))
        rows.sort()
        out, leg = {}, -1
        for _, name, val in rows:
            if MARKER in name:
                leg += 1
                continue
            if "pgsd_amd::" not in name or "select_" in name or not (0 <= leg < len(child["order"])):
                continue
            key = child["order"][leg]
            out[key] = out.get(key, 0.0) + val
        return {k: v / child["launches"][k] for k, v in out.items()} or None
    except Exception as e:  # time-out, unreadable output, ...
        print("bench_legs: %s pass failed (%s)" % (counter, e), file=sys.stderr)
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def add_traffic(legs, N):
    fetch = pmc_pass("FETCH_SIZE", N, 300)
    write = pmc_pass("WRITE_SIZE", N, 300) if fetch else None
    if not fetch or not write:
        return "no counter passes (rocprofv3 unavailable or failed)"
    for e in legs:
        k = e["name"]
        if k not in fetch or k not in write:
            continue
        wr = int(write[k] * 1024)
        if k.startswith("gather"):
            # random 16-byte rows: 64-byte requests, which FETCH_SIZE tallies whole; only the streamed index is halved.
            # Uncalibrated for this access shape (MI355X_MICROARCH.md, HBM): both readings are given.
            rd = int(fetch[k] * 1024)
            e["traffic"] = rd + wr
            e["fetch_KiB"], e["write_KiB"], e["fetch_x2"] = round(fetch[k], 1), round(write[k], 1), False
        else:
            rd = int(fetch[k] * 1024 * 2)
            e["traffic"] = rd + wr       # bytes = KiB * 1024; FETCH_SIZE x2 on gfx950 (MI355X_MICROARCH.md, HBM)
            e["fetch_KiB"], e["write_KiB"], e["fetch_x2"] = round(fetch[k], 1), round(write[k], 1), True
        e["traffic_over_algorithmic"] = round(e["traffic"] / e["algorithmic_bytes"], 3)
    return "live: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench_legs.py --pmc-child"


def flat_summary(legs):
    """The legs' essentials as flat scalars for bench.py's `roofline` object (the driver's record keeps the scalar
    members of `roofline`; nested objects of the line only survive in its bounded stdout tail):
    legs_<name>_us / _frac / _traffic_ratio."""
    out = {}
    for e in legs:
        if "error" in e:
            out["legs_%s_error" % e["name"]] = e["error"][:80]
            continue
        out["legs_%s_us" % e["name"]] = e["avg_us"]
        out["legs_%s_frac" % e["name"]] = e["frac"]
        if e.get("traffic_over_algorithmic") is not None:
            out["legs_%s_traffic_ratio" % e["name"]] = e["traffic_over_algorithmic"]
    return out


def run_legs(N, out_dir, fstype, traffic=True, only=None):
    """All legs on cuda:0 of this process -> (list of leg dicts, traffic source text)."""
    import numpy as np
    import torch
    import pgsd.fl as fl
    from pgsd import _lib
    legs = []
    todo = [("config2", lambda: leg_config2(torch, np, fl, _lib)),
            ("config4_sph", lambda: leg_config4("sph", N, out_dir, fstype, torch, np, fl)),
            ("config4_union", lambda: leg_config4("union", N, out_dir, fstype, torch, np, fl)),
            ("config5_read", lambda: leg_config5(out_dir, torch, np, fl, _lib, N_file=8 * N, row0=N * 7 // 2, n_read=N)),
            ("gather_uniform", lambda: leg_gather("uniform", N, torch, np, fl, _lib)),
            ("gather_hilbert", lambda: leg_gather("hilbert", N, torch, np, fl, _lib))]
    for name, fn in todo:
        if only and name not in only:
            continue
        try:
            e = fn()
        except Exception as ex:  # a leg that fails is reported, the headline stands
            e = {"name": name, "error": "%s: %s" % (type(ex).__name__, ex)}
        legs.append(e)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    src = None
    if traffic:
        t0 = time.perf_counter()
        src = add_traffic([e for e in legs if "error" not in e], N)
        src += " (%.0f s)" % (time.perf_counter() - t0)
    return legs, src


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc-child", default=None, help="internal: run every leg's kernels under a counter pass")
    ap.add_argument("--reps", type=int, default=3, help="--pmc-child: launches per leg")
    ap.add_argument("--particles", type=int, default=10_000_000)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--no-traffic", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    if a.pmc_child:
        pmc_child(a.pmc_child, a.particles, a.reps)
    else:
        legs, src = run_legs(a.particles, a.dir, None, traffic=not a.no_traffic,
                             only=[x for x in a.only.split(",") if x] or None)
        print(json.dumps({"legs": legs, "legs_traffic_source": src}))
