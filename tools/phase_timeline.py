#!/usr/bin/env python3
"""Phase timeline of one frame from a rocprofv3 run with the library's roctx ranges on (PGSD_TRACE=1):

    cd /tmp && PGSD_TRACE=1 rocprofv3 --marker-trace --kernel-trace --memory-copy-trace --output-format csv \
        -d <dir> -- python3 <repo>/bench.py --steps 5 --warmup 2 --no-cpu-baseline --traffic off --no-stall-test --no-exchange-probe
    python tools/phase_timeline.py <dir> [frame]

Prints, for the chosen frame (default: the last but one `pgsd:end_frame`), every pgsd:* range (host threads), every
kernel and every device<->host copy on one time axis (us from the frame's pack launch), and how much of the frame's
pwrite time ran while a device->host copy was in flight (the overlap DESIGN section 4 describes)."""
import csv
import glob
import os
import sys


def rows(d, pattern):
    out = []
    for p in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(p, newline="") as f:
            out += list(csv.DictReader(f))
    return out


def main():
    d = sys.argv[1]
    markers = rows(d, "*marker_api_trace.csv")
    kernels = rows(d, "*kernel_trace.csv")
    copies = rows(d, "*memory_copy_trace.csv")
    ev = []
    for r in markers:
        if r["Function"].startswith("pgsd:"):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "host tid %s" % r["Thread_Id"], r["Function"]))
    for r in kernels:
        # hipMemcpyAsync device -> pinned host runs as a blit kernel on these boxes (it shows in the kernel trace as
        # __amd_rocclr_copyBuffer, not in the memory-copy trace): counted as a copy
        blit = "copyBuffer" in r["Kernel_Name"]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "gpu copy" if blit else "gpu kernel",
                   r["Kernel_Name"][:70]))
    for r in copies:
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "gpu copy", r.get("Direction", "copy")))
    ends = sorted(e for e in ev if e[3].startswith("pgsd:end_frame"))
    if not ends:
        sys.exit("no pgsd:end_frame ranges: was PGSD_TRACE=1 set and --marker-trace given?")
    which = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 2
    frame = ends[which]
    stages = [e for e in ev if e[3].startswith("pgsd:stage") and e[0] <= frame[0]]
    t0 = stages[-1][0] if stages else frame[0]
    t1 = frame[1]
    inside = sorted(e for e in ev if e[0] >= t0 - 2000 and e[0] <= t1)
    print("# frame: %s   window %.1f us" % (frame[3], (t1 - t0) / 1e3))
    print("# %10s %10s  %-22s %s" % ("start_us", "dur_us", "where", "what"))
    for s, e, where, what in inside:
        print("  %10.1f %10.1f  %-22s %s" % ((s - t0) / 1e3, (e - s) / 1e3, where, what))
    pw = [(s, e) for s, e, w, n in inside if n.startswith("pgsd:pwrite")]
    cp = [(s, e) for s, e, w, n in inside if w == "gpu copy"]

    def overlap(a, bs):
        tot = 0
        for s, e in bs:
            tot += max(0, min(a[1], e) - max(a[0], s))
        return tot
    pw_total = sum(e - s for s, e in pw)
    pw_overl = sum(overlap(p, cp) for p in pw)
    cp_total = sum(e - s for s, e in cp)
    if pw and cp:
        print("# pwrite ranges: %d, %.1f us in all; device->host copies: %d, %.1f us in all" % (len(pw), pw_total / 1e3, len(cp), cp_total / 1e3))
        print("# copy time that ran UNDER a pwrite of an earlier piece: %.1f us = %.0f %% of the copy time"
              % (pw_overl / 1e3, 100.0 * pw_overl / max(cp_total, 1)))
        first_pw = min(s for s, e in pw)
        print("# first pwrite starts %.1f us after the pack launch; last pwrite ends at %.1f us; frame sealed at %.1f us"
              % ((first_pw - t0) / 1e3, (max(e for s, e in pw) - t0) / 1e3, (t1 - t0) / 1e3))


if __name__ == "__main__":
    main()
