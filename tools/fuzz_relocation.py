#!/usr/bin/env python3
"""Relocation-heavy scenarios (the random scripts of tests/test_fuzz_parity.py rarely fill the 128-entry index): many
chunks per frame, many frames, index_entries_to_buffer, flushes in the middle of frames, close + re-open for append,
zero-row ranks, one-rank-only direct writes (logical file size ahead of the true end) -- replayed through the compiled
reference (under mpiexec), the oracle and the product in its four placement modes, with PGSD_CHECK_EOF=1 (the computed
end of file against fstat in every relocation) and without.  Build container only (needs oracle/_ref).

    python tools/fuzz_relocation.py [first_seed=1] [n_seeds=40]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pgsd-sph_amd"))
import product
import scenario as S
import test_fuzz_parity as F

make_script = F.make_relocation_script


if __name__ == "__main__":
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    strip = lambda lines: [re.sub(r"line=\d+ ", "", ln) for ln in lines]
    bad, ran, ref_fail, relocated = [], 0, 0, 0
    with tempfile.TemporaryDirectory() as tmp:
        for seed in range(first, first + count):
            for P in (1, 2, 3, 5, 8):
                scn = os.path.join(tmp, "r.scn")
                open(scn, "w").write(make_script(seed, P))
                o_path = os.path.join(tmp, "oracle.gsd")
                o_log = S.run_oracle(scn, o_path, P)
                if [ln for ln in o_log if ln.startswith("rc ")]:
                    continue
                want = open(o_path, "rb").read()
                relocated += any("index_allocated=" in ln and "index_allocated=128 " not in ln for ln in o_log)
                for check in ("", "1"):
                    os.environ.pop("PGSD_CHECK_EOF", None)
                    if check:
                        os.environ["PGSD_CHECK_EOF"] = "1"
                    for mode in (0, 1, 2, 3):
                        p_path = os.path.join(tmp, "p.gsd")
                        if os.path.exists(p_path):
                            os.unlink(p_path)
                        s2 = scn if mode == 0 else product.batched_script(scn, os.path.join(tmp, "b.scn"), mode)
                        log = product.run_driver(s2, p_path, P, threads=(P == 8))
                        if open(p_path, "rb").read() != want or strip(log) != strip(o_log):
                            bad.append((seed, P, mode, check))
                            print("MISMATCH product seed %d P %d mode %d check_eof %r" % (seed, P, mode, check), flush=True)
                os.environ.pop("PGSD_CHECK_EOF", None)
                if P <= 5:
                    r_path = os.path.join(tmp, "ref.gsd")
                    if os.path.exists(r_path):
                        os.unlink(r_path)
                    try:
                        out = subprocess.run([F.MPIEXEC, "-n", str(P), F.REF_DRIVER, scn, r_path], capture_output=True, timeout=120)
                        finished = out.returncode == 0
                    except subprocess.TimeoutExpired:
                        finished = False
                    if not finished:
                        ref_fail += 1
                    elif open(r_path, "rb").read() != want:
                        bad.append((seed, P, "reference"))
                        print("MISMATCH reference vs oracle seed %d P %d" % (seed, P), flush=True)
                ran += 1
            if (seed - first + 1) % 10 == 0:
                print("seeds %d..%d: %d scenarios, %d with a relocated index, mismatches %d, reference did not finish %d"
                      % (first, seed, ran, relocated, len(bad), ref_fail), flush=True)
    print("TOTAL: %d scenarios x {4 placement modes} x {PGSD_CHECK_EOF off, on} at 1/2/3/5 ranks as processes and 8 as threads, "
          "%d with at least one index relocation: product == oracle (== reference wherever it finishes: %d did not) -- mismatches %s"
          % (ran, relocated, ref_fail, bad))
    sys.exit(1 if bad else 0)
