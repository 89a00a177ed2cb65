#!/usr/bin/env python3
"""One-off wide campaign of the host-layer fuzzer (tests/test_fuzz_parity.py's script generator): every seed is
replayed through the compiled reference (oracle/_ref, under mpiexec), the oracle and the product -- unbatched and with
the frame exchange batched -- and files and state traces are compared.  Build container only (needs oracle/_ref).

    python tools/fuzz_campaign.py [first_seed=1000] [n_seeds=300]
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

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
strip = lambda lines: [re.sub(r"line=\d+ ", "", ln) for ln in lines]
bad, ran, ref_fail = [], 0, 0
with tempfile.TemporaryDirectory() as tmp:
    for seed in range(first, first + count):
        for P in (1, 2, 3, 4):
            scn = os.path.join(tmp, "f.scn")
            open(scn, "w").write(F.make_script(seed, P))
            paths = {k: os.path.join(tmp, k + ".gsd") for k in ("ref", "oracle", "product", "batched", "deferred", "declared")}
            for p in paths.values():
                if os.path.exists(p):
                    os.unlink(p)
            o_log = S.run_oracle(scn, paths["oracle"], P)
            if [ln for ln in o_log if ln.startswith("rc ")]:
                continue                        # scripts with failing calls are the errors scenario's business
            p_log = product.run_driver(scn, paths["product"], P)
            b_log = product.run_driver(product.batched_script(scn, os.path.join(tmp, "b.scn")), paths["batched"], P)
            d_log = product.run_driver(product.batched_script(scn, os.path.join(tmp, "d.scn"), 2), paths["deferred"], P)
            t_log = product.run_driver(product.batched_script(scn, os.path.join(tmp, "t.scn"), 3), paths["declared"], P)
            try:
                out = subprocess.run([F.MPIEXEC, "-n", str(P), F.REF_DRIVER, scn, paths["ref"]], capture_output=True, timeout=60)
                finished = out.returncode == 0
            except subprocess.TimeoutExpired:
                finished = False
            if not finished:
                # the reference dead-locks or aborts on some legal call sequences at P > 1 (a read whose rank-0 check
                # returns before the collective): oracle and product must still agree
                ref_fail += 1
                same = (open(paths["oracle"], "rb").read() == open(paths["product"], "rb").read()
                        == open(paths["batched"], "rb").read() == open(paths["deferred"], "rb").read()
                        == open(paths["declared"], "rb").read()
                        and p_log == o_log and strip(b_log) == strip(o_log) and strip(d_log) == strip(o_log)
                        and strip(t_log) == strip(o_log))
                if not same:
                    bad.append((seed, P, "oracle/product"))
                    print("MISMATCH (oracle vs product) seed %d P %d" % (seed, P), flush=True)
                continue
            r_log = [ln for ln in out.stdout.decode().splitlines() if ln.strip()]
            data = {k: open(p, "rb").read() for k, p in paths.items()}
            ok = (data["oracle"] == data["ref"] and data["product"] == data["ref"] and data["batched"] == data["ref"]
                  and data["deferred"] == data["ref"] and data["declared"] == data["ref"]
                  and o_log == r_log and p_log == r_log and strip(b_log) == strip(r_log) and strip(d_log) == strip(r_log)
                  and strip(t_log) == strip(r_log))
            ran += 1
            if not ok:
                bad.append((seed, P))
                print("MISMATCH seed %d P %d" % (seed, P), flush=True)
        if (seed - first) % 25 == 24:
            print("seeds %d..%d: %d three-way cases identical, %d mismatches, %d the reference did not finish"
                  % (first, seed, ran - len(bad), len(bad), ref_fail), flush=True)
print("TOTAL: %d cases (seeds %d..%d x 1-4 ranks): reference == oracle == product == product(batched) == product(batched, deferred rows) == product(declared partitions) in %d, mismatches %s, "
      "reference did not finish %d (oracle == product there)" % (ran, first, first + count - 1, ran - len(bad), bad, ref_fail))
