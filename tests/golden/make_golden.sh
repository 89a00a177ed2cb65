#!/bin/bash
# Regenerate the golden .gsd fixtures with the REFERENCE ITSELF (oracle/_ref/ref_driver =
# /root/reference/pgsd/pgsd/pgsd.c compiled in place + tests/drivers/scenario_driver.c),
# run under MPICH mpiexec at several rank counts.  Only runs where /root/reference and
# /opt/conda MPICH exist (the build container).  The fixtures are data (bytes the
# reference wrote for closed-form inputs); no reference source is stored here.
set -euo pipefail
cd "$(dirname "$0")/../.."
make -C oracle ref >/dev/null
MPIEXEC=${MPIEXEC:-/opt/conda/bin/mpiexec}
OUT=tests/golden/files
mkdir -p "$OUT"
rm -f "$OUT"/*.gsd "$OUT"/*.log
# the reference's own GSD v1.0 read fixture (data file of its test suite, test_fl.py:613-651);
# vone_append.scn starts from a copy of it (`prefill`)
cp /root/reference/pgsd/pgsd/test/test_gsd_v1.gsd "$OUT/reference_test_gsd_v1.gsd"
declare -A RANKS=(
  [posvelid]="1 2 4 8" [sph_full]="1 2 4 8" [index_expand]="1 2 4 8"
  [zero_rank]="1 2 4 8" [alltypes]="1 3" [names_reloc]="1 2 5" [maxbuf]="1 2 4"
  [reopen]="1 2 4" [midflush]="1 2 3" [benchlike]="1 2 4 8"
  [defaultargs]="1 2 3 4" [readback]="1 2 4" [vone_append]="1 2 3" [idxbuf]="1 2 3" [errors]="1"
)
for scn in tests/golden/scenarios/*.scn; do
  name=$(basename "$scn" .scn)
  for p in ${RANKS[$name]}; do
    rm -f /tmp/golden_$$.gsd
    # errors.scn makes calls that fail on purpose: the driver then exits with 1 after finishing the script
    timeout 120 $MPIEXEC -n "$p" oracle/_ref/ref_driver "$scn" /tmp/golden_$$.gsd > "$OUT/$name.p$p.log" \
      || [ "$name" = errors ]
    mv /tmp/golden_$$.gsd "$OUT/$name.p$p.gsd"
  done
done
( cd "$OUT" && sha256sum *.gsd *.log > SHA256SUMS )
du -sh "$OUT"
