#!/bin/bash
# A/B of variant builds of the library inside the bench loop (GPU box): per variant the bench line of 2000 steps
# and the rocprofv3 per-kernel averages of 400 steps.   tools/variant_ab.sh build_diag/lib_a.so build_diag/lib_b.so ...
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
LOG="$ROOT/gpurun_out/variant_ab.log"
: > "$LOG"
FLAGS="--no-cpu-baseline --no-exact --no-default-mode --no-pipelined --no-general-path"
for lib in "$@"; do
  for rep in 1 2; do
    MN_LIB="$ROOT/$lib" python3 "$ROOT/bench.py" --steps 2000 --warmup 20 $FLAGS 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$lib rep $rep: %.1f %s, %.4f ms per step, id_match %s' % (d['value'], d['unit'], d['ms_per_step'], d.get('id_match', {}).get('equal')))" >> "$LOG"
  done
  OUT=/tmp/vab_$$; mkdir -p $OUT
  (cd /tmp && export TMPDIR=/tmp && MN_LIB="$ROOT/$lib" rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 "$ROOT/bench.py" --steps 400 --warmup 20 $FLAGS > $OUT/stdout.log 2>&1)
  f=$(find $OUT -name '*kernel_stats.csv' | head -1)
  python3 - "$f" "$lib" >> "$LOG" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0]
    if 'mn_' in n and int(r['Calls']) > 100:
        print("  %s  %-36s calls %5s avg %8.1f us" % (sys.argv[2], n[:36], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -rf $OUT
done
cat "$LOG"
