#!/bin/bash
# The driver's command line (--steps 20 --warmup 5) with different ring depths, three runs each, on one box.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
for rep in 1 2 3; do
  for c in 4 6 8 12; do
    python bench.py --steps 20 --warmup 5 --contexts $c --no-default-mode --no-exact --no-cpu-baseline --no-general-path --no-pipelined 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('contexts $c rep $rep: %.1f Mpixel/s %.4f ms per step' % (d['value'], d['ms_per_step']))"
  done
done
