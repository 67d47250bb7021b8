#!/bin/bash
# ring of contexts dealt over 1 / 2 / 3 compute streams (GPU box): bench lines of 2000 and of 20 steps
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LOG="$ROOT/gpurun_out/streams_sweep.log"; : > "$LOG"
FLAGS="--no-cpu-baseline --no-exact --no-default-mode --no-pipelined --no-general-path"
for s in 1 2 3 1 2 3; do
  for k in 2000 20; do
    python3 "$ROOT/bench.py" --steps $k --warmup 20 --streams $s $FLAGS 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('streams $s steps $k: %.1f %s, %.4f ms per step, sweep %.1f us, id_match %s' % (d['value'], d['unit'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3, d.get('id_match', {}).get('equal')))" >> "$LOG"
  done
done
cat "$LOG"
