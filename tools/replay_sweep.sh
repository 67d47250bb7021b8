#!/bin/bash
# hipGraph replay of the launches behind the sweep (bench --replay) against plain launches, 20 and 2000 steps (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LOG="$ROOT/gpurun_out/replay_sweep.log"; : > "$LOG"
FLAGS="--no-cpu-baseline --no-exact --no-default-mode --no-pipelined --no-general-path"
for rep in 1 2 3; do
for r in "" "--replay"; do
  for k in 20 2000; do
    python3 "$ROOT/bench.py" --steps $k --warmup 5 $r $FLAGS 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('rep $rep replay [$r] steps $k: %.1f %s, %.4f ms per step, id_match %s' % (d['value'], d['unit'], d['ms_per_step'], d.get('id_match', {}).get('equal')))" >> "$LOG"
  done
done
done
cat "$LOG"
