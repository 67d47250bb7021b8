#!/bin/bash
# bench.py for a few ring depths, with and without replay (GPU box); two passes to see the noise
for pass in 1 2; do
for c in 2 3 4 6; do
  for r in "" "--replay"; do
    python bench.py --no-cpu-baseline --no-general-path --no-pipelined --steps 1000 --contexts $c $r 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pass $pass contexts $c %-9s %8.1f Mpixel/s  %.4f ms/step  sweep %.1f us' % ('$r', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3))
"
  done
done
done
