#!/bin/bash
# bench.py for a few ring depths, side stream at lowest / default priority (GPU box); two passes to see the noise
for pass in 1 2; do
for c in 2 3 4; do
  for pr in low default; do
    MN_SIDE_PRIORITY=$pr python bench.py --no-cpu-baseline --no-general-path --no-pipelined --steps 1000 --contexts $c 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pass $pass contexts $c side-priority $pr %8.1f Mpixel/s  %.4f ms/step  sweep %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3))
"
  done
done
done
