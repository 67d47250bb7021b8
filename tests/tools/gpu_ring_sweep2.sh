#!/bin/bash
# bench.py for deeper rings (side stream at its default, lowest priority), GPU box:
#   bash tests/tools/gpu_ring_sweep2.sh
for c in 4 5 6 8 12 16; do
    python bench.py --no-cpu-baseline --no-general-path --no-pipelined --steps 1000 --contexts $c 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('contexts $c %8.1f Mpixel/s  %.4f ms/step  sweep %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3))
"
done
