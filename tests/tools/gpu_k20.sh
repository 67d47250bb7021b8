# The driver's command line (--steps 20 --warmup 5) against the ring depth, three runs each.
for c in 3 4 6 8 12; do
  for rep in 1 2 3; do
    python bench.py --no-cpu-baseline --no-general-path --no-pipelined --no-exact --steps 20 --warmup 5 --contexts $c 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('contexts %-3s %8.1f Mpixel/s  %.4f ms/step  sweep %.2f us' % ('$c', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3))
"
  done
done
