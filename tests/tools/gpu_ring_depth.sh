# Ring depth (contexts in rotation) against the step time, at 2000 steps and at the driver's 20.
for c in 8 16 24 32; do
  for k in 2000 20; do
    python bench.py --no-cpu-baseline --no-general-path --no-pipelined --no-exact --steps $k --warmup 5 --contexts $c 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('contexts %-3s steps %-5s %8.1f Mpixel/s  %.4f ms/step  sweep %.2f us' % ('$c', '$k', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3))
"
  done
done
