# What the per-dispatch events on the sweep cost the loop: with and without them, two passes.
for pass in 1 2; do
for a in "" "--no-kernel-events"; do
  python bench.py --no-cpu-baseline --no-general-path --no-pipelined --no-exact --steps 2000 $a 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pass $pass %-20s %8.1f Mpixel/s  %.4f ms/step  sweep %.2f us' % ('$a', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3))
"
done
done
