for r in "" "--replay" "" "--replay"; do
    python bench.py --no-cpu-baseline --no-general-path --no-pipelined $r 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('replay [$r] %8.1f Mpixel/s  %.4f ms/step  sweep %.1f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3))
"
done
