for c in 8 4 3 8 4; do
    python3 bench.py --gpus 1 --steps 20 --warmup 5 --contexts $c --no-cpu-baseline --no-general-path --no-pipelined 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('K=20 W=5 contexts $c %8.1f Mpixel/s  %.4f ms/step  sweep %.1f us frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3, d['roofline']['frac']))
"
done
python3 bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python tools/show_bench.py /dev/stdin | head -3
