python -m pytest tests/test_gpu_parity.py tests/test_gpu_phase_a.py -q -m gpu -x 2>&1 | tail -2
for f in 0 128 0 128; do
    MN_BENCH_FLAGS=$f python bench.py --no-cpu-baseline --no-general-path --no-pipelined 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('extra flags $f %8.1f Mpixel/s  %.4f ms/step  sweep by events %.1f us frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3, d['roofline']['frac']))
"
done
