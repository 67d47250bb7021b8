# Bench loop against runtime knobs: side-stream priority, ring depth.   bash tests/tools/gpu_loop_knobs.sh
run() {
  env "$@" python bench.py --no-cpu-baseline --no-general-path --no-pipelined --no-exact --steps 1500 $EXTRA 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-44s %8.1f Mpixel/s  %.4f ms/step  sweep %.2f us' % ('$* $EXTRA', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3))
"
}
EXTRA="" run MN_NOOP=1
EXTRA="" run MN_SIDE_PRIORITY=default
EXTRA="--contexts 16" run MN_NOOP=1
EXTRA="--contexts 16" run MN_SIDE_PRIORITY=default
EXTRA="--contexts 4" run MN_NOOP=1
EXTRA="--contexts 4" run MN_SIDE_PRIORITY=default
