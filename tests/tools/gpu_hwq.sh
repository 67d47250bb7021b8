# Bench loop against the number of hardware queues the HIP runtime spreads its streams over
# (GPU_MAX_HW_QUEUES, default 4): the ring's side streams share them.   bash tests/tools/gpu_hwq.sh 4 8 16
for q in "$@"; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --no-general-path --no-pipelined --no-exact --steps ${MN_AB_STEPS:-1500} $MN_AB_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hw queues %-3s %8.1f Mpixel/s  %.4f ms/step  sweep %.2f us  equal %s' % ('$q', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3, d.get('id_match', {}).get('equal')))
"
done
