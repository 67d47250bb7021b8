# Sweep duration (HIP events, one context: the sweep runs alone) per variant build; results are NOT checked
# (diagnostic builds skip work on purpose).   bash tests/tools/gpu_ab_sweep_only.sh x0 x1 ...
for pass in 1 2; do
for tag in "$@"; do
  export MN_LIB=$PWD/mergenet_amd/libmergenet_hip_$tag.so
  python bench.py --no-cpu-baseline --no-general-path --no-pipelined --no-exact --contexts 1 --steps 400 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pass $pass lib %-6s sweep %.2f us  (%.4f ms/step, equal %s)' % ('$tag', d['roofline']['avg_launch_ms'] * 1e3, d['ms_per_step'], d.get('id_match', {}).get('equal')))
"
done
done
