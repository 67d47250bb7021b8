# A/B of variant builds of the library (mergenet_amd/libmergenet_hip_<tag>.so, built with other -D flags):
#   bash tests/tools/gpu_ab_variants.sh d1 d2 d3      (two passes over the tags; bench figures per tag)
for pass in 1 2; do
for tag in "$@"; do
  export MN_LIB=$PWD/mergenet_amd/libmergenet_hip_$tag.so
  python bench.py --no-cpu-baseline --no-general-path --no-pipelined --no-exact --steps ${MN_AB_STEPS:-1500} $MN_AB_ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pass $pass lib %-6s %8.1f Mpixel/s  %.4f ms/step  sweep %.2f us  equal %s' % ('$tag', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3, d.get('id_match', {}).get('equal')))
"
done
done
