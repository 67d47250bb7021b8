for pass in 1 2 3; do
for lib in old new; do
  if [ $lib = old ]; then export MN_LIB=$PWD/mergenet_amd/libmergenet_hip_old.so; else unset MN_LIB; fi
  python bench.py --no-cpu-baseline --no-general-path --no-pipelined --steps 1500 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pass $pass lib $lib %8.1f Mpixel/s  %.4f ms/step  sweep %.2f us' % (d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'] * 1e3))
"
done
done
