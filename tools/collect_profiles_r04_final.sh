#!/bin/bash
# The part of tools/collect_profiles_r04.sh that the last code changes of the round touch (class sums, tie stamps):
# bench lines, kernel stats of the bench loop, exact-engine run log + kernel stats.  Writes under gpurun_out/r04f/.
TAG=r04f
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/$TAG
O=gpurun_out/$TAG
step() { echo "[collect $(date +%H:%M:%S)] $*"; }
step "bench, driver's command line"
python bench.py --steps 20 --warmup 5 > $O/bench_line_driver_cmd.json 2> $O/bench.err
step "bench, default command (2000 steps)"
python bench.py > $O/bench_line.json 2>> $O/bench.err
step "kernel stats of the bench loop (fast path), 200 steps"
MN_PROF_BENCH=1 MN_PROF_ARGS="--no-default-mode" bash tools/prof_kernels.sh ${TAG}_bench 200 > $O/bench_kernel_stats_per_image.txt 2>&1
cp gpurun_out/${TAG}_bench_kernel_stats.csv $O/bench_kernel_stats.csv
step "exact engine: kernel stats (512x1024 + blurred 256x512)"
( cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/gpurun_out/${TAG}_exact && mkdir -p $ROOT/gpurun_out/${TAG}_exact && \
  MN_TOOL_TIE_ORDER=2 MN_TRACE_EXACT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_exact -o run -- python3 $ROOT/tests/tools/gpu_exact.py 600000 512x1024_s1000 blur_256x512 > $ROOT/$O/exact_engine_run.log 2>&1 )
cp $(find gpurun_out/${TAG}_exact -name '*kernel_stats.csv' | head -1) $O/exact_kernel_stats.csv
step "done"
