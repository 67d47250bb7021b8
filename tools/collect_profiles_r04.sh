#!/bin/bash
# Round 4's profiles/ evidence in one GPU call (writes under gpurun_out/r04/, one progress line per step):
#   bash tools/collect_profiles_r04.sh
TAG=r04
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
step "kernel stats of the default bench command"
bash tools/prof_default_bench.sh $TAG > $O/default_command.txt 2>&1
cp gpurun_out/${TAG}_default_command_kernel_stats.csv gpurun_out/${TAG}_default_command_bench_line.json $O/ 2>/dev/null
step "PMC: sweep bytes (FETCH_SIZE / WRITE_SIZE) and SQ counters of the fast path"
bash tools/pmc_kernel.sh ${TAG}_pmc > /dev/null 2>&1
cp gpurun_out/${TAG}_pmc_pmc.txt $O/pmc_sq_counters.txt
F=$(ls gpurun_out/${TAG}_pmc_4/*counter_collection.csv gpurun_out/${TAG}_pmc_4/*/*counter_collection.csv 2>/dev/null | head -1)
W=$(ls gpurun_out/${TAG}_pmc_5/*counter_collection.csv gpurun_out/${TAG}_pmc_5/*/*counter_collection.csv 2>/dev/null | head -1)
python tools/pmc_summary.py "$F" "$W" $O/pmc_components_1024x2048.json > /dev/null
step "exact engine: kernel stats (512x1024 + blurred 256x512) and PMC of the loop kernel"
( cd /tmp && export TMPDIR=/tmp && rm -rf $ROOT/gpurun_out/${TAG}_exact && mkdir -p $ROOT/gpurun_out/${TAG}_exact && \
  MN_TOOL_TIE_ORDER=2 MN_TRACE_EXACT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_exact -o run -- python3 $ROOT/tests/tools/gpu_exact.py 600000 512x1024_s1000 blur_256x512 > $ROOT/$O/exact_engine_run.log 2>&1 )
cp $(find gpurun_out/${TAG}_exact -name '*kernel_stats.csv' | head -1) $O/exact_kernel_stats.csv
bash tools/pmc_exact.sh ${TAG}_pmcx > /dev/null 2>&1
cp gpurun_out/${TAG}_pmcx_pmc_exact.txt $O/pmc_exact.txt
step "exact engine, batches in one launch: 512x1024"
python tests/tools/gpu_exact_batch.py 1 64 192 256 > $O/exact_batch_512x1024.log 2>&1
step "exact engine, batches in one launch: 1024x2048"
MN_H=1024 MN_W=2048 MN_NO_REF=1 python tests/tools/gpu_exact_batch.py 1 64 116 > $O/exact_batch_1024x2048.log 2>&1
step "campaign of 84 fresh images, default tie mode, batched"
python tests/tools/gpu_exact_campaign_batch.py 12 > $O/exact_campaign_batch_default_ties.log 2>&1
step "done"
