#!/bin/bash
# Everything the round's profiles/ directory is made of, in one GPU call:
#   bash tools/collect_profiles.sh r02
set -x
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
mkdir -p gpurun_out/$TAG
python bench.py > gpurun_out/$TAG/bench_line.json 2> gpurun_out/$TAG/bench.err
python bench.py --mode 2 --steps 6 --warmup 1 --no-cpu-baseline --no-pipelined > gpurun_out/$TAG/bench_line_rounds_mode.json 2>> gpurun_out/$TAG/bench.err
MN_PROF_BENCH=1 bash tools/prof_kernels.sh ${TAG}_bench 200 > gpurun_out/$TAG/kernel_stats.txt 2>&1
cp gpurun_out/${TAG}_bench_kernel_stats.csv gpurun_out/$TAG/bench_kernel_stats.csv
MN_PROF_BENCH=1 MN_PROF_ARGS="--mode 2" bash tools/prof_kernels.sh ${TAG}_rounds 6 > gpurun_out/$TAG/kernel_stats_rounds.txt 2>&1
cp gpurun_out/${TAG}_rounds_kernel_stats.csv gpurun_out/$TAG/bench_kernel_stats_rounds_mode.csv
bash tools/pmc_kernel.sh ${TAG}_pmc > /dev/null 2>&1
cp gpurun_out/${TAG}_pmc_pmc.txt gpurun_out/$TAG/pmc_sq_counters.txt
F=$(ls gpurun_out/${TAG}_pmc_4/*counter_collection.csv gpurun_out/${TAG}_pmc_4/*/*counter_collection.csv 2>/dev/null | head -1)
W=$(ls gpurun_out/${TAG}_pmc_5/*counter_collection.csv gpurun_out/${TAG}_pmc_5/*/*counter_collection.csv 2>/dev/null | head -1)
cp "$F" gpurun_out/$TAG/pmc_fetch_components_counter_collection.csv
cp "$W" gpurun_out/$TAG/pmc_write_components_counter_collection.csv
python tools/pmc_summary.py "$F" "$W" gpurun_out/$TAG/pmc_components_1024x2048.json
# configs[4]: the fused four-pixel sweep at W % 4 != 0 (kernel stats + HBM bytes of the sweep)
cd /tmp && export TMPDIR=/tmp
rm -rf $ROOT/gpurun_out/${TAG}_cfg5 && mkdir -p $ROOT/gpurun_out/${TAG}_cfg5
rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_cfg5 -o run -- python3 $ROOT/tools/prof_cfg5.py 10 > $ROOT/gpurun_out/$TAG/cfg5_stdout.log 2>&1
cp $(find $ROOT/gpurun_out/${TAG}_cfg5 -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/$TAG/cfg5_kernel_stats.csv
for cnt in FETCH_SIZE WRITE_SIZE; do
  rm -rf $ROOT/gpurun_out/${TAG}_cfg5_$cnt
  rocprofv3 --pmc $cnt --output-format csv -d $ROOT/gpurun_out/${TAG}_cfg5_$cnt -o run -- python3 $ROOT/tools/prof_cfg5.py 3 > /dev/null 2>&1
done
python3 $ROOT/tools/pmc_summary.py $(find $ROOT/gpurun_out/${TAG}_cfg5_FETCH_SIZE -name '*counter_collection.csv' | head -1) $(find $ROOT/gpurun_out/${TAG}_cfg5_WRITE_SIZE -name '*counter_collection.csv' | head -1) $ROOT/gpurun_out/$TAG/pmc_cfg5_800x1333.json > /dev/null
# the exact engine: kernel stats of one 512x1024 image and one blurred 256x512 map
rm -rf $ROOT/gpurun_out/${TAG}_exact && mkdir -p $ROOT/gpurun_out/${TAG}_exact
MN_TRACE_EXACT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/${TAG}_exact -o run -- python3 $ROOT/tests/tools/gpu_exact.py 600000 512x1024_s1000 blur_256x512 > $ROOT/gpurun_out/$TAG/exact_stdout.log 2>&1
cp $(find $ROOT/gpurun_out/${TAG}_exact -name '*kernel_stats.csv' | head -1) $ROOT/gpurun_out/$TAG/exact_kernel_stats.csv
cd $ROOT
python tests/tools/gpu_exact_campaign.py 8 > gpurun_out/$TAG/exact_campaign.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-exact --no-general-path --no-pipelined > gpurun_out/$TAG/bench_line_steps20_warmup5.json 2>> gpurun_out/$TAG/bench.err
# the memory system's ceiling for the sweep's traffic, and the sweep alone back to back
tools/stream_ceiling > gpurun_out/$TAG/stream_ceiling.log 2>&1
python tests/tools/gpu_sweep_time.py > gpurun_out/$TAG/sweep_alone.log 2>&1
MN_H=800 MN_W=1333 MN_C=81 MN_OA=80,16 python tests/tools/gpu_sweep_time.py > gpurun_out/$TAG/sweep_alone_cfg5.log 2>&1
# the exact engine, batches in one launch
python tests/tools/gpu_exact_batch.py 1 64 192 > gpurun_out/$TAG/exact_batch.log 2>&1
python tests/tools/gpu_exact_ties.py > gpurun_out/$TAG/exact_ties.log 2>&1
bash tools/prof_default_bench.sh $TAG > gpurun_out/$TAG/default_command.txt 2>&1
