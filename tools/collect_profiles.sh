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
