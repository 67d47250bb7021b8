#!/bin/bash
# rocprofv3 --kernel-trace --stats of the DEFAULT bench command (no flags but --no-cpu-baseline, which
# only drops the CPU leg); the trace itself (hundreds of thousands of dispatches) stays on the box,
# the per-kernel summary goes to gpurun_out/<tag>_default_command_kernel_stats.csv
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=/tmp/prof_default_$$
mkdir -p "$OUT" "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 "$ROOT/bench.py" --no-cpu-baseline --no-exact > "$OUT/stdout.log" 2>&1
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
cp "$f" "$ROOT/gpurun_out/${TAG}_default_command_kernel_stats.csv"
grep '^{' "$OUT/stdout.log" | tail -1 > "$ROOT/gpurun_out/${TAG}_default_command_bench_line.json"
head -8 "$ROOT/gpurun_out/${TAG}_default_command_kernel_stats.csv" | cut -c1-160
rm -rf "$OUT"
