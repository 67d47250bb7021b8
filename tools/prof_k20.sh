#!/bin/bash
# Kernel trace of the driver's command line (--steps 20 --warmup 5): where the 3 ms window goes.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/k20
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT" -o run -- python3 "$ROOT/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-general-path --no-pipelined --no-exact > "$OUT/stdout.log" 2>&1
tail -1 "$OUT/stdout.log" | cut -c1-200
