#!/bin/bash
# one 1024x2048 image through the exact engine under workspace knobs (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LOG="$ROOT/gpurun_out/single_knobs.log"; : > "$LOG"
run() { echo "== $1" >> "$LOG"; env $1 MN_TRACE_EXACT=1 python3 "$ROOT/tests/tools/gpu_exact.py" 3000000 ${2:-cseg_synth_1024x2048_cfg2} 2>&1 | grep -v amdgpu.ids | grep -E "exact engine|OK|BAD|workspace" >> "$LOG"; }
run "MN_X_TABLE_PERMILLE=600"
run "MN_X_TABLE_PERMILLE=310"
run "MN_X_TABLE_PERMILLE=310 MN_X_NO_TIE_TRACKING=1"
run "MN_X_TABLE_PERMILLE=200 MN_X_ARENA_EXTRA=400"
cat "$LOG"
