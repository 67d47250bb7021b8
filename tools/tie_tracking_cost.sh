#!/bin/bash
# what the tie tracking costs while it runs (GPU box): the same images with and without it
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LOG="$ROOT/gpurun_out/tie_tracking_cost.log"; : > "$LOG"
for rep in 1 2; do
for e in "MN_X_TRACK=default" "MN_X_NO_TIE_TRACKING=1"; do
  echo "== $e (rep $rep)" >> "$LOG"
  env $e MN_TOOL_TIE_ORDER=2 python3 "$ROOT/tests/tools/gpu_exact.py" 600000 blur_256x512 blur_64x128_r2_s8001 512x1024_s1000 blur_512x1024 2>&1 | grep -E "OK|BAD" >> "$LOG"
done
done
cat "$LOG"
