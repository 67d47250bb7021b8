#!/bin/bash
# Per-kernel durations of the default path (rocprofv3 --kernel-trace --stats), GPU box:
#   bash tools/prof_kernels.sh <tag> [images]      -> gpurun_out/<tag>_kernel_stats.csv
set -e
TAG=${1:-prof}; NIMG=${2:-20}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
if [ "${MN_PROF_BENCH:-1}" = "1" ]; then
  # the bench's own timed loop (4 images in rotation, launch of step i+1 before the read-back of step i)
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 "$ROOT/bench.py" --steps "$NIMG" --warmup 4 --no-cpu-baseline --no-general-path --no-pipelined --no-exact --spin-seconds 0 ${MN_PROF_ARGS:-} > "$OUT/stdout.log" 2>&1
  NIMG=$((NIMG + 4 + 1 + 4))     # + warm-up, the initialisation call and the four id-match images
else
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 "$ROOT/tools/prof_components.py" "$NIMG" > "$OUT/stdout.log" 2>&1
fi
f=$(find "$OUT" -name '*kernel_stats.csv' | head -1)
cp "$f" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"
python3 - "$ROOT/gpurun_out/${TAG}_kernel_stats.csv" "$NIMG" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
tot = 0.0
for r in rows:
    name = r["Name"].split("(")[0]
    if not (name.startswith("mn_") or name.startswith("void mn_")): continue
    t = float(r["TotalDurationNs"]) / 1e3 / n
    tot += t
    print("%-60s calls/img %5.1f  avg %8.2f us  per image %8.2f us" % (name[:60], int(r["Calls"]) / n, float(r["AverageNs"]) / 1e3, t))
print("sum of mn_ kernels per image: %.1f us" % tot)
PY
tail -2 "$OUT/stdout.log"
