#!/bin/bash
# Issue / stall counters of the exact engine's loop kernel (mn_x_run) on one blurred 256x512 image (GPU box):
#   bash tools/pmc_exact.sh <tag>      -> gpurun_out/<tag>_pmc_exact.txt      (separate --pmc passes, no tracing)
TAG=${1:-pmcx}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export MN_TOOL_TIE_ORDER=2
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_BUSY_CYCLES" "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_INSTS_FLAT"; do
  i=$((i+1))
  rm -rf $ROOT/gpurun_out/${TAG}_$i
  rocprofv3 --pmc $set --output-format csv -d $ROOT/gpurun_out/${TAG}_$i -o run -- python3 $ROOT/tests/tools/gpu_exact.py 200000 blur_256x512 > $ROOT/gpurun_out/${TAG}_$i.log 2>&1
done
python3 - $ROOT/gpurun_out $TAG <<'PY' | tee $ROOT/gpurun_out/${TAG}_pmc_exact.txt
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/%s_*/**/*counter_collection.csv" % (root, tag), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if name.startswith("mn_x_run"):
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    for c, v in sorted(acc[k].items()):
        print("%s %s = %.5g (%d launches)" % (k, c, sum(v), len(v)))
PY
