#!/bin/bash
# PMC counters of the default path's kernels (separate passes, no tracing), GPU box:
#   bash tools/pmc_kernel.sh <tag>      -> gpurun_out/<tag>_pmc.txt
TAG=${1:-pmc}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rm -rf $ROOT/gpurun_out/${TAG}_$i
  rocprofv3 --pmc $set --output-format csv -d $ROOT/gpurun_out/${TAG}_$i -o run -- python3 $ROOT/tools/prof_components.py 3 > $ROOT/gpurun_out/${TAG}_$i.log 2>&1
done
python3 - $ROOT/gpurun_out $TAG <<'PY' | tee $ROOT/gpurun_out/${TAG}_pmc.txt
import csv, glob, sys, collections
root, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("%s/%s_*/**/*counter_collection.csv" % (root, tag), recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if name.startswith("mn_"):
            acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(k, " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
