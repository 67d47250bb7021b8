#!/bin/bash
# Issue/stall breakdown of the components-mode kernels (GPU box): three counter passes.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/sq1 gpurun_out/sq2 gpurun_out/sq3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d gpurun_out/sq1 -- python tools/prof_components.py 3 > gpurun_out/sq1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d gpurun_out/sq2 -- python tools/prof_components.py 3 > gpurun_out/sq2.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAVES --output-format csv -d gpurun_out/sq3 -- python tools/prof_components.py 3 > gpurun_out/sq3.log 2>&1
python - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for d in ("sq1", "sq2", "sq3"):
    for f in glob.glob("gpurun_out/%s/*/*_counter_collection.csv" % d):
        for row in csv.DictReader(open(f)):
            n = row["Kernel_Name"].split("(")[0]
            if n.startswith("mn_"):
                acc[n][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU",
         "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_LDS", "SQ_INSTS_SALU"]
print("kernel," + ",".join(names))
for k in sorted(acc):
    print(k + "," + ",".join("%.3g" % (sum(acc[k][n]) / max(1, len(acc[k][n]))) for n in names))
PY
