"""The prefix executor of the CPU model (XM_EXECUTOR=K, with the sorted front XM_FRONT=M) against the sequential model
on fresh random images of several families: same partition, classes, pops, merges; no wrong window.
    python tests/tools/executor_campaign.py [n_per_family]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mergenet_amd import synth
import exact_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
fams = []
for i in range(n):
    fams.append(("adv48", synth.adversarial(48, 48, 4, synth.generate_offsets(6, 5), 1300 + i), synth.generate_offsets(6, 5), (1.0, 0.0)))
    fams.append(("adv64b", synth.adversarial(64, 64, 5, synth.generate_offsets(12, 6), 1400 + i), synth.generate_offsets(12, 6), (0.25, 0.02)))
    fams.append(("blur2", synth.blurred_v1(64, 128, 9, synth.generate_offsets(40, 10), 9100 + i, radius=2, noise=0.05), synth.generate_offsets(40, 10), (1.0, 0.03)))
    fams.append(("blur4", synth.blurred_v1(64, 128, 9, synth.generate_offsets(40, 10), 9300 + i, radius=4, noise=0.05), synth.generate_offsets(40, 10), (1.0, 0.03)))
    fams.append(("synth35", synth.synth_v1(48, 96, 9, synth.generate_offsets(40, 10), 9400 + i, noise=0.35, num_instances=4), synth.generate_offsets(40, 10), (1.0, 0.03)))
    fams.append(("crowd", synth.synth_v1(64, 96, 9, synth.generate_offsets(40, 10), 9500 + i, noise=0.15, num_instances=12), synth.generate_offsets(40, 10), (1.0, 0.03)))
bad = 0
tot = {}
for j, (name, s, offs, (omf, bias)) in enumerate(fams):
    for k in ("XM_EXECUTOR", "XM_FRONT"):
        os.environ.pop(k, None)
    p0, c0, s0 = exact_model.run(s.class_probs, s.sameness_probs, offs, omf, bias, track=False)
    K = (2, 7, 16, 64)[j % 4]
    os.environ["XM_EXECUTOR"] = str(K)
    os.environ["XM_FRONT"] = str((K + 3, 4 * K, 512)[j % 3])
    p1, c1, s1 = exact_model.run(s.class_probs, s.sameness_probs, offs, omf, bias, track=False)
    ok = np.array_equal(p0, p1) and np.array_equal(c0, c1) and (s0["steps"], s0["merges"]) == (s1["steps"], s1["merges"]) and s1["front_errors"] == 0
    t = tot.setdefault(name, dict(n=0, equal=0, pops=0, rounds=0))
    t["n"] += 1; t["equal"] += ok; t["pops"] += s1["steps"]; t["rounds"] += s1["executor_rounds"]
    if not ok:
        bad += 1
        print("DIFFERENT", name, K, s0, s1, flush=True)
for k, v in tot.items():
    print(k, v)
print("different:", bad)
sys.exit(1 if bad else 0)
