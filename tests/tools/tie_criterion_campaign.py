"""Soundness campaign of the tie-conflict criterion on the CPU: wherever the model (lowest record id among equals)
reports tied pops but NO tie conflict, its partition must equal the oracle's (the reference's heap order).
usage: python tests/tools/tie_criterion_campaign.py [n_per_family]"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mergenet_amd import synth
from oracle import checker as ck
import exact_model

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
fams = []
for i in range(n):
    fams.append(("adv48", lambda i=i: (synth.adversarial(48, 48, 4, synth.generate_offsets(6, 5), 300 + i), synth.generate_offsets(6, 5), 4, (1.0, 0.0))))
    fams.append(("adv64b", lambda i=i: (synth.adversarial(64, 64, 5, synth.generate_offsets(12, 6), 400 + i), synth.generate_offsets(12, 6), 5, (0.25, 0.02))))
    fams.append(("blur2", lambda i=i: (synth.blurred_v1(64, 128, 9, synth.generate_offsets(40, 10), 8100 + i, radius=2, noise=0.05), synth.generate_offsets(40, 10), 9, (1.0, 0.03))))
    fams.append(("blur1n", lambda i=i: (synth.blurred_v1(64, 128, 9, synth.generate_offsets(40, 10), 8200 + i, radius=1, noise=0.10), synth.generate_offsets(40, 10), 9, (1.0, 0.03))))
    fams.append(("blur4", lambda i=i: (synth.blurred_v1(64, 128, 9, synth.generate_offsets(40, 10), 8300 + i, radius=4, noise=0.05), synth.generate_offsets(40, 10), 9, (1.0, 0.03))))
    fams.append(("synth35", lambda i=i: (synth.synth_v1(48, 96, 9, synth.generate_offsets(40, 10), 8400 + i, noise=0.35, num_instances=4), synth.generate_offsets(40, 10), 9, (1.0, 0.03))))
tot = {}
for name, mk in fams:
    s, offs, C, (omf, bias) = mk()
    ref = ck.run_csegment(s.class_probs, s.sameness_probs, C, offs, 0.0, omf, bias)
    part, ocls, st = exact_model.run(s.class_probs, s.sameness_probs, offs, omf, bias)
    same = ck.same_partition(part, ref.partition)
    t = tot.setdefault(name, dict(n=0, equal=0, tied=0, tied_noconf=0, noconf_equal=0, unsound=0))
    t["n"] += 1; t["equal"] += same
    if st["tied_steps"] > 0:
        t["tied"] += 1
        if st["tied_conflicts"] == 0:
            t["tied_noconf"] += 1; t["noconf_equal"] += same
            if not same: t["unsound"] += 1; print("UNSOUND", name, st)
    if st["tied_steps"] == 0 and not same: print("BROKEN (no ties, differs)", name, st)
for k, v in tot.items():
    print(k, v)
