"""GPU probe: the engine's tie-conflict verdict (stats.tied_conflicts > 0) against the CPU model's, image by image, on
families where the verdict varies (blurred 64x128 maps: about half of them have tied pops that provably commute).
    python tests/tools/gpu_tie_verdicts.py [n_per_family]
TEST TOOL (uses tests/tools/exact_model.cpp)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np

import exact_model
from mergenet_amd import labels, segmenter as seg, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
fams = []
for i in range(n):
    fams.append(("blur2", synth.blurred_v1(64, 128, 9, synth.generate_offsets(40, 10), 8100 + i, radius=2, noise=0.05), synth.generate_offsets(40, 10), 9, (1.0, 0.03)))
    fams.append(("adv64b", synth.adversarial(64, 64, 5, synth.generate_offsets(12, 6), 400 + i), synth.generate_offsets(12, 6), 5, (0.25, 0.02)))
    fams.append(("blur2w", synth.blurred_v1(96, 160, 9, synth.generate_offsets(40, 10), 8500 + i, radius=2, noise=0.05), synth.generate_offsets(40, 10), 9, (1.0, 0.03)))
bad = 0
tot = {}
for name, s, offs, C, (omf, bias) in fams:
    H, W = s.class_probs.shape[1:]
    part, ocls, m = exact_model.run(s.class_probs, s.sameness_probs, offs, omf, bias)
    ctx = seg.HostContext(H, W, C, len(offs))
    o = seg.default_options(object_merge_factor=omf, merge_logprob_bias=bias, mode=seg.MN_MODE_EXACT, clip_inputs=1,
                            tie_order=seg.MN_TIES_LOWEST_ID)
    mask, classes, gpart, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
    ctx.close()
    same = labels.same_partition(np.asarray(gpart).reshape(-1), part.reshape(-1))
    ok = same and (st["finisher_steps"], st["merges"], st["tied_steps"]) == (m["steps"], m["merges"], m["tied_steps"]) \
        and (st["tied_conflicts"] > 0) == (m["tied_conflicts"] > 0)
    t = tot.setdefault(name, dict(n=0, agree=0, tied=0, no_conflict=0))
    t["n"] += 1; t["agree"] += ok; t["tied"] += m["tied_steps"] > 0; t["no_conflict"] += (m["tied_steps"] > 0 and m["tied_conflicts"] == 0)
    if not ok:
        bad += 1
        print("DISAGREE", name, st["finisher_steps"], m["steps"], st["tied_steps"], m["tied_steps"], st["tied_conflicts"], m["tied_conflicts"], same, flush=True)
for k, v in tot.items():
    print(k, v)
print("disagreements:", bad)
sys.exit(1 if bad else 0)
