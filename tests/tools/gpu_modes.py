"""ROUNDS vs COMPONENTS mode on golden fixtures: equality with the reference and timing (GPU box)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import golden_util as gu
from mergenet_amd import segmenter as seg, labels
for name in sys.argv[1:]:
    g = gu.load(name)
    H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
    ctx = seg.HostContext(H, W, C, len(g["offsets"]))
    for mode in (seg.MN_MODE_COMPONENTS, seg.MN_MODE_ROUNDS):
        sdb, omf, bias = g["spec"]["opts"]
        o = seg.default_options(mode=mode, same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias, clip_inputs=1, finish_limit=int(__import__("os").environ.get("MN_FINISH", "0")))
        for rep in range(2):
            mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
        eq = labels.masks_equivalent(mask, classes, g["mask"], g["object_class"])
        print(name, "mode", mode, "used", st["mode_used"], "equal", eq, "inst", st["num_instances"], "ref", len(g["object_class"]),
              "rounds", st["rounds"], "steps", st["finisher_steps"], "cert", st["certified"], "ms %.2f" % st["ms_total"], "merge %.2f" % st["ms_merge"], flush=True)
    ctx.close()
