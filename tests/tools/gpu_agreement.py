"""GPU probe: how far from the reference is the exact engine's own tie rule on a tie-decided vector?
    python tests/tools/gpu_agreement.py NAME [...]   -> pixels in agreement (best label matching), instance counts"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import golden_util as gu
from mergenet_amd import labels, segmenter as seg

for name in sys.argv[1:]:
    g = gu.load(name)
    H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
    ctx = seg.HostContext(H, W, C, len(g["offsets"]))
    o = seg.default_options(same_different_bias=g["spec"]["opts"][0], object_merge_factor=g["spec"]["opts"][1],
                            merge_logprob_bias=g["spec"]["opts"][2])
    mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
    ctx.close()
    agree = labels.agreement(mask, g["mask"])
    print("%s: mode %d proof %d tied %d conflicts %d; %d of %d pixels agree (%.4f %%), instances %d (reference %d), equal %s" % (
        name, st["mode_used"], st["proof"], st["tied_steps"], st["tied_conflicts"], agree, mask.size, 100.0 * agree / mask.size,
        len(classes), len(g["object_class"]), labels.masks_equivalent(mask, classes, g["mask"], g["object_class"])), flush=True)
