"""Diagnostic: a crowded 256x512 image through the exact engine with MN_X_FORCE_RELOCATE / MN_X_CHECK_SLOTS (the
library under test is chosen by MN_LIB).  usage: python tests/tools/gpu_slot_check.py [golden name]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MN_X_FORCE_RELOCATE"] = "1"
os.environ["MN_X_CHECK_SLOTS"] = "1"
import golden_util as gu
from mergenet_amd import segmenter as seg
from mergenet_amd import labels
g = gu.load(sys.argv[1] if len(sys.argv) > 1 else "cseg_crowd48_256x512_s6400")
H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
ctx = seg.HostContext(H, W, C, len(g["offsets"]))
sdb, omf, bias = g["spec"]["opts"]
o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias,
                        mode=seg.MN_MODE_EXACT, clip_inputs=1, tie_order=seg.MN_TIES_LOWEST_ID)
try:
    mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
    print("status ok; equal to the reference:", labels.masks_equivalent(mask, classes, g["mask"], g["object_class"]))
except Exception as e:
    print("call failed:", e)
