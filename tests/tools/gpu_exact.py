"""GPU probe: MN_MODE_EXACT (the exact engine) against reference vectors; prints time and steps.

usage: python tests/tools/gpu_exact.py [max_pixels] [name_substring ...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import golden_util as gu
from mergenet_amd import segmenter as seg
from oracle import checker as ck

max_px = int(sys.argv[1]) if len(sys.argv) > 1 else 256 * 512
subs = sys.argv[2:]
bad = []
for name in gu.names("cseg_"):
    if subs and not any(s in name for s in subs):
        continue
    g = gu.load(name)
    H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
    if H * W > max_px:
        continue
    ctx = seg.HostContext(H, W, C, len(g["offsets"]))
    o = seg.default_options(same_different_bias=g["spec"]["opts"][0], object_merge_factor=g["spec"]["opts"][1],
                            merge_logprob_bias=g["spec"]["opts"][2], mode=seg.MN_MODE_EXACT, clip_inputs=1,
                            tie_order=int(os.environ.get("MN_TOOL_TIE_ORDER", "0")))
    t = time.time()
    mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
    dt = time.time() - t
    ok = ck.masks_equivalent(mask, classes, g["mask"], g["object_class"])
    print("%-34s %4dx%-4d %s  %.2f s  steps %d merges %d objects %d inst %d ms_merge %.1f proof %d cert %d tied %d conflicts %d" % (
        name, H, W, "OK " if ok else "BAD", dt, st["finisher_steps"], st["merges"], st["num_objects"],
        st["num_instances"], st["ms_merge"], st["proof"], st["certified"], st["tied_steps"], st.get("tied_conflicts", -1)), flush=True)
    if not ok:
        bad.append(name)
    print("    workspace %.3f GB (%.0f B per pixel)" % (ctx.workspace_bytes() / 1e9, ctx.workspace_bytes() / (H * W)))
    ctx.close()
print("mismatching:", bad)
sys.exit(1 if bad else 0)
