import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import golden_util as gu
from mergenet_amd import segmenter as seg, labels
for name in ("cseg_synth_800x1333_cfg5", "cseg_synth_400x667_c81"):
    g = gu.load(name)
    H, W = g["mask"].shape
    ctx = seg.HostContext(H, W, g["class_probs"].shape[0], g["sameness_probs"].shape[0])
    for mode in (seg.MN_MODE_AUTO, seg.MN_MODE_ROUNDS):
        o = seg.default_options(merge_logprob_bias=g["spec"]["opts"][2], mode=mode, clip_inputs=1, require_proof=-1)
        ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
        mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
        print(name, "mode", mode, "used", st["mode_used"], "equal", labels.masks_equivalent(mask, classes, g["mask"], g["object_class"]),
              "%.3f ms" % st["ms_total"], "sweep %.3f" % st["ms_cc_edges"], "label %.3f sums %.3f cross %.3f" % (st["ms_cc_label"], st["ms_cc_sums"], st["ms_cc_cross"]), "rounds", st["rounds"], flush=True)
