"""Which (golden, mode) pairs equal the reference on the order-dependent round-2 fixtures (GPU box)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import golden_util as gu
from mergenet_amd import segmenter as seg, labels

NAMES = ["cseg_blur_64x128_r2", "cseg_blur_64x128_r2_s8001", "cseg_blur_256x512_r2",
         "cseg_crowd48_256x512_s6400", "cseg_crowd48_256x512_s6408", "cseg_checker_96x128_b015",
         "cseg_synth_32x64_n60", "cseg_synth_64x128_n60"]
for name in NAMES:
    g = gu.load(name)
    H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
    sdb, omf, bias = g["spec"]["opts"]
    for mode in (seg.MN_MODE_AUTO, seg.MN_MODE_EXACT, seg.MN_MODE_ROUNDS, seg.MN_MODE_COMPONENTS):
        if mode == seg.MN_MODE_EXACT and H * W > 64 * 128:
            continue
        ctx = seg.HostContext(H, W, C, len(g["offsets"]))
        o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias,
                                mode=mode, clip_inputs=1)
        t = time.time()
        mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
        dt = time.time() - t
        ctx.close()
        ok = labels.masks_equivalent(mask, classes, g["mask"], g["object_class"])
        print("%-30s mode %d used %d %-5s inst %d ref %d certified %d steps %d rounds %d %.2f s" % (
            name, mode, st["mode_used"], "EQUAL" if ok else "DIFF", len(classes), len(g["object_class"]),
            st["certified"], st["finisher_steps"], st["rounds"], dt), flush=True)
