"""The parallel rounds themselves (cores on, clusters off) against every reference vector, for a few
core radii (GPU box; argv: radii, -1 = every offset short): equality with the reference, time, rounds."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import golden_util as gu
from mergenet_amd import segmenter as seg, labels

bands = [int(b) for b in sys.argv[1:]] or [-1, 6]
names = [n for n in gu.names("cseg_") + gu.names("py_")]
tot = {b: [0, 0, 0.0] for b in bands}
for n in names:
    g = gu.load(n)
    H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
    sdb, omf, bias = g["spec"]["opts"]
    variant = seg.MN_VARIANT_PYSEGMENTER if n.startswith("py_") else seg.MN_VARIANT_CSEGMENT
    ctx = seg.HostContext(H, W, C, len(g["offsets"]))
    line = "%-36s" % n
    for b in bands:
        o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias,
                                mode=seg.MN_MODE_ROUNDS, clip_inputs=1, core_radius=b, variant=variant)
        try:
            mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
            eq = labels.masks_equivalent(mask, classes, g["mask"], g["object_class"])
            ag = labels.agreement(mask, g["mask"]) / mask.size
            line += " | core %3d: %s %.4f %6.2f ms r%3d" % (b, "EQ  " if eq else "DIFF", ag, st["ms_total"], st["rounds"])
            tot[b][0] += int(eq); tot[b][1] += 1; tot[b][2] += st["ms_total"]
        except Exception as e:
            line += " | core %3d: error %s" % (b, str(e)[:30])
    ctx.close()
    print(line, flush=True)
for b in bands:
    print("core radius %d: %d / %d equal, %.1f ms in all" % (b, tot[b][0], tot[b][1], tot[b][2]))
