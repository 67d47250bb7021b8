"""Network-like (blurred) 1024x2048 map through the general path for a few settings (GPU box)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from mergenet_amd import segmenter as seg, labels, synth

offs = synth.generate_offsets(40, 10)
s = synth.blurred_v1(1024, 2048, 9, offs, 4242, radius=2, noise=0.05)
ctx = seg.HostContext(1024, 2048, 9, 10)
base = None
for kw in (dict(), dict(finish_limit=3072), dict(finish_limit=2048), dict(finish_limit=1536), dict(finish_limit=6144), dict(band_permille=100), dict(band_permille=20)):
    o = seg.default_options(mode=seg.MN_MODE_ROUNDS, clip_inputs=1, **kw)
    ctx.segment(s.class_probs, s.sameness_probs, offs, o)
    mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
    if base is None:
        base = mask
    print("%-28s K %3d  %.2f ms  rounds %3d  steps %5d  agree with default %d / %d" % (
        kw, len(classes), st["ms_total"], st["rounds"], st["finisher_steps"], labels.agreement(base, mask), mask.size), flush=True)

# the same settings on reference vectors, clusters off so that the rounds really run
import golden_util as gu
for name in ("cseg_synth_1024x2048_cfg2", "cseg_synth_1024x2048_s1003"):
    g = gu.load(name)
    for fl in (4096, 2048):
        o = seg.default_options(mode=seg.MN_MODE_ROUNDS, clip_inputs=1, finish_limit=fl)
        ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
        mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
        print("%s finish_limit %d: equal %d  %.2f ms rounds %d steps %d" % (name, fl, labels.masks_equivalent(mask, classes, g["mask"], g["object_class"]),
              st["ms_total"], st["rounds"], st["finisher_steps"]), flush=True)
