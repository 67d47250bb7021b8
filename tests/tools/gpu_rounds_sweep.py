"""ROUNDS mode at 1024x2048: time and equality with the reference goldens for a few settings (GPU box)."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import golden_util as gu
from mergenet_amd import segmenter as seg, labels

names = ["cseg_synth_1024x2048_cfg2", "cseg_synth_1024x2048_s1001", "cseg_synth_1024x2048_s1002", "cseg_synth_1024x2048_s1003"]
gs = [gu.load(n) for n in names]
ctx = seg.HostContext(1024, 2048, 9, 10)
for fl in (4096, 1024, 256):
    for sub in (32, 8, 4):
        eq, ms, steps, rounds = 0, 0.0, 0, 0
        for g in gs:
            o = seg.default_options(mode=seg.MN_MODE_ROUNDS, finish_limit=fl, subrounds=sub, clip_inputs=1)
            mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
            eq += labels.masks_equivalent(mask, classes, g["mask"], g["object_class"])
            ms += st["ms_total"]; steps += st["finisher_steps"]; rounds += st["rounds"]
        print("finish_limit %5d subrounds %2d: %d/%d equal, %.1f ms per image, %d steps, %d rounds" % (fl, sub, eq, len(gs), ms / len(gs), steps // len(gs), rounds // len(gs)), flush=True)
