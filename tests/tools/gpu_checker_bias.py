"""Checkerboard of small cells with a bias large enough to merge cells in phase 2 (GPU box)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from mergenet_amd import synth, segmenter as seg
from oracle import checker as ck
from test_gpu_parity import _checkerboard

offs = synth.generate_offsets(6, 4)
for cellpx, bias, seed in [(4, 0.3, 3), (4, 0.15, 4), (8, 0.3, 5), (6, 0.2, 6)]:
    cp, sp = _checkerboard(96, 128, 3, offs, cellpx, seed)
    ref = ck.run_csegment(cp, sp, 3, offs, 0.0, 1.0, bias)
    for mode in (seg.MN_MODE_COMPONENTS, seg.MN_MODE_ROUNDS, seg.MN_MODE_EXACT):
        ctx = seg.HostContext(96, 128, 3, len(offs))
        o = seg.default_options(merge_logprob_bias=bias, mode=mode, clip_inputs=1)
        mask, classes, part, st = ctx.segment(cp, sp, offs, o)
        ctx.close()
        print("cell %d bias %.2f mode %d used %d equal %s inst %d ref %d rounds %d steps %d dLL %.2e" % (
            cellpx, bias, mode, st["mode_used"], ck.masks_equivalent(mask, classes, ref.mask, ref.object_class),
            len(classes), len(ref.object_class), st["rounds"], st["finisher_steps"],
            abs(st["total_logprob"] - ref.total_logprob) / abs(ref.total_logprob)))
