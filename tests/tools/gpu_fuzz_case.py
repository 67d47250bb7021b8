"""Re-run one case of the components-mode fuzz test with details (GPU box)."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from mergenet_amd import synth, segmenter as seg
from oracle import checker as ck

H, W, C = 47, 72, 9
offs = [(1, 0), (-1, 5)]
opts = (0.0, 1.0, 0.03)
s = synth.synth_v1(H, W, C, offs, 9004, noise=0.1)
ref = ck.run_csegment(s.class_probs, s.sameness_probs, C, offs, *opts)
for mode in (3, 2, 1):
    ctx = seg.HostContext(H, W, C, len(offs))
    o = seg.default_options(same_different_bias=opts[0], object_merge_factor=opts[1], merge_logprob_bias=opts[2],
                            mode=mode, clip_inputs=1)
    mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
    ctx.close()
    print("mode", mode, "used", st["mode_used"], "inst", len(classes), "ref", len(ref.object_class),
          "equal", ck.masks_equivalent(mask, classes, ref.mask, ref.object_class),
          "same_part", ck.same_partition(part, ref.partition), "objs", st["num_objects"],
          "ref objs", len(np.unique(ref.partition)), "rounds", st["rounds"], "steps", st["finisher_steps"],
          "viol", st["cert_edge_violations"], st["cert_class_violations"], st["cert_record_violations"])
    print("  classes gpu", sorted(classes)[:30], "\n  classes ref", sorted(ref.object_class)[:30])
