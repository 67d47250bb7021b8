"""How far ROUNDS mode is from the sequential reference order on NON-separable inputs (study)."""
import sys
sys.path.insert(0, '.')
import numpy as np
from mergenet_amd import synth, segmenter as seg
from oracle import checker as ck
offs = synth.generate_offsets(40, 10)
ctx = seg.HostContext(128, 256, 9, len(offs))
for noise in (0.40, 0.45, 0.5, 0.6):
    eq = 0; mism = []; dlp = []
    for seed in range(3000, 3008):
        s = synth.synth_v1(128, 256, 9, offs, seed, noise=noise)
        ref = ck.run_csegment(s.class_probs, s.sameness_probs, 9, offs, 0.0, 1.0, 0.03)
        o = seg.default_options(mode=seg.MN_MODE_ROUNDS, clip_inputs=1)
        mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
        same = ck.same_partition(part, ref.partition)
        eq += int(same)
        if not same:
            # pixels outside the 1:1 matched parts
            mism.append(ck.partition_mismatch(part, ref.partition))
        dlp.append((st["total_logprob"] - ref.total_logprob) / abs(ref.total_logprob))
    print("noise %.2f: %d/8 partitions equal; mismatching px (of 32768) in the others: %s; rel. log-lik diff min %.2e max %.2e"
          % (noise, eq, mism, min(dlp), max(dlp)), flush=True)
