"""Parity campaign on fresh seeds at the reference's own working size (GPU box).

For each seed: synth-v1 512x1024 (C=9, O=10, Cityscapes options) -> HIP merger (ROUNDS, defaults)
and the CPU oracle (our restatement, pinned bit-for-bit to the reference; ~60 s per image, run in
a process pool).  Reports how many images give identical instance masks and class lists.
"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multiprocessing import Pool

H, W, C = (int(sys.argv[1]), int(sys.argv[2])) + (9,) if len(sys.argv) > 2 else (512, 1024, 9)
SEEDS = list(range(int(sys.argv[3]) if len(sys.argv) > 3 else 2000, (int(sys.argv[3]) if len(sys.argv) > 3 else 2000) + (int(sys.argv[4]) if len(sys.argv) > 4 else 16)))
NOISE = float(sys.argv[5]) if len(sys.argv) > 5 else 0.15
VARIANT = sys.argv[6] if len(sys.argv) > 6 else "csegment"     # or "pysegmenter" (options 0, 1/O, 0)
MODE = int(sys.argv[7]) if len(sys.argv) > 7 else 0            # 0 AUTO, 2 ROUNDS, 3 COMPONENTS
import os
INSTANCES = int(os.environ["MN_CAMPAIGN_INSTANCES"]) if "MN_CAMPAIGN_INSTANCES" in os.environ else None
BIAS = float(os.environ.get("MN_CAMPAIGN_BIAS", "0.03"))       # merge_logprob_bias (csegment variant)
FINISH = int(os.environ.get("MN_CAMPAIGN_FINISH", "0"))        # finish_limit (0 = library default)


def oracle_one(seed):
    from mergenet_amd import synth
    from oracle import checker as ck
    offs = synth.generate_offsets(40, 10)
    s = synth.synth_v1(H, W, C, offs, seed, noise=NOISE, num_instances=INSTANCES)
    t = time.time()
    if VARIANT == "pysegmenter":
        r = ck.run_pysegmenter(s.class_probs, s.sameness_probs, C, offs, 0.0, 1.0 / len(offs), 0.0)
    else:
        r = ck.run_csegment(s.class_probs, s.sameness_probs, C, offs, 0.0, 1.0, BIAS)
    return seed, r.mask, r.object_class, r.total_logprob, time.time() - t


if __name__ == "__main__":
    from mergenet_amd import synth, segmenter as seg, labels
    offs = synth.generate_offsets(40, 10)
    with Pool(8) as pool:
        pending = pool.map_async(oracle_one, SEEDS)
        ctx = seg.HostContext(H, W, C, len(offs))
        gpu = {}
        for seed in SEEDS:
            s = synth.synth_v1(H, W, C, offs, seed, noise=NOISE, num_instances=INSTANCES)
            if VARIANT == "pysegmenter":
                o = seg.default_options(mode=MODE, variant=seg.MN_VARIANT_PYSEGMENTER,
                                        object_merge_factor=1.0 / len(offs), merge_logprob_bias=0.0)
            else:
                o = seg.default_options(mode=MODE, merge_logprob_bias=BIAS, finish_limit=FINISH)
            mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
            gpu[seed] = (mask, classes, st)
        while not pending.ready():
            print("waiting for the CPU oracle ...", flush=True)
            pending.wait(45)
        res = pending.get()
    eq = 0
    for seed, rmask, rcls, rlp, dt in res:
        mask, classes, st = gpu[seed]
        ok = labels.masks_equivalent(mask, classes, rmask, rcls)
        eq += int(ok)
        print("seed %d: %s  mode_used %d  instances gpu %d ref %d  certified %d  gpu %.1f ms  oracle %.0f s  rel.loglik diff %.1e"
              % (seed, "EQUAL" if ok else "DIFFERENT", st["mode_used"], len(classes), len(rcls), st["certified"], st["ms_total"], dt,
                 abs(st["total_logprob"] - rlp) / abs(rlp)), flush=True)
    print("parity campaign %s mode %d %dx%d noise %.2f instances %s bias %.3f finish_limit %d: %d/%d images identical to the sequential reference order" % (VARIANT, MODE, H, W, NOISE, INSTANCES, BIAS, FINISH, eq, len(SEEDS)))
