"""Campaign of the exact engine (MN_MODE_EXACT) against the CPU oracle on FRESH inputs (GPU box).

Families (128x256 unless stated, C=9, O=10 generate_offsets(40,10) unless stated):
  blur     blurred_v1 radius 2 / 4 (network-like, order-dependent)
  crowd    synth_v1 with 24 overlapping instances at 128x256
  noise    synth_v1 at noise 0.35 / 0.6 (ties: a third of the values clipped)
  (blur4 also holds ties: the out-of-image sameness value 1.0 blurs into the maps near the image border and
   is clipped to 0.99 there; where the order among bit-equal priorities decides, the engine's rule -- lowest
   record id -- and the reference's heap mechanics can part ways: reported, with the pixel agreement)
  adv      uniform(0.02, 0.98) everything, 48x64 .. 96x128, other options
  c81      C=81, O=16 generate_offsets(80,16), occlusion layout, 96x160
For every image: the reference's partition, background set and classes (masks_equivalent), the same
partition as pixel sets (same_partition), the log-likelihood within 1e-5 relative, the merge count, and
-- on the tie-free families (blur, adv) -- the same number of live pops as the oracle's run.
    python tests/tools/gpu_exact_campaign.py [seeds per family, default 8]
"""
import os
import sys
import time
from multiprocessing import Pool

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np



def make(family, seed):
    from mergenet_amd import synth
    offs = synth.generate_offsets(40, 10)
    opts = (0.0, 1.0, 0.03)
    if family.startswith("blur"):
        s = synth.blurred_v1(128, 256, 9, offs, seed, radius=int(family[4:]), noise=0.05)
    elif family == "crowd":
        s = synth.synth_v1(128, 256, 9, offs, seed, num_instances=24)
    elif family.startswith("noise"):
        s = synth.synth_v1(128, 256, 9, offs, seed, noise=float(family[5:]) / 100.0)
    elif family == "adv":
        rng = np.random.default_rng(seed)
        H, W = int(rng.integers(48, 97)), int(rng.integers(64, 129))
        offs = synth.generate_offsets(int(rng.integers(6, 20)), int(rng.integers(4, 9)))
        s = synth.adversarial(H, W, int(rng.integers(2, 8)), offs, seed)
        opts = [(0.0, 1.0, 0.03), (0.3, 0.5, 0.0), (0.0, 2.0, 0.1), (-0.2, 0.25, 0.0)][seed % 4]
    else:
        offs = synth.generate_offsets(80, 16)
        s = synth.synth_v1(96, 160, 81, offs, seed, occlusion=True)
    return s, offs, opts


def oracle_one(job):
    from oracle import checker as ck
    family, seed = job
    s, offs, opts = make(family, seed)
    r = ck.run_csegment(s.class_probs, s.sameness_probs, s.class_probs.shape[0], offs, *opts)
    return family, seed, r.mask, r.object_class, r.partition, r.total_logprob, r.stats["n_merges"], (r.stats["n_live_pops"], r.stats["n_pops"])


if __name__ == "__main__":
    NSEED = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    from mergenet_amd import segmenter as seg
    from oracle import checker as ck
    fams = ["blur2", "blur4", "crowd", "noise35", "noise60", "adv", "c81"]
    jobs = [(f, 5000 + 100 * i + k) for i, f in enumerate(fams) for k in range(NSEED)]
    t0 = time.time()
    with Pool(min(16, os.cpu_count() or 4)) as pool:
        refs = pool.map(oracle_one, jobs)
    print("oracle: %d images in %.1f s" % (len(jobs), time.time() - t0), flush=True)
    tally = {f: [0, 0, 0, 0, 0, 0] for f in fams}  # equal, same event count, total, images with tied pops, tied steps, tied merges
    bad = []
    used = {}
    tg = 0.0
    for (family, seed, rmask, rcls, rpart, rlp, rmerges, rpops) in refs:
        s, offs, opts = make(family, seed)
        C, H, W = s.class_probs.shape
        ctx = seg.HostContext(H, W, C, len(offs))
        o = seg.default_options(same_different_bias=opts[0], object_merge_factor=opts[1],
                                merge_logprob_bias=opts[2], mode=seg.MN_MODE_EXACT, clip_inputs=1,
                                tie_order=int(os.environ.get("MN_TIES", seg.MN_TIES_LOWEST_ID)))
        t = time.time()
        mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
        tg += time.time() - t
        ctx.close()
        ok = ck.masks_equivalent(mask, classes, rmask, rcls) and ck.same_partition(part, rpart) and \
            abs(st["total_logprob"] - rlp) <= 1e-5 * abs(rlp) and st["merges"] == rmerges
        ev = st["finisher_steps"] == rpops[1 if st["tie_order_used"] == seg.MN_TIES_REFERENCE else 0]
        used[st["tie_order_used"]] = used.get(st["tie_order_used"], 0) + 1
        print("  %-8s seed %d: %5.1f s, tie order used %d, tied steps %d, equal %s" % (
            family, seed, time.time() - t, st["tie_order_used"], st["tied_steps"], ok), flush=True)
        tally[family][0] += int(ok); tally[family][1] += int(ev); tally[family][2] += 1
        tally[family][3] += int(st["tied_steps"] > 0); tally[family][4] += st["tied_steps"]; tally[family][5] += st["tied_merges"]
        if not ok:
            from mergenet_amd import labels
            agree = labels.agreement(mask, rmask) / float(mask.size)
            bad.append((family, seed, "instances %d vs %d" % (len(classes), len(rcls)), "pixels agreeing %.4f" % agree,
                        "tied steps %d merges %d" % (st["tied_steps"], st["tied_merges"])))
    for f in fams:
        print("%-8s equal to the oracle %2d / %2d   same number of pops as the oracle %2d / %2d   images with tied pops %2d "
              "(tied steps %d, of which merges %d)" %
              (f, tally[f][0], tally[f][2], tally[f][1], tally[f][2], tally[f][3], tally[f][4], tally[f][5]))
    print("exact engine: %.1f s for %d images (tie order used: %s); mismatching: %s" % (tg, len(jobs), used, bad))
    sys.exit(1 if bad else 0)
