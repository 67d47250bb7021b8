"""The campaign of tests/tools/gpu_exact_campaign.py in the library's DEFAULT tie mode through BATCHES
(mn_segment_exact_batch): images of one shape and one option set share a launch of the exact engine, and those
whose tied choices conflict are redone TOGETHER by the reference-order loop (a workgroup per image).  Every image:
the oracle's partition, background set, classes, log-likelihood (1e-5), merge count and -- where the reference's
order was run -- the oracle's number of pops.
    python tests/tools/gpu_exact_campaign_batch.py [seeds per family, default 12]
"""
import os
import sys
import time
from multiprocessing import Pool

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from gpu_exact_campaign import make, oracle_one

if __name__ == "__main__":
    NSEED = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    import torch
    from mergenet_amd import segmenter as seg
    from oracle import checker as ck
    fams = ["blur2", "blur4", "crowd", "noise35", "noise60", "adv", "c81"]
    jobs = [(f, 5000 + 100 * i + k) for i, f in enumerate(fams) for k in range(NSEED)]
    t0 = time.time()
    with Pool(min(16, os.cpu_count() or 4)) as pool:
        refs = pool.map(oracle_one, jobs)
    print("oracle: %d images in %.1f s" % (len(jobs), time.time() - t0), flush=True)
    groups = {}
    for ref in refs:
        family, seed = ref[0], ref[1]
        s, offs, opts = make(family, seed)
        key = (s.class_probs.shape, tuple(map(tuple, offs)), tuple(opts))
        groups.setdefault(key, []).append((ref, s, offs, opts))
    tally = {f: [0, 0, 0, 0] for f in fams}     # equal, pops equal (where the reference's order ran), total, redone
    bad = []
    tg = 0.0
    for key, items in groups.items():
        (C, H, W), offs, opts = key[0], items[0][2], items[0][3]
        o = seg.default_options(same_different_bias=opts[0], object_merge_factor=opts[1], merge_logprob_bias=opts[2],
                                mode=seg.MN_MODE_EXACT, clip_inputs=1)
        cps = [torch.from_numpy(np.ascontiguousarray(it[1].class_probs, dtype=np.float32)).cuda() for it in items]
        sps = [torch.from_numpy(np.ascontiguousarray(it[1].sameness_probs, dtype=np.float32)).cuda() for it in items]
        batch = seg.ExactBatch(H, W, C, len(offs), len(items))
        torch.cuda.synchronize()
        t = time.time()
        res = batch.segment(cps, sps, offs, o, want_partition=True)
        torch.cuda.synchronize()
        dt = time.time() - t
        tg += dt
        n_ref = sum(1 for r in res if r[3]["tie_order_used"] == seg.MN_TIES_REFERENCE)
        print("  batch of %3d images %dx%d C=%d O=%d: %.1f s, %d redone in the reference's order" % (
            len(items), H, W, C, len(offs), dt, n_ref), flush=True)
        for (ref, s, _, _), (mask, table, part, st) in zip(items, res):
            family, seed, rmask, rcls, rpart, rlp, rmerges, rpops = ref
            mask = mask.cpu().numpy(); part = part.cpu().numpy()
            classes = seg._class_list(table.cpu().numpy())
            ok = ck.masks_equivalent(mask, classes, rmask, rcls) and ck.same_partition(part, rpart) and \
                abs(st["total_logprob"] - rlp) <= 1e-5 * abs(rlp) and st["merges"] == rmerges
            redone = st["tie_order_used"] == seg.MN_TIES_REFERENCE
            ev = (st["finisher_steps"] == rpops[1]) if redone else (st["proof"] == seg.MN_PROOF_SEQUENTIAL)
            tally[family][0] += int(ok); tally[family][1] += int(ev); tally[family][2] += 1; tally[family][3] += int(redone)
            if not ok or not ev:
                bad.append((family, seed, ok, ev, st["tie_order_used"], st["proof"], st["tied_steps"], st["tied_conflicts"]))
        batch.close()
    for f in fams:
        print("%-8s equal to the oracle %2d / %2d   redone in the reference's order %2d (oracle's pop count, or proof 2 "
              "without it: %2d / %2d)" % (f, tally[f][0], tally[f][2], tally[f][3], tally[f][1], tally[f][2]))
    print("batches: %.1f s for %d images; mismatching: %s" % (tg, len(jobs), bad))
    sys.exit(1 if bad else 0)
