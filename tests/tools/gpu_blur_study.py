"""How the modes behave on maps whose certainty fades near instance boundaries (GPU box).

synth-v1 draws every value independently around 0.9 / 0.1; a network's maps are smooth and unsure
near boundaries.  Here the noise-free synth-v1 maps are box-blurred (radius r) before the noise is
added, so sameness and class values pass through 0.5 next to every boundary.  Reports, per image,
which mode AUTO ended in and whether AUTO / the rounds equal the CPU oracle.

    python tests/tools/gpu_blur_study.py [H W n_images radius noise]
"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from multiprocessing import Pool

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
NIMG = int(sys.argv[3]) if len(sys.argv) > 3 else 16
RADIUS = int(sys.argv[4]) if len(sys.argv) > 4 else 2
NOISE = float(sys.argv[5]) if len(sys.argv) > 5 else 0.05
C = 9


def box_blur(a, r):
    """Mean over a (2r+1)^2 window, edges replicated; a is [K,H,W]."""
    if r == 0:
        return a
    p = np.pad(a, ((0, 0), (r, r), (r, r)), mode="edge").astype(np.float64)
    cs = p.cumsum(1).cumsum(2)
    cs = np.pad(cs, ((0, 0), (1, 0), (1, 0)))
    k = 2 * r + 1
    s = cs[:, k:, k:] - cs[:, :-k, k:] - cs[:, k:, :-k] + cs[:, :-k, :-k]
    return (s / (k * k)).astype(np.float32)


def make(seed):
    from mergenet_amd import synth
    offs = synth.generate_offsets(40, 10)
    clean = synth.synth_v1(H, W, C, offs, seed, noise=0.0)
    noisy = synth.synth_v1(H, W, C, offs, seed, noise=NOISE)
    cp = box_blur(clean.class_probs, RADIUS) + (noisy.class_probs - clean.class_probs)
    sp = box_blur(clean.sameness_probs, RADIUS) + (noisy.sameness_probs - clean.sameness_probs)
    return offs, np.clip(cp, 0.01, 0.99).astype(np.float32), np.clip(sp, 0.01, 0.99).astype(np.float32)


def oracle_one(seed):
    from oracle import checker as ck
    offs, cp, sp = make(seed)
    t = time.time()
    r = ck.run_csegment(cp, sp, C, offs, 0.0, 1.0, 0.03)
    return seed, r.mask, r.object_class, r.total_logprob, time.time() - t


if __name__ == "__main__":
    from mergenet_amd import segmenter as seg, labels
    seeds = list(range(8000, 8000 + NIMG))
    with Pool(8) as pool:
        pending = pool.map_async(oracle_one, seeds)
        ctx = seg.HostContext(H, W, C, 10)
        got = {}
        for sd in seeds:
            offs, cp, sp = make(sd)
            for mode in ((seg.MN_MODE_EXACT, seg.MN_MODE_ROUNDS) if H * W <= 16384 else (seg.MN_MODE_AUTO, seg.MN_MODE_ROUNDS)):
                o = seg.default_options(mode=mode, clip_inputs=1, band_permille=int(__import__('os').environ.get('MN_BAND', '0')),
                                        subrounds=int(__import__('os').environ.get('MN_SUB', '0')),
                                        finish_limit=int(__import__('os').environ.get('MN_FINISH', '0')),
                                        core_radius=int(__import__('os').environ.get('MN_CORE', '0')))
                mask, classes, part, st = ctx.segment(cp, sp, offs, o)
                got[(sd, mode)] = (mask, classes, st)
        res = pending.get()
    tally = {0: 0, 1: 0, 2: 0}
    used = {1: 0, 2: 0, 3: 0}
    for sd, rmask, rcls, rlp, dt in res:
        line = "seed %d: ref %d instances (%.0f s)" % (sd, len(rcls), dt)
        for mode in ((1, 2) if H * W <= 16384 else (0, 2)):
            mask, classes, st = got[(sd, mode)]
            ok = labels.masks_equivalent(mask, classes, rmask, rcls)
            tally[mode] += int(ok)
            if mode in (0, 1):
                used[st["mode_used"]] += 1
            # pixels on which the two label maps agree under the best one-to-one matching by overlap
            a, b = mask.astype(np.int64).ravel(), rmask.astype(np.int64).ravel()
            cont = np.zeros((a.max() + 1, b.max() + 1), np.int64)
            np.add.at(cont, (a, b), 1)
            agree, c2 = 0, cont.copy()
            for _ in range(min(c2.shape)):
                i, j = np.unravel_index(np.argmax(c2), c2.shape)
                if c2[i, j] <= 0:
                    break
                agree += c2[i, j]
                c2[i, :] = -1
                c2[:, j] = -1
            line += " | mode %d used %d %s %d inst %.1f ms dLL %.1e differing pixels %d (%.3f%%)" % (
                mode, st["mode_used"], "EQUAL" if ok else "DIFF", len(classes), st["ms_total"],
                abs(st["total_logprob"] - rlp) / abs(rlp), a.size - agree, 100.0 * (a.size - agree) / a.size)
        print(line, flush=True)
    print("blur study %dx%d radius %d noise %.2f: AUTO %d/%d (ended in exact/rounds/components: %d/%d/%d), ROUNDS %d/%d equal to the oracle"
          % (H, W, RADIUS, NOISE, tally[0] + tally[1], NIMG, used[1], used[2], used[3], tally[2], NIMG))
