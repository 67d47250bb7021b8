"""Exact engine, a batch of images in ONE launch of its loop (mn_segment_exact_batch: a workgroup per image).
    python tests/tools/gpu_exact_batch.py [N ...]   (default 1 4 16 32; MN_H, MN_W: image size)
Different seeds per image; every result is compared with the single-image call of the first image's seed.
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mergenet_amd import synth, segmenter as seg


def main():
    H, W, C = (int(os.environ.get("MN_H", 512)), int(os.environ.get("MN_W", 1024)), 9)
    offs = synth.generate_offsets(40, 10)
    counts = [int(a) for a in sys.argv[1:]] or [1, 4, 16, 32]
    nmax = max(counts)
    cps, sps = [], []
    for i in range(min(nmax, 8)):
        im = synth.synth_v1(H, W, C, offs, 1000 + i)
        cps.append(torch.from_numpy(im.class_probs).cuda())
        sps.append(torch.from_numpy(im.sameness_probs).cuda())
    o = seg.default_options(merge_logprob_bias=0.03, mode=seg.MN_MODE_EXACT, clip_inputs=1)
    ref = None
    if not os.environ.get("MN_NO_REF"):              # (single calls to compare with: slow at 1024x2048)
        one = seg.Merger(H, W, C, len(offs))
        ref = [one.segment(cps[i], sps[i], offs, o) for i in range(len(cps))]
        one.close()
    for n in counts:
        free0 = torch.cuda.mem_get_info()[0]
        b = seg.ExactBatch(H, W, C, len(offs), n)
        cp = [cps[i % len(cps)] for i in range(n)]
        sp = [sps[i % len(sps)] for i in range(n)]
        if os.environ.get("MN_WARM_FULL"):
            b.segment(cp, sp, offs, o)
        torch.cuda.synchronize()
        t = time.perf_counter()
        res = b.segment(cp, sp, offs, o)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        same = "not checked" if ref is None else all(
            torch.equal(res[i][0], ref[i % len(ref)][0]) and res[i][3]["num_objects"] == ref[i % len(ref)][3]["num_objects"]
            for i in range(n))
        steps = res[0][3]["finisher_steps"]
        print("%2d per launch: %.2f s for all, %.3f Mpixel/s, %.2f us per step per image, equal to single calls: %s" % (
            n, dt, n * H * W / dt / 1e6, dt / steps * 1e6, same), flush=True)
        print("    workspace per image %.2f GB (device memory in use %.1f GB)" % (b.mergers[0].workspace_bytes() / 1e9, (free0 - torch.cuda.mem_get_info()[0]) / 1e9), flush=True)
        b.close()


if __name__ == "__main__":
    main()
