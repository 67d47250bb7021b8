"""Exact engine, images in flight side by side (MergerPool: a context, a stream and a host thread each).
Contexts are warmed first (workspace allocated), then N concurrent 512x1024 images are timed.
    python tests/tools/gpu_exact_concurrency.py [N ...]   (default 1 2 4 8 16)
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from mergenet_amd import synth, segmenter as seg

H, W, C = (int(os.environ.get("MN_H", 512)), int(os.environ.get("MN_W", 1024)), 9)
offs = synth.generate_offsets(40, 10)
im = synth.synth_v1(H, W, C, offs, 1000)
cp = torch.from_numpy(im.class_probs).cuda()
sp = torch.from_numpy(im.sameness_probs).cuda()
o = seg.default_options(merge_logprob_bias=0.03, mode=seg.MN_MODE_EXACT, clip_inputs=1)
for n in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]:
    pool = seg.MergerPool(H, W, C, len(offs), depth=n)
    small = synth.synth_v1(32, 64, C, offs, 1, num_instances=2)
    # warm: allocate every context's exact workspace for the full image (first EXACT call at this size)
    futs = [pool.submit(cp, sp, offs, o) for _ in range(n)] if os.environ.get("MN_WARM_FULL") else []
    for f in futs:
        f.result()
    torch.cuda.synchronize()
    t = time.perf_counter()
    futs = [pool.submit(cp, sp, offs, o) for _ in range(n)]
    res = [f.result() for f in futs]
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    steps = res[0][3]["finisher_steps"]
    print("%2d in flight: %.2f s for all, %.3f Mpixel/s, %.2f us per step per image" % (
        n, dt, n * H * W / dt / 1e6, dt / steps * 1e6), flush=True)
    pool.close()
