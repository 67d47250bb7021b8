"""Network-like (blurred) 1024x2048 maps through MergerPool: images per second with several images in
flight (each image's general path is a chain of short kernels and host round trips)."""
import sys, time
sys.path.insert(0, '.')
import torch
from mergenet_amd import synth, segmenter as seg
H, W, C = 1024, 2048, 9
offs = synth.generate_offsets(40, 10)
imgs = []
for sd in (4242, 4243, 4244, 4245):
    s = synth.blurred_v1(H, W, C, offs, sd, radius=2, noise=0.05)
    imgs.append((torch.from_numpy(s.class_probs).cuda(), torch.from_numpy(s.sameness_probs).cuda()))
opts = seg.default_options(merge_logprob_bias=0.03, clip_inputs=1)
m = seg.Merger(H, W, C, len(offs))
ref = [m.segment(cp, sp, offs, opts) for cp, sp in imgs]
torch.cuda.synchronize(); t = time.perf_counter()
n = 24
for i in range(n):
    m.segment(*imgs[i % 4], offs, opts)
torch.cuda.synchronize(); dt = time.perf_counter() - t
print("one at a time: %.2f ms per image (mode_used %d, %d instances)" % (dt / n * 1e3, ref[0][3]["mode_used"], ref[0][3]["num_instances"]), flush=True)
for depth in (2, 4, 8):
    pool = seg.MergerPool(H, W, C, len(offs), depth=depth)
    fs = [pool.submit(*imgs[i % 4], offs, opts) for i in range(2 * depth)]
    [f.result() for f in fs]
    torch.cuda.synchronize(); t = time.perf_counter()
    n = 48
    fs = [pool.submit(*imgs[i % 4], offs, opts) for i in range(n)]
    res = [f.result() for f in fs]
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    same = all(bool(torch.equal(res[i][0], ref[i % 4][0])) for i in range(n))
    print("pool depth %d: %.2f ms per image, results equal the serial ones: %s" % (depth, dt / n * 1e3, same), flush=True)
    pool.close()
