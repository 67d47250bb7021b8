"""Timings of the "next" rows of SURVEY 8f at the benchmark shape (GPU box): torch events, 20 repeats."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from mergenet_amd import synth, segmenter as seg

H, W, C = 1024, 2048, 9
offs = synth.generate_offsets(40, 10)
s = synth.synth_v1(H, W, C, offs, 1000)
m = seg.Merger(H, W, C, len(offs))
cp = torch.from_numpy(s.class_probs).cuda(); sp = torch.from_numpy(s.sameness_probs).cuda()
mask, table, _, st = m.segment(cp, sp, offs, seg.default_options())
K = st["num_instances"]


def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3   # us


logits = torch.randn((C + len(offs), 512, 1024), device="cuda")          # network output at half size
rows = []
t = timed(lambda: m.prepare(logits, H, W, apply_sigmoid=True))
rows.append(("mn_prepare_device: 19 planes 512x1024 logits -> sigmoid, bilinear x2, clip", t,
             (logits.numel() + 19 * H * W) * 4))
t = timed(lambda: m.upsample_mask(mask, 2 * H, 2 * W))
rows.append(("mn_upsample_mask_device: 1024x2048 -> 2048x4096 nearest", t, (H * W + 4 * H * W) * 4))
t = timed(lambda: m.encode_rle(mask, K), n=5)
rows.append(("encode_rle: device run boundaries + host counts strings, %d instances" % K, t, H * W * 4))
t = timed(lambda: m.sameness_targets(mask, offs))
rows.append(("mn_sameness_targets_device: mask -> 10 target planes", t, (H * W + 10 * H * W) * 4))
t = timed(lambda: m.instance_scores(K))
rows.append(("mn_instance_scores_device", t, 0))
for name, us, nbytes in rows:
    print("%-80s %9.1f us  %6.2f TB/s" % (name, us, nbytes / us / 1e6 if nbytes else 0.0))
