"""General rounds with and without the core contraction: time, rounds and agreement with the
reference's vectors (GPU box).  python tests/tools/gpu_cores_probe.py [quick]"""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import golden_util as gu
from mergenet_amd import segmenter as seg, labels, synth

quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
big = ["cseg_synth_1024x2048_cfg2", "cseg_synth_1024x2048_s1001"] + ([] if quick else ["cseg_synth_1024x2048_s1002", "cseg_synth_1024x2048_s1003"])
small = ["cseg_blur_64x128_r2", "cseg_blur_64x128_r2_s8001", "cseg_blur_256x512_r2", "cseg_crowd48_256x512_s6400",
         "cseg_crowd48_256x512_s6408", "cseg_checker_96x128_b015", "cseg_synth_512x1024_s1000"]
small += [n for n in gu.names("cseg_") if ("n35" in n or "n60" in n or "noise" in n)][:6]


def run(ctx, g, flags, **kw):
    o = seg.default_options(mode=seg.MN_MODE_ROUNDS, clip_inputs=1, core_radius=flags, **kw)
    mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
    eq = labels.masks_equivalent(mask, classes, g["mask"], g["object_class"])
    agree = labels.agreement(mask, g["mask"])
    return eq, agree, st, len(classes)


for group in (small, big):
    ctxs = {}
    for n in group:
        g = gu.load(n)
        H, W = g["mask"].shape
        C, O = g["class_probs"].shape[0], g["sameness_probs"].shape[0]
        key = (H, W, C, O)
        if key not in ctxs:
            ctxs[key] = seg.HostContext(H, W, C, O)
        ctx = ctxs[key]
        for flags, tag in ((-1, "strict"), (10, "r10   "), (6, "r6    ")):
            run(ctx, g, flags)
            eq, agree, st, K = run(ctx, g, flags)
            print("%-34s %s: equal %d  agree %8d / %8d (%.4f %%)  K %3d (ref %3d)  %.2f ms  rounds %3d  steps %5d  records %d" % (
                n, tag, eq, agree, H * W, 100.0 * agree / (H * W), K, len(g["object_class"]), st["ms_total"], st["rounds"],
                st["finisher_steps"], st["initial_records"]), flush=True)
    ctxs.clear()

# a network-like map at the benchmark size (no reference vector: 435 s per image on the CPU)
offs = synth.generate_offsets(40, 10)
s = synth.blurred_v1(1024, 2048, 9, offs, 4242, radius=2, noise=0.05)
ctx = seg.HostContext(1024, 2048, 9, 10)
res = {}
for flags, tag in ((-1, "strict"), (10, "r10   "), (6, "r6    ")):
    o = seg.default_options(mode=seg.MN_MODE_ROUNDS, clip_inputs=1, core_radius=flags)
    ctx.segment(s.class_probs, s.sameness_probs, offs, o)
    mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
    res[tag] = mask
    print("blurred 1024x2048 %s: K %d  %.2f ms  rounds %d  steps %d" % (tag, len(classes), st["ms_total"], st["rounds"], st["finisher_steps"]), flush=True)
print("blurred 1024x2048: the two agree on %d of %d pixels" % (labels.agreement(res["strict"], res["r6    "]), 1024 * 2048))
