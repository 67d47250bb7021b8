"""The reference-order loop (mn_options.tie_order = MN_TIES_REFERENCE) on golden vectors: time, pops, equality.
    python tests/tools/gpu_reforder.py [name ...]      (default: a few small vectors and the tie-decided ones)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import golden_util as gu
    from mergenet_amd import labels, segmenter as seg
    names = sys.argv[1:] or ["cseg_adv_32x32_o0", "cseg_synth_32x64_n60", "cseg_blur_64x128_r2", "cseg_synth_64x128_n60",
                             "cseg_checker_96x128_b015", "cseg_blur4_128x256_s5100", "cseg_blur4_128x256_s5103"]
    for name in names:
        g = gu.load(name)
        H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
        sdb, omf, bias = g["spec"]["opts"]
        ctx = seg.HostContext(H, W, C, len(g["offsets"]))
        for ties in (seg.MN_TIES_LOWEST_ID, seg.MN_TIES_REFERENCE):
            o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias,
                                    mode=seg.MN_MODE_EXACT, clip_inputs=1, tie_order=ties)
            t = time.time()
            mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
            dt = time.time() - t
            eq = labels.masks_equivalent(mask, classes, g["mask"], g["object_class"])
            print("%-28s tie order %d: %6.2f s  steps %8d merges %7d proof %d  equal to the reference %s" % (
                name, ties, dt, st["finisher_steps"], st["merges"], st["proof"], eq), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
