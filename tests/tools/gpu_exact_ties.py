"""Tied pops of the exact engine on the reference vectors: for each cseg_* golden (up to 512x1024) the number of
pops at which a second live record held the bit-equal stored priority, and whether the result equals the vector.
    python tests/tools/gpu_exact_ties.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import golden_util as gu
    from mergenet_amd import labels, segmenter as seg
    for name in gu.names("cseg_"):
        g = gu.load(name)
        H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
        if H * W > 512 * 1024:
            continue
        sdb, omf, bias = g["spec"]["opts"]
        ctx = seg.HostContext(H, W, C, len(g["offsets"]))
        o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias,
                                mode=seg.MN_MODE_EXACT, clip_inputs=1)
        mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
        ctx.close()
        eq = labels.masks_equivalent(mask, classes, g["mask"], g["object_class"])
        print("%-34s steps %8d merges %7d tied steps %7d tied merges %7d equal %s" % (
            name, st["finisher_steps"], st["merges"], st["tied_steps"], st["tied_merges"], eq), flush=True)


if __name__ == "__main__":
    main()
