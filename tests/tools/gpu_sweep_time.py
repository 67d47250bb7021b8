"""The sweep alone, back to back (mn_sweep_time_device), for one or several builds of the library:
    python tests/tools/gpu_sweep_time.py [tag ...]     (tag: mergenet_amd/libmergenet_hip_<tag>.so; none: the default)
Each tag runs in a child process (a process binds one library).  4 input sets of 1024x2048 (C=9, O=10) in rotation:
638 MB, beyond the 256 MB Infinity Cache.  MN_H, MN_W, MN_C, MN_OA ("80,16"): another shape."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child():
    sys.path.insert(0, ROOT)
    import torch
    from mergenet_amd import synth, segmenter as seg
    H, W, C = int(os.environ.get("MN_H", 1024)), int(os.environ.get("MN_W", 2048)), int(os.environ.get("MN_C", 9))
    oa = tuple(int(x) for x in os.environ.get("MN_OA", "40,10").split(","))
    offs = synth.generate_offsets(*oa)
    ins = []
    for i in range(4):
        im = synth.synth_v1(H, W, C, offs, 1000 + i, occlusion=(C == 81))
        ins.append((torch.from_numpy(im.class_probs).cuda(), torch.from_numpy(im.sameness_probs).cuda()))
    m = seg.Merger(H, W, C, len(offs))
    o = seg.default_options(merge_logprob_bias=0.03)
    res = [m.sweep_time(ins, offs, o, reps=400) for _ in range(3)]
    one = m.sweep_time(ins[:1], offs, o, reps=400)
    nbytes = 4.0 * (C + len(offs)) * H * W
    print("%-8s sweep alone, back to back: %s us per launch (4 input sets in rotation) -> %.3f of 8 TB/s; one input set %.2f us" % (
        os.environ.get("MN_TAG", "default"), " ".join("%.2f" % r for r in res), nbytes / (min(res) * 1e-6) / 8e12, one), flush=True)


if __name__ == "__main__":
    if os.environ.get("MN_CHILD"):
        child()
    else:
        for tag in sys.argv[1:] or [""]:
            env = dict(os.environ, MN_CHILD="1", MN_TAG=tag or "default")
            if tag:
                env["MN_LIB"] = os.path.join(ROOT, "mergenet_amd", "libmergenet_hip_%s.so" % tag)
            subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, check=False)
