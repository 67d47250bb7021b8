"""Whole merger in the default mode, repeated with cold caches (for rocprofv3 --pmc / --kernel-trace
on the GPU box): python tools/prof_components.py [n]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mergenet_amd import synth, segmenter as seg
H, W = 1024, 2048
offs = synth.generate_offsets(40, 10)
# MN_PROF_BLUR=1: a network-like map (certainty fades at the boundaries: not sign-separable)
if os.environ.get('MN_PROF_BLUR') == '1':
    s = synth.blurred_v1(H, W, 9, offs, 4242, radius=2, noise=0.05)
else:
    s = synth.synth_v1(H, W, 9, offs, 1000)
m = seg.Merger(H, W, 9, len(offs))
cp = torch.from_numpy(s.class_probs).cuda(); sp = torch.from_numpy(s.sameness_probs).cuda()
# a 1 GiB scratch write between launches evicts the 256 MiB Infinity Cache so each image reads HBM
scratch = torch.empty(256 << 20, dtype=torch.float32, device='cuda')
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tot = 0.0
for it in range(n):
    scratch.fill_(float(it))
    torch.cuda.synchronize()
    _, _, _, st = m.segment(cp, sp, offs, seg.default_options(mode=int(os.environ.get('MN_PROF_MODE', '0')), require_proof=-1))
    tot += st["ms_total"]
print("mode_used %d, avg device time %.3f ms per image (cold caches), rounds %d, finisher steps %d, instances %d" % (
    st["mode_used"], tot / n, st["rounds"], st["finisher_steps"], st["num_instances"]))
