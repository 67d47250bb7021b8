"""Phase A only, repeated (for rocprofv3 --pmc / --kernel-trace on the GPU box)."""
import sys
sys.path.insert(0, '.')
import torch
from mergenet_amd import synth, segmenter as seg
H, W = 1024, 2048
offs = synth.generate_offsets(40, 10)
s = synth.synth_v1(H, W, 9, offs, 1000)
m = seg.Merger(H, W, 9, len(offs))
cp = torch.from_numpy(s.class_probs).cuda(); sp = torch.from_numpy(s.sameness_probs).cuda()
# a 1 GiB scratch write between launches evicts the 256 MiB Infinity Cache so each launch reads HBM
scratch = torch.empty(256 << 20, dtype=torch.float32, device='cuda')
tot = [0.0, 0.0]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for it in range(n):
    scratch.fill_(float(it))
    a, b = m.score(cp, sp, offs, seg.default_options())
    tot[0] += a; tot[1] += b
print("avg class pass %.2f us, edge pass %.2f us (cold caches)" % (tot[0] / n * 1e3, tot[1] / n * 1e3))
