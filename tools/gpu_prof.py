"""One timed run of the merger for rocprofv3 (GPU box): python tools/gpu_prof.py H W [subrounds] [finish]."""
import sys
sys.path.insert(0, '.')
import numpy as np, torch
from mergenet_amd import synth, segmenter as seg
H, W = int(sys.argv[1]), int(sys.argv[2])
sub = int(sys.argv[3]) if len(sys.argv) > 3 else 16
fin = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
offs = synth.generate_offsets(40, 10)
s = synth.synth_v1(H, W, 9, offs, 1000)
m = seg.Merger(H, W, 9, len(offs))
cp = torch.from_numpy(s.class_probs).cuda(); sp = torch.from_numpy(s.sameness_probs).cuda()
o = seg.default_options(mode=int(sys.argv[5]) if len(sys.argv) > 5 else seg.MN_MODE_AUTO, subrounds=sub, finish_limit=fin)
for it in range(2):
    mask, table, part, st = m.segment(cp, sp, offs, o, want_partition=True)
torch.cuda.synchronize()
print(st)
np.savez_compressed("gpurun_out/part_%dx%d.npz" % (H, W), part=part.cpu().numpy(), mask=mask.cpu().numpy(),
                    classes=table.cpu().numpy()[: st["num_instances"]], inst=s.instances)
