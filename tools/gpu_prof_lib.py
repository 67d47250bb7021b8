"""gpu_prof.py against another build of the library: python tools/gpu_prof_lib.py LIB H W mode."""
import sys
sys.path.insert(0, '.')
import torch
from mergenet_amd import synth, segmenter as seg
seg.LIB_PATH = sys.argv[1]
H, W = int(sys.argv[2]), int(sys.argv[3])
offs = synth.generate_offsets(40, 10)
s = synth.synth_v1(H, W, 9, offs, 1000)
m = seg.Merger(H, W, 9, len(offs))
cp = torch.from_numpy(s.class_probs).cuda(); sp = torch.from_numpy(s.sameness_probs).cuda()
o = seg.default_options(mode=int(sys.argv[4]))
for it in range(3):
    try:
        mask, table, part, st = m.segment(cp, sp, offs, o)
    except Exception as e:
        print("segment failed:", e); break
torch.cuda.synchronize()
