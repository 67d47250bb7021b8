"""Quick timing of the full merger on one synthetic image (GPU box)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from mergenet_amd import synth, segmenter as seg
H, W = int(sys.argv[1]), int(sys.argv[2]); C = int(sys.argv[3]) if len(sys.argv) > 3 else 9
offs = synth.generate_offsets(40, 10) if C == 9 else synth.generate_offsets(80, 16)
t = time.time(); s = synth.synth_v1(H, W, C, offs, 1000, occlusion=(C != 9)); print("gen %.1fs" % (time.time() - t), flush=True)
m = seg.Merger(H, W, C, len(offs)); print("workspace GB", m.workspace_bytes() / 1e9)
cp = torch.from_numpy(s.class_probs).cuda(); sp = torch.from_numpy(s.sameness_probs).cuda()
for sub in (8, 4, 16):
    for fin in (8192, 2048, 32768):
        o = seg.default_options(mode=seg.MN_MODE_ROUNDS, subrounds=sub, finish_limit=fin)
        best = None
        for it in range(3):
            torch.cuda.synchronize(); t = time.time()
            mask, table, part, st = m.segment(cp, sp, offs, o, want_partition=True)
            torch.cuda.synchronize(); dt = time.time() - t
            best = dt if best is None else min(best, dt)
        print("sub", sub, "fin", fin, "wall ms %.2f" % (best * 1e3), {k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items()}, flush=True)
gt_ok = None
from oracle import checker as ck
print("partition == generator ground truth:", ck.same_partition(part.cpu().numpy(), s.instances))
for it in range(5):
    a, b = m.score(cp, sp, offs, seg.default_options())
    print("score: class pass %.1f us, edge pass %.1f us" % (a * 1e3, b * 1e3))
