"""configs[4] (800x1333, C=81, O=16, occlusion layout) through the speculative fast path, repeated with cold
caches -- for rocprofv3 --kernel-trace / --pmc on the GPU box: python tools/prof_cfg5.py [n]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from mergenet_amd import synth, segmenter as seg
H, W, C = 800, 1333, 81
offs = synth.generate_offsets(80, 16)
s = synth.synth_v1(H, W, C, offs, 1000, occlusion=True)
m = seg.Merger(H, W, C, len(offs))
cp = torch.from_numpy(s.class_probs).cuda(); sp = torch.from_numpy(s.sameness_probs).cuda()
scratch = torch.empty(256 << 20, dtype=torch.float32, device='cuda')     # evicts the Infinity Cache between images
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
tot = sweep = 0.0
for it in range(n):
    scratch.fill_(float(it))
    torch.cuda.synchronize()
    _, _, _, st = m.segment(cp, sp, offs, seg.default_options(require_proof=-1))
    tot += st["ms_total"]; sweep += st["ms_cc_edges"]
print("mode_used %d, device %.3f ms per image, sweep %.4f ms by HIP events = %.3f of 8 TB/s on 413.76 MB, instances %d" % (
    st["mode_used"], tot / n, sweep / n, 4.0 * (C + len(offs)) * H * W / (sweep / n * 1e-3) / 8e12, st["num_instances"]))
