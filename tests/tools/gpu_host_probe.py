"""Where the host's time goes in the bench loop (GPU box): launch vs read-back, with and without the
per-kernel events."""
import sys, time, faulthandler
faulthandler.dump_traceback_later(45, exit=True)
sys.path.insert(0, '.')
import torch
from collections import deque
from mergenet_amd import synth, segmenter as seg
H, W, C = 1024, 2048, 9
offs = synth.generate_offsets(40, 10)
imgs = []
for sd in (1000, 1001, 1002, 1003):
    s = synth.synth_v1(H, W, C, offs, sd)
    imgs.append((torch.from_numpy(s.class_probs).cuda(), torch.from_numpy(s.sameness_probs).cuda()))
for nctx in (8,):
    ring_m = [seg.Merger(H, W, C, len(offs)) for _ in range(nctx)]
    outs = [(torch.empty((H, W), dtype=torch.int32, device='cuda'), torch.empty((H * W,), dtype=torch.int32, device='cuda')) for _ in range(nctx)]
    for flags in (0, 16, 16 | 32):
        opts = seg.default_options(merge_logprob_bias=0.03, debug_flags=flags, require_proof=seg.MN_PROVE_NEVER)
        for rep in range(2):
            ring = deque(); tl = tr = 0.0
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 1000
            for i in range(n):
                a = time.perf_counter()
                ring.append(ring_m[i % nctx].segment_async(*imgs[i % 4], offs, opts, out=outs[i % nctx]))
                b = time.perf_counter(); tl += b - a
                if len(ring) >= nctx:
                    ring.popleft().result()
                    tr += time.perf_counter() - b
            while ring: ring.popleft().result()
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("contexts %d flags %d: %.1f us/step; host in launch %.1f us, in read-back %.1f us" % (nctx, flags, dt / n * 1e6, tl / n * 1e6, tr / n * 1e6), flush=True)

# the replayed result equals the ordinary one
m = seg.Merger(H, W, C, len(offs))
ref, tab, _, st0 = m.segment(*imgs[1], offs, seg.default_options(merge_logprob_bias=0.03, require_proof=seg.MN_PROVE_NEVER))
o = seg.default_options(merge_logprob_bias=0.03, debug_flags=48, require_proof=seg.MN_PROVE_NEVER)
out = (torch.empty((H, W), dtype=torch.int32, device='cuda'), torch.empty((H * W,), dtype=torch.int32, device='cuda'))
for k in range(5):
    out[0].zero_()
    mk, tb, _, st = m.segment_async(*imgs[1], offs, o, out=out).result()
    print("call %d: equal %s, instances %d (%d), loglik %.6f (%.6f), ms_cc_edges %.4f" % (k, bool(torch.equal(mk, ref)) and bool(torch.equal(tb[:st["num_instances"]], tab[:st0["num_instances"]])),
          st["num_instances"], st0["num_instances"], st["total_logprob"], st0["total_logprob"], st["ms_cc_edges"]), flush=True)
