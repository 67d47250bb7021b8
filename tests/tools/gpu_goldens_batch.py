"""All eight 1024x2048 golden images (configs[1-2]: the reference's own outputs, 380-540 s each there) through ONE
mn_segment_exact_batch launch with default options; every mask compared with the reference's.
    python tests/tools/gpu_goldens_batch.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import golden_util as gu
from mergenet_amd import segmenter as seg, labels

names = [n for n in gu.names("cseg_synth_1024x2048")]
gs = [gu.load(n) for n in names]
g0 = gs[0]
H, W, C = g0["spec"]["H"], g0["spec"]["W"], g0["spec"]["C"]
sdb, omf, bias = g0["spec"]["opts"]
o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias, mode=seg.MN_MODE_EXACT, clip_inputs=1)
cps = [torch.from_numpy(np.ascontiguousarray(g["class_probs"], dtype=np.float32)).cuda() for g in gs]
sps = [torch.from_numpy(np.ascontiguousarray(g["sameness_probs"], dtype=np.float32)).cuda() for g in gs]
batch = seg.ExactBatch(H, W, C, len(g0["offsets"]), len(gs))
torch.cuda.synchronize()
t = time.time()
res = batch.segment(cps, sps, g0["offsets"], o)
torch.cuda.synchronize()
dt = time.time() - t
bad = 0
for n, g, (mask, table, part, st) in zip(names, gs, res):
    ok = labels.masks_equivalent(mask.cpu().numpy(), seg._class_list(table.cpu().numpy()), g["mask"], g["object_class"])
    bad += 0 if ok else 1
    print("%-30s equal to the reference %s  proof %d steps %d merges %d tied pops %d conflicts %d instances %d" % (
        n, ok, st["proof"], st["finisher_steps"], st["merges"], st["tied_steps"], st["tied_conflicts"], st["num_instances"]), flush=True)
print("%d images in one launch: %.1f s; mismatching: %d" % (len(gs), dt, bad))
batch.close()
sys.exit(1 if bad else 0)
