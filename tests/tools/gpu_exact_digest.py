"""Masks, partitions and step counts of the exact engine on the given reference vectors, as JSON (the library
under test is chosen by MN_LIB: tests compare the product build with the -DMN_X_PARANOID build).
usage: python tests/tools/gpu_exact_digest.py name [name ...]"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import golden_util as gu
from mergenet_amd import segmenter as seg

out = {"lib": os.path.basename(seg.LIB_PATH)}
for name in sys.argv[1:]:
    g = gu.load(name)
    H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
    ctx = seg.HostContext(H, W, C, len(g["offsets"]))
    sdb, omf, bias = g["spec"]["opts"]
    o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias,
                            mode=seg.MN_MODE_EXACT, clip_inputs=1, tie_order=seg.MN_TIES_LOWEST_ID)
    mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
    ctx.close()
    out[name] = {"mask": hashlib.sha256(np.ascontiguousarray(mask).tobytes()).hexdigest(),
                 "partition": hashlib.sha256(np.ascontiguousarray(part).tobytes()).hexdigest(),
                 "classes": classes, "steps": st["finisher_steps"], "merges": st["merges"],
                 "tied_steps": st["tied_steps"], "tied_conflicts": st["tied_conflicts"]}
print(json.dumps(out))
