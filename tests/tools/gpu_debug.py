import sys, os, ctypes
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
if len(sys.argv) > 1 and sys.argv[1] == 'torch':
    import torch
    print("torch avail", torch.cuda.is_available(), torch.cuda.device_count(), torch.version.hip)
    maps = open('/proc/self/maps').read()
    print(sorted({l.split()[-1] for l in maps.splitlines() if 'amdhip' in l or 'hsa-runtime' in l}))
    from mergenet_amd import segmenter as seg
    lib = seg.load_library()
    maps = open('/proc/self/maps').read()
    print(sorted({l.split()[-1] for l in maps.splitlines() if 'amdhip' in l or 'hsa-runtime' in l}))
    hip = ctypes.CDLL('libamdhip64.so.7')
    n = ctypes.c_int(-1)
    rc = hip.hipGetDeviceCount(ctypes.byref(n)); print("hipGetDeviceCount rc", rc, n.value)
    hip.hipGetErrorString.restype = ctypes.c_char_p
    print(hip.hipGetErrorString(rc))
    x = torch.zeros(4, device='cuda'); print(x)
    rc = hip.hipGetDeviceCount(ctypes.byref(n)); print("after tensor: rc", rc, n.value)
    h = lib.mn_create(0, 16, 16, 3, 3); print("mn_create", h)
    sys.exit(0)
import golden_util as gu
from mergenet_amd import segmenter as seg
from oracle import checker as ck
for name in sys.argv[1:]:
    g = gu.load(name)
    H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
    ref = ck.run_csegment(g["class_probs"], g["sameness_probs"], C, g["offsets"], *g["spec"]["opts"])
    for mode in (seg.MN_MODE_EXACT, seg.MN_MODE_ROUNDS):
        ctx = seg.HostContext(H, W, C, len(g["offsets"]))
        sdb, omf, bias = g["spec"]["opts"]
        o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias, mode=mode, clip_inputs=1)
        mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
        ctx.close()
        print(name, "mode", mode, "same_partition", ck.same_partition(part, ref.partition), "mismatch_px", ck.partition_mismatch(part, ref.partition),
              "objs", st["num_objects"], ref.stats["n_objects"], "merges", st["merges"], ref.stats["n_merges"], "steps", st["finisher_steps"], ref.stats["n_live_pops"],
              "rounds", st["rounds"], "cert", st["certified"], "lp", st["total_logprob"], ref.total_logprob, "ms", round(st["ms_total"],3), flush=True)
