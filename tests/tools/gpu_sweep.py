"""Sweep (subrounds, finish_limit) on one golden fixture and report equality with the reference."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import golden_util as gu
from mergenet_amd import segmenter as seg
from oracle import checker as ck
for name in sys.argv[1:]:
  g = gu.load(name)
  H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
  ctx = seg.HostContext(H, W, C, len(g["offsets"]))
  for norefresh in (100, 50, 25):     # here: band factor in per mille (-1 = no band)
   for sub, fin in [(32, 8192), (16, 8192), (8, 8192), (32, 4096), (64, 8192)]:
    o = seg.default_options(mode=seg.MN_MODE_ROUNDS, subrounds=sub, finish_limit=fin)
    o.band_permille = norefresh
    mask, classes, part, st = ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
    eq = ck.masks_equivalent(mask, classes, g["mask"], g["object_class"])
    print(name, "band", norefresh, "sub", sub, "fin", fin, "equal", eq, "inst", st["num_instances"], "ref", len(g["object_class"]),
          "rounds", st["rounds"], "steps", st["finisher_steps"], "recviol", st["cert_record_violations"],
          "lp %.2f" % st["total_logprob"], "ms %.1f" % st["ms_total"], flush=True)
