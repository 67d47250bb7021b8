"""Which inputs does the order among bit-equal priorities decide?  Runs the CPU model of the exact engine's
semantics (lowest record id first) beside the oracle (the reference's heap order) -- no GPU needed.
usage: python tests/tools/tie_search.py H W radius seed0 n"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mergenet_amd import synth, labels
from oracle import checker as ck
import exact_model

H, W, radius, seed0, n = (int(a) for a in sys.argv[1:6])
offs = synth.generate_offsets(40, 10)
for seed in range(seed0, seed0 + n):
    s = synth.blurred_v1(H, W, 9, offs, seed, radius=radius, noise=0.05)
    t = time.time()
    ref = ck.run_csegment(s.class_probs, s.sameness_probs, 9, offs, 0.0, 1.0, 0.03)
    t1 = time.time()
    part, ocls, st = exact_model.run(s.class_probs, s.sameness_probs, offs, 1.0, 0.03)
    t2 = time.time()
    same = ck.same_partition(part, ref.partition)
    print("seed %d: model == oracle partition: %s; oracle %.1fs (live pops %d merges %d) model %.1fs %s" %
          (seed, same, t1 - t, ref.stats["n_live_pops"], ref.stats["n_merges"], t2 - t1, st), flush=True)
