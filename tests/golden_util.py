"""Loading of tests/golden/*.npz (vectors produced from the reference by make_golden.py)."""
import glob
import hashlib
import json
import os

import numpy as np

from mergenet_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def names(prefix):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, prefix + "*.npz")))


def closed_form(layout, H, W, C, offs):
    inst = np.zeros((H, W), np.int32)
    cls = [0]
    if layout == "single_instance":
        inst[:] = 1
        cls = [0, 2]
    elif layout == "two_halves":
        inst[:, : W // 2] = 1
        inst[:, W // 2:] = 2
        cls = [0, 1, 2]
    pix_cls = np.asarray(cls, np.int32)[inst]
    cp = np.where(np.arange(C)[:, None, None] == pix_cls[None], 0.9, 0.05).astype(np.float32)
    sp = np.ones((len(offs), H, W), np.float32)
    for k, (di, dj) in enumerate(offs):
        r0, r1 = max(0, -di), min(H, H - di)
        c0, c1 = max(0, -dj), min(W, W - dj)
        if r0 < r1 and c0 < c1:
            same = inst[r0:r1, c0:c1] == inst[r0 + di:r1 + di, c0 + dj:c1 + dj]
            sp[k, r0:r1, c0:c1] = np.where(same, 0.9, 0.1)
    return cp, sp


def load(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    spec = json.loads(str(z["spec"]))
    offs = synth.generate_offsets(*spec["offsets"])
    H, W, C = spec["H"], spec["W"], spec["C"]
    if spec["kind"] == "synth":
        s = synth.synth_v1(H, W, C, offs, spec["seed"], noise=spec.get("noise", 0.15),
                           num_instances=spec.get("num_instances"),
                           occlusion=spec.get("occlusion", False))
        cp, sp = s.class_probs, s.sameness_probs
    elif spec["kind"] == "blur":
        s = synth.blurred_v1(H, W, C, offs, spec["seed"], radius=spec["radius"], noise=spec["noise"],
                             num_instances=spec.get("num_instances"))
        cp, sp = s.class_probs, s.sameness_probs
    elif spec["kind"] == "checker":
        s = synth.checkerboard(H, W, C, offs, spec["cell_px"], spec["seed"])
        cp, sp = s.class_probs, s.sameness_probs
    elif spec["kind"] == "adversarial":
        s = synth.adversarial(H, W, C, offs, spec["seed"])
        cp, sp = s.class_probs, s.sameness_probs
    else:
        cp, sp = closed_form(spec["layout"], H, W, C, offs)
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(cp).tobytes())
    h.update(np.ascontiguousarray(sp).tobytes())
    assert h.hexdigest() == str(z["sha256"]), "generator drifted: inputs no longer match the fixture"
    return dict(spec=spec, offsets=offs, class_probs=cp, sameness_probs=sp, mask=z["mask"],
                object_class=[int(c) for c in z["object_class"]], error=str(z["error"]))


def sequential_proof(st):
    """What mn_stats.proof must say of an exact-engine result: 2 (the reference's order, nothing left to a tie
    rule) when every pop was forced, no tied pop had a conflict, or the reference's own order among equals was
    run; 3 otherwise."""
    from mergenet_amd import segmenter as seg
    forced = (st["tied_steps"] == 0 or st["tie_order_used"] == seg.MN_TIES_REFERENCE or
              st.get("tied_conflicts", 1) == 0)
    return seg.MN_PROOF_SEQUENTIAL if forced else seg.MN_PROOF_SEQUENTIAL_TIES
