#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (needs /root/reference; the GPU box never runs this):

    python tests/golden/make_golden.py [--big]

* ``cseg_*.npz``  expected outputs of the reference C++ merger: its own
  ``utils/csegment/segment.cc`` compiled unmodified into ``oracle/_ref/`` by ``oracle/Makefile``
  and called through the ABI at ``segment.cc:742-754`` with the binding's preprocessing
  (``c_segment.pyx:53-84``).
* ``py_*.npz``    expected outputs of the reference Python merger, obtained by importing
  ``utils.segmenter.ObjectSegmenter`` from /root/reference on float64 inputs
  (``utils/segmenter.py:225-483``); ``prune`` disabled variants patch the instance's
  ``prune`` attribute with a no-op to expose the raw merge result.

A fixture holds DATA only: the generator call that rebuilds the inputs from
``mergenet_amd.synth`` (plus a sha256 of the input bytes), the options, and the reference's
mask / class list.  No reference source text is stored.
"""

from __future__ import annotations

import argparse
import contextlib
import hashlib
import io
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HUGE = bool(int(os.environ.get("MN_GOLDEN_HUGE", "0")))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from mergenet_amd import synth  # noqa: E402
from oracle import checker as ck  # noqa: E402


def make_inputs(spec: dict):
    kind = spec["kind"]
    offs = synth.generate_offsets(*spec["offsets"])
    H, W, C = spec["H"], spec["W"], spec["C"]
    if kind == "synth":
        s = synth.synth_v1(H, W, C, offs, spec["seed"], noise=spec.get("noise", 0.15),
                           num_instances=spec.get("num_instances"),
                           occlusion=spec.get("occlusion", False))
        return s.class_probs, s.sameness_probs, offs
    if kind == "adversarial":
        s = synth.adversarial(H, W, C, offs, spec["seed"])
        return s.class_probs, s.sameness_probs, offs
    if kind == "blur":
        s = synth.blurred_v1(H, W, C, offs, spec["seed"], radius=spec["radius"], noise=spec["noise"],
                             num_instances=spec.get("num_instances"))
        return s.class_probs, s.sameness_probs, offs
    if kind == "checker":
        s = synth.checkerboard(H, W, C, offs, spec["cell_px"], spec["seed"])
        return s.class_probs, s.sameness_probs, offs
    if kind == "closed_form":
        return closed_form(spec["layout"], H, W, C, offs) + (offs,)
    raise ValueError(kind)


def closed_form(name: str, H: int, W: int, C: int, offs):
    """Closed-form layouts of SURVEY.md section 8c: no noise, exact expected answers."""
    inst = np.zeros((H, W), np.int32)
    cls = [0]
    if name == "all_background":
        pass
    elif name == "single_instance":
        inst[:] = 1
        cls = [0, 2]
    elif name == "two_halves":
        inst[:, : W // 2] = 1
        inst[:, W // 2:] = 2
        cls = [0, 1, 2]
    else:
        raise ValueError(name)
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


def digest(cp, sp) -> str:
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(cp).tobytes())
    h.update(np.ascontiguousarray(sp).tobytes())
    return h.hexdigest()


def cseg_specs(big: bool):
    specs = []
    adv_opts = [(0.0, 1.0, 0.03), (0.0, 0.25, 0.0), (0.5, 1.0, 0.0)]
    for i, n in enumerate([8, 16, 24, 32, 48, 64]):
        for j, o in enumerate(adv_opts):
            specs.append(dict(name="cseg_adv_%dx%d_o%d" % (n, n, j), kind="adversarial", H=n, W=n,
                              C=4, offsets=[6, 5], seed=100 + i, opts=o))
    specs.append(dict(name="cseg_adv_20x36_o0", kind="adversarial", H=20, W=36, C=3,
                      offsets=[10, 6], seed=120, opts=(0.0, 1.0, 0.03)))
    for noise in (0.15, 0.35, 0.6):
        specs.append(dict(name="cseg_synth_32x64_n%02d" % int(noise * 100), kind="synth", H=32,
                          W=64, C=9, offsets=[40, 10], seed=1000, noise=noise, num_instances=4,
                          opts=(0.0, 1.0, 0.03)))
        specs.append(dict(name="cseg_synth_64x128_n%02d" % int(noise * 100), kind="synth", H=64,
                          W=128, C=9, offsets=[40, 10], seed=1001, noise=noise, num_instances=4,
                          opts=(0.0, 1.0, 0.03)))
    specs.append(dict(name="cseg_synth_48x80_c81", kind="synth", H=48, W=80, C=81, offsets=[80, 16],
                      seed=1005, noise=0.15, num_instances=4, occlusion=True,
                      opts=(0.0, 1.0, 0.03)))
    for nm in ("all_background", "single_instance", "two_halves"):
        specs.append(dict(name="cseg_closed_%s" % nm, kind="closed_form", layout=nm, H=24, W=40, C=3,
                          offsets=[10, 6], seed=0, opts=(0.0, 1.0, 0.03)))
    specs.append(dict(name="cseg_synth_128x256", kind="synth", H=128, W=256, C=9, offsets=[40, 10],
                      seed=1000, noise=0.15, opts=(0.0, 1.0, 0.03)))
    # order-dependent inputs (round 2): maps whose certainty fades at the instance boundaries
    # (blurred), images crowded with 48 overlapping instances, a bias-dominated checkerboard
    specs.append(dict(name="cseg_blur_64x128_r2", kind="blur", H=64, W=128, C=9, offsets=[40, 10],
                      seed=8000, radius=2, noise=0.05, opts=(0.0, 1.0, 0.03)))
    specs.append(dict(name="cseg_blur_64x128_r2_s8001", kind="blur", H=64, W=128, C=9, offsets=[40, 10],
                      seed=8001, radius=2, noise=0.05, opts=(0.0, 1.0, 0.03)))
    specs.append(dict(name="cseg_checker_96x128_b015", kind="checker", H=96, W=128, C=3,
                      offsets=[6, 4], seed=4, cell_px=4, opts=(0.0, 1.0, 0.15)))
    # (round 3) a stronger blur carries the out-of-image sameness value 1.0 into the maps near the image
    # border, where it is clipped to 0.99: bit-equal priorities whose ORDER decides the result
    for sd in (5100, 5103):
        specs.append(dict(name="cseg_blur4_128x256_s%d" % sd, kind="blur", H=128, W=256, C=9, offsets=[40, 10],
                          seed=sd, radius=4, noise=0.05, opts=(0.0, 1.0, 0.03)))
    if big:
        # (round 4) a tie-decided vector ABOVE MN_TIE_LIMIT_RECORDS (1.25 M records): the exact engine's own rule
        # (lowest record id) differs from the reference here (found with tests/tools/tie_search.py: CPU model of
        # the engine's semantics beside the oracle)
        specs.append(dict(name="cseg_blur4_256x512_s5200", kind="blur", H=256, W=512, C=9, offsets=[40, 10],
                          seed=5200, radius=4, noise=0.05, opts=(0.0, 1.0, 0.03)))
        specs.append(dict(name="cseg_blur_256x512_r2", kind="blur", H=256, W=512, C=9,
                          offsets=[40, 10], seed=8000, radius=2, noise=0.05, opts=(0.0, 1.0, 0.03)))
        for sd in (6400, 6408):
            specs.append(dict(name="cseg_crowd48_256x512_s%d" % sd, kind="synth", H=256, W=512, C=9,
                              offsets=[40, 10], seed=sd, noise=0.15, num_instances=48,
                              opts=(0.0, 1.0, 0.03)))
    if big:
        specs.append(dict(name="cseg_synth_256x512", kind="synth", H=256, W=512, C=9,
                          offsets=[40, 10], seed=1000, noise=0.15, opts=(0.0, 1.0, 0.03)))
    if HUGE:  # BASELINE.json configs[1]: 1024x2048 (the reference needs ~9 min and 6.5 GB each)
        specs.append(dict(name="cseg_synth_1024x2048_cfg2", kind="synth", H=1024, W=2048, C=9,
                          offsets=[40, 10], seed=1000, noise=0.15, opts=(0.0, 1.0, 0.03)))
        for sd in (1001, 1002, 1003, 1004, 1005, 1006, 1007):   # images of ranks 1..7 (configs[2])
            specs.append(dict(name="cseg_synth_1024x2048_s%d" % sd, kind="synth", H=1024, W=2048,
                              C=9, offsets=[40, 10], seed=sd, noise=0.15, opts=(0.0, 1.0, 0.03)))
        # BASELINE.json configs[4]: COCO-shape maps, 81 classes, extended offset set, occlusion
        specs.append(dict(name="cseg_synth_400x667_c81", kind="synth", H=400, W=667, C=81,
                          offsets=[80, 16], seed=1000, noise=0.15, occlusion=True,
                          opts=(0.0, 1.0, 0.03)))
        specs.append(dict(name="cseg_synth_800x1333_cfg5", kind="synth", H=800, W=1333, C=81,
                          offsets=[80, 16], seed=1000, noise=0.15, occlusion=True,
                          opts=(0.0, 1.0, 0.03)))
        # (round 4) network-like maps at the BASELINE size: continuous values (certainty fades at the instance
        # boundaries), so bit-equal priorities are rare and far apart -- what the default mode's proof looks like
        # on inputs without synth-v1's plateaus
        specs.append(dict(name="cseg_blur_1024x2048_r2_s4242", kind="blur", H=1024, W=2048, C=9, offsets=[40, 10],
                          seed=4242, radius=2, noise=0.05, opts=(0.0, 1.0, 0.03)))
        specs.append(dict(name="cseg_blur_512x1024_r2_s4243", kind="blur", H=512, W=1024, C=9, offsets=[40, 10],
                          seed=4243, radius=2, noise=0.05, opts=(0.0, 1.0, 0.03)))
        for sd in (1000, 1001, 1002):  # the size the reference's own caller uses (segment.py:93)
            specs.append(dict(name="cseg_synth_512x1024_s%d" % sd, kind="synth", H=512, W=1024,
                              C=9, offsets=[40, 10], seed=sd, noise=0.15, opts=(0.0, 1.0, 0.03)))
    return specs


def py_specs(big: bool):
    specs = []
    for i, n in enumerate([12, 16, 24, 32, 40]):
        for j, o in enumerate([(0.0, 0.2, 0.0), (0.0, 1.0, 0.0), (0.3, 0.5, 0.5)]):
            for prune in (False, True):
                specs.append(dict(name="py_adv_%dx%d_o%d_%s" % (n, n, j, "prune" if prune else "raw"),
                                  kind="adversarial", H=n, W=n, C=4, offsets=[6, 5], seed=200 + i,
                                  opts=o, prune=prune))
    for noise in (0.15, 0.6):
        for prune in (False, True):
            specs.append(dict(name="py_synth_64x128_n%02d_%s" % (int(noise * 100),
                                                                 "prune" if prune else "raw"),
                              kind="synth", H=64, W=128, C=9, offsets=[40, 10], seed=1000,
                              noise=noise, num_instances=4, opts=(0.0, 0.1, 0.0), prune=prune))
    if big:  # BASELINE.json configs[0]: 256x512 through the reference Python merger
        specs.append(dict(name="py_synth_256x512_cfg1", kind="synth", H=256, W=512, C=9,
                          offsets=[40, 10], seed=1000, noise=0.15, opts=(0.0, 0.1, 0.0), prune=True))
    return specs


def run_py_reference(cp, sp, C, offs, opts, prune: bool):
    sys.path.insert(0, "/root/reference")
    from utils.segmenter import ObjectSegmenter, SegmenterOptions  # reference import (validation only)
    with contextlib.redirect_stdout(io.StringIO()):
        seg = ObjectSegmenter(cp.astype(np.float64), sp.astype(np.float64), C, offs,
                              SegmenterOptions(*opts))
        if not prune:
            seg.prune = lambda *a, **k: None
        try:
            mask, classes = seg.run_segmentation()
        except NameError:
            return None, None, "NameError"
    return np.asarray(mask, np.int64), [int(c) for c in classes], ""


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also the 256x512 cases (minutes)")
    ap.add_argument("--only", default="", help="substring filter on fixture names")
    args = ap.parse_args()
    ck.build()
    if not ck.have_reference():
        raise SystemExit("oracle/_ref/libcsegment_ref.so missing: /root/reference not present?")
    index = {}
    for spec in cseg_specs(args.big):
        if args.only and args.only not in spec["name"]:
            continue
        cp, sp, offs = make_inputs(spec)
        t = time.time()
        ref = ck.run_reference(cp, sp, spec["C"], offs, *spec["opts"])
        dt = time.time() - t
        np.savez_compressed(os.path.join(HERE, spec["name"] + ".npz"),
                            spec=json.dumps(spec), sha256=digest(cp, sp),
                            mask=ref.mask.astype(np.int32),
                            object_class=np.asarray(ref.object_class, np.int32),
                            error="", ref_seconds=dt)
        index[spec["name"]] = dict(instances=len(ref.object_class), seconds=round(dt, 3))
        print("%-34s K=%-4d %.2fs" % (spec["name"], len(ref.object_class), dt), flush=True)
    for spec in py_specs(args.big):
        if args.only and args.only not in spec["name"]:
            continue
        cp, sp, offs = make_inputs(spec)
        t = time.time()
        mask, classes, err = run_py_reference(cp, sp, spec["C"], offs, spec["opts"], spec["prune"])
        dt = time.time() - t
        np.savez_compressed(os.path.join(HERE, spec["name"] + ".npz"),
                            spec=json.dumps(spec), sha256=digest(cp, sp),
                            mask=(mask if mask is not None else np.zeros((0, 0), np.int64)),
                            object_class=np.asarray(classes if classes is not None else [], np.int32),
                            error=err, ref_seconds=dt)
        index[spec["name"]] = dict(instances=(len(classes) if classes is not None else -1),
                                   error=err, seconds=round(dt, 3))
        print("%-34s K=%-4s %s %.2fs" % (spec["name"], len(classes) if classes is not None else "-",
                                         err, dt), flush=True)
    print(json.dumps(index)[:200])


if __name__ == "__main__":
    main()
