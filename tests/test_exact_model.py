"""tests/tools/exact_model.cpp -- the CPU model of the exact engine's semantics (the reference's lazy greedy with ties
going to the lowest record id) and of its tie-conflict criterion -- against vectors taken from the reference
(utils/csegment/segment.cc:539-727).  CPU only.  The GPU suite then compares the engine with this model
(tests/test_gpu_exact.py::test_tie_conflict_verdict_equals_the_cpu_models)."""
import os
import sys

import numpy as np
import pytest

import golden_util as gu
from mergenet_amd import labels

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
import exact_model  # noqa: E402


def _run(name):
    g = gu.load(name)
    sdb, omf, bias = g["spec"]["opts"]
    assert sdb == 0.0
    part, ocls, st = exact_model.run(g["class_probs"], g["sameness_probs"], g["offsets"], omf, bias)
    # the reference's mask: label 0 for class-0 objects, one label per other object (segment.cc:491-517)
    mask = np.where(ocls == 0, 0, part + 1).astype(np.int64)
    _, inv = np.unique(mask, return_inverse=True)
    mask = inv.reshape(mask.shape)
    if (ocls == 0).sum() == 0:
        mask = mask + 1
    classes = [int(ocls.reshape(-1)[np.argmax(mask.reshape(-1) == k)]) for k in range(1, int(mask.max()) + 1)]
    return g, mask, classes, st


@pytest.mark.parametrize("name", ["cseg_adv_32x32_o0", "cseg_adv_48x48_o0", "cseg_adv_48x48_o1", "cseg_adv_64x64_o1",
                                  "cseg_blur_64x128_r2_s8001"])
def test_no_tie_conflict_means_the_references_partition(name):
    """Vectors with no tied pop, or with tied pops that the criterion calls harmless (cseg_blur_64x128_r2_s8001 has
    86): the model -- whose order among equals is NOT the reference's -- must reproduce the reference's result."""
    g, mask, classes, st = _run(name)
    assert st["tied_conflicts"] == 0, st
    assert labels.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


def test_a_tie_decided_vector_is_reported_as_conflicting():
    """cseg_blur4_128x256_s5100: the lowest-id rule ends elsewhere than the reference (115 199 pops against the
    oracle's 115 297) -- and the criterion must have said so."""
    g, mask, classes, st = _run("cseg_blur4_128x256_s5100")
    assert st["steps"] == 115199 and st["tied_steps"] > 0 and st["tied_conflicts"] > 0, st
    assert not labels.masks_equivalent(mask, classes, g["mask"], g["object_class"])


@pytest.mark.parametrize("window", [3, 16, 64])
@pytest.mark.parametrize("name", ["cseg_adv_32x32_o0", "cseg_adv_48x48_o1", "cseg_synth_32x64_n60", "cseg_blur_64x128_r2",
                                  "cseg_checker_96x128_b015"])
def test_prefix_executor_reproduces_the_sequential_order(name, window, monkeypatch):
    """The side-by-side executor sketched in DESIGN.md section 9 (XM_EXECUTOR=K in the model): per round the K best
    queue entries are planned from the state at the START of the round, and the longest prefix is committed in which
    no committed pop produced a queue key ahead of the next entry and no pop writes an object that an earlier one
    read or wrote (or reads one it wrote).  Claim: the state after every round is the sequential one -- same
    partition, same classes, same number of pops and merges as the sequential model."""
    g = gu.load(name)
    sdb, omf, bias = g["spec"]["opts"]
    monkeypatch.delenv("XM_EXECUTOR", raising=False)
    p0, c0, s0 = exact_model.run(g["class_probs"], g["sameness_probs"], g["offsets"], omf, bias, track=False)
    monkeypatch.setenv("XM_EXECUTOR", str(window))
    p1, c1, s1 = exact_model.run(g["class_probs"], g["sameness_probs"], g["offsets"], omf, bias, track=False)
    assert np.array_equal(p0, p1) and np.array_equal(c0, c1)
    assert (s0["steps"], s0["merges"]) == (s1["steps"], s1["merges"])


@pytest.mark.parametrize("name", ["cseg_adv_48x48_o0", "cseg_synth_32x64_n60", "cseg_blur_64x128_r2"])
def test_sorted_front_hands_out_the_queues_best_entries(name, monkeypatch):
    """The queue the executor would use on the GPU (XM_FRONT=M): a sorted front of the M best entries with the invariant
    "every entry outside is behind the front's last", rebuilt when fewer than K are left.  Every round's window must
    be the K best entries of the whole queue (checked inside the model), and the result the sequential one."""
    g = gu.load(name)
    sdb, omf, bias = g["spec"]["opts"]
    monkeypatch.delenv("XM_EXECUTOR", raising=False)
    monkeypatch.delenv("XM_FRONT", raising=False)
    p0, c0, s0 = exact_model.run(g["class_probs"], g["sameness_probs"], g["offsets"], omf, bias, track=False)
    monkeypatch.setenv("XM_EXECUTOR", "16")
    monkeypatch.setenv("XM_FRONT", "48")
    p1, c1, s1 = exact_model.run(g["class_probs"], g["sameness_probs"], g["offsets"], omf, bias, track=False)
    assert s1["front_errors"] == 0 and s1["front_rebuilds"] > 0
    assert np.array_equal(p0, p1) and (s0["steps"], s0["merges"]) == (s1["steps"], s1["merges"])
