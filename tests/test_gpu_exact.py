"""The exact engine (MN_MODE_EXACT: the reference's sequential order at any image size) on the GPU box.

Reference: RunSegmentation + Merge, utils/csegment/segment.cc:539-573, 602-727, with the float32
arithmetic of segment.cc:5-46, 107-150.  Vectors: tests/golden (outputs of the reference's compiled
segment.cc); phase A against the oracle's restatement, BIT FOR BIT (integer/bit work: no tolerance).
"""
import os

import numpy as np
import pytest

import golden_util as gu
from mergenet_amd import segmenter as seg
from mergenet_amd import synth

pytestmark = pytest.mark.gpu


def _run(g, mode=seg.MN_MODE_AUTO, **kw):
    H, W, C = g["spec"]["H"], g["spec"]["W"], g["spec"]["C"]
    ctx = seg.HostContext(H, W, C, len(g["offsets"]))
    sdb, omf, bias = g["spec"]["opts"]
    o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf,
                            merge_logprob_bias=bias, mode=mode, clip_inputs=1, **kw)
    try:
        return ctx.segment(g["class_probs"], g["sameness_probs"], g["offsets"], o)
    finally:
        ctx.close()


# ---- phase A: glibc's logf and the reference's float32 operation order, bit for bit -----------------
@pytest.mark.parametrize("shape", [(256, 512, 9, (40, 10), 0.15, (0.0, 1.0, 0.03)),
                                   (1024, 2048, 9, (40, 10), 0.15, (0.0, 1.0, 0.03)),
                                   (400, 667, 81, (80, 16), 0.15, (0.0, 1.0, 0.03)),
                                   (96, 160, 5, (12, 6), 0.45, (0.0, 0.25, 0.0)),
                                   (128, 256, 9, (40, 10), 0.25, (0.4, 1.0, 0.03)),       # same_different_bias != 0:
                                   (96, 160, 5, (12, 6), 0.45, (-0.7, 0.5, 0.01))])        # glibc's expf restated too
def test_exact_phase_a_is_bit_identical_to_the_oracle(oracle, shape):
    """Arg-max class, per-record log-odds (logf(p) - (float)log(1 - p), segment.cc:33-36) and initial
    priority (segment.cc:107-150) of EVERY record: identical bits (NaN where the edge leaves the image) --
    also with same_different_bias != 0, whose logit / sigmoid round trip goes through the C library's logf,
    log and expf in the reference (segment.cc:183-195; mn_ref_logf / mn_ref_expf here)."""
    import torch
    H, W, C, oa, noise, opts = shape
    offs = synth.generate_offsets(*oa)
    s = synth.synth_v1(H, W, C, offs, 1000, noise=noise, occlusion=(C == 81))
    merger = seg.Merger(H, W, C, len(offs))
    try:
        o = seg.default_options(same_different_bias=opts[0], object_merge_factor=opts[1],
                                merge_logprob_bias=opts[2], clip_inputs=1)
        cp = torch.from_numpy(np.ascontiguousarray(s.class_probs)).cuda()
        sp = torch.from_numpy(np.ascontiguousarray(s.sameness_probs)).cuda()
        cls, oml, prio = merger.exact_phase_a(cp, sp, offs, o)
        ref_cls, ref_oml, ref_prio = oracle.phase_a(s.class_probs, s.sameness_probs, C, offs, *opts)
        assert np.array_equal(cls.cpu().numpy().astype(np.int32), ref_cls)
        got_oml, got_prio = oml.cpu().numpy(), prio.cpu().numpy()
        assert np.array_equal(np.isnan(got_oml), np.isnan(ref_oml))
        ok = ~np.isnan(ref_oml)
        n_oml = int((got_oml[ok].view(np.uint32) != ref_oml[ok].view(np.uint32)).sum())
        n_prio = int((got_prio[ok].view(np.uint32) != ref_prio[ok].view(np.uint32)).sum())
        assert n_oml == 0 and n_prio == 0, (n_oml, n_prio, int(ok.sum()))
    finally:
        merger.close()


# ---- the order itself ----------------------------------------------------------------------------------
TIE_DECIDED_ALL = gu.names("cseg_blur4_")
TIE_DECIDED = [n for n in TIE_DECIDED_ALL if "128x256" in n]          # within MN_TIE_LIMIT_RECORDS
TIE_DECIDED_LARGE = [n for n in TIE_DECIDED_ALL if n not in TIE_DECIDED]   # above it (256x512: 1.25 M records)
ALL_CSEG = [n for n in gu.names("cseg_") if not any(t in n for t in ("1024x2048", "800x1333")) and n not in TIE_DECIDED_ALL]


@pytest.mark.parametrize("name", ALL_CSEG)
def test_exact_engine_equals_the_reference_on_every_vector(oracle, name):
    """Every reference vector up to 512x1024 -- adversarial uniform-random maps, the tie-dominated
    noise-0.6 fixtures, blurred (network-like) maps, the crowded 48-instance images, the bias-dominated
    checkerboard, C = 81 -- in MN_MODE_EXACT: the reference's partition, background set and classes."""
    g = gu.load(name)
    mask, classes, part, st = _run(g, seg.MN_MODE_EXACT)
    assert st["mode_used"] == seg.MN_MODE_EXACT and st["proof"] >= 1
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st
    assert st["merges"] == g["spec"]["H"] * g["spec"]["W"] - st["num_objects"]


# ---- the order among bit-equal priorities ----------------------------------------------------------------
@pytest.mark.parametrize("name", TIE_DECIDED)
def test_default_reproduces_the_reference_where_tie_order_decides(oracle, name):
    """A radius-4 blur carries the out-of-image sameness value 1.0 into the maps near the image border, where it
    is clipped to 0.99: the queue starts with runs of equal keys, and WHICH of them goes first decides instance
    borders.  The reference pops them as its std::priority_queue's heap and its unordered_map's iteration order
    produce (segment.h:237-242, 270-275; segment.cc:650-652).  MN_TIES_DEFAULT: the exact engine meets tied pops
    (stats.tied_steps), the image is small enough, so it is redone by the reference-order loop (mn_reforder.h:
    that heap and that hash map restated) -- and equals the reference's own output."""
    g = gu.load(name)
    mask, classes, part, st = _run(g, seg.MN_MODE_EXACT)
    assert st["tie_order_used"] == seg.MN_TIES_REFERENCE and st["tied_steps"] > 0 and st["proof"] == seg.MN_PROOF_SEQUENTIAL
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO)
    assert st["tie_order_used"] == seg.MN_TIES_REFERENCE
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


@pytest.mark.parametrize("name", TIE_DECIDED_LARGE)
def test_large_tie_decided_image_is_not_called_proven_and_require_proof_redoes_it(oracle, name):
    """A tie-decided vector ABOVE MN_TIE_LIMIT_RECORDS (radius-4 blur at 256x512, 1.25 M records; taken from the
    reference by tests/golden/make_golden.py).  Default options keep the exact engine's answer there -- which is
    NOT the reference's on this input (next test) -- and must not call it proven: proof == 3, never 2.
    require_proof = 1 does not accept that: the image is redone in the reference's order among equals and equals
    the reference (segment.h:270-275, segment.cc:553-565, 650-652)."""
    g = gu.load(name)
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO)
    assert st["mode_used"] == seg.MN_MODE_EXACT and st["tie_order_used"] == seg.MN_TIES_LOWEST_ID
    assert st["tied_steps"] > 0 and st["tied_conflicts"] > 0
    assert st["proof"] == seg.MN_PROOF_SEQUENTIAL_TIES
    from mergenet_amd import labels
    assert labels.agreement(mask, g["mask"]) >= 0.985 * mask.size, st
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO, require_proof=seg.MN_PROVE_ALWAYS)
    assert st["tie_order_used"] == seg.MN_TIES_REFERENCE and st["proof"] == seg.MN_PROOF_SEQUENTIAL
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


@pytest.mark.xfail(strict=True, reason="above MN_TIE_LIMIT_RECORDS the default keeps the exact engine's own tie rule "
                   "(lowest record id), which differs from the reference on this input; the result says proof == 3 "
                   "(test above) and require_proof = 1 / tie_order = MN_TIES_REFERENCE gives the reference's")
@pytest.mark.parametrize("name", TIE_DECIDED_LARGE)
def test_default_mode_on_a_large_tie_decided_image(oracle, name):
    g = gu.load(name)
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO)
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


# ---- the default path at the sizes BASELINE.json names -------------------------------------------------
@pytest.mark.parametrize("name", ["cseg_synth_1024x2048_cfg2", "cseg_synth_1024x2048_s1001", "cseg_synth_800x1333_cfg5"])
def test_default_mode_at_the_baseline_sizes_equals_the_reference(oracle, name):
    """configs[1], configs[2] (another rank's image) and configs[4] through the library's DEFAULT options (AUTO, no
    require_proof): the speculative attempt cannot certify these images (second-phase merges), so they are
    redone by the exact engine -- the reference's partition, background set and classes, and a proof field
    that says what was and was not left to the engine's tie rule."""
    g = gu.load(name)
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO)
    assert st["mode_used"] == seg.MN_MODE_EXACT, st
    assert st["proof"] == gu.sequential_proof(st) and st["proof"] >= seg.MN_PROOF_SEQUENTIAL
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st
    assert st["merges"] == g["spec"]["H"] * g["spec"]["W"] - st["num_objects"]


# ---- ... and on maps whose certainty fades at the instance boundaries (blurred), at those sizes -----------------
def test_default_mode_on_a_blurred_map_of_512x1024_equals_the_reference(oracle):
    """The size the reference's own caller uses (segment.py:93), a blurred map (2.07 M pops, a quarter of them tied):
    default options -> the exact engine, the reference's result, and a proof field that follows the tie verdict."""
    g = gu.load("cseg_blur_512x1024_r2_s4243")
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO)
    assert st["mode_used"] == seg.MN_MODE_EXACT
    assert st["proof"] == gu.sequential_proof(st) and st["proof"] >= seg.MN_PROOF_SEQUENTIAL
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


_blur_1024 = {}


def _blurred_1024x2048_default():
    if not _blur_1024:
        g = gu.load("cseg_blur_1024x2048_r2_s4242")
        _blur_1024["g"] = g
        _blur_1024["out"] = _run(g, seg.MN_MODE_AUTO)
    return _blur_1024["g"], _blur_1024["out"]


def test_a_tie_decided_map_at_1024x2048_is_reported_as_such_and_stays_close():
    """cseg_blur_1024x2048_r2_s4242 (the reference: 561 s): half of its 8.1 M pops are tied and their order decides
    468 pixels.  Above MN_TIE_LIMIT_RECORDS the default keeps the exact engine's own rule, so the result must say
    proof == 3 (never 2), and it stays within 0.1 % of the reference's pixels with the same number of instances;
    require_proof = 1 / MN_TIES_REFERENCE would redo it in the reference's order (ten minutes at this size:
    profiles/r04_reforder_1024x2048_cfg2.log -- not part of the suite)."""
    from mergenet_amd import labels
    g, (mask, classes, part, st) = _blurred_1024x2048_default()
    assert st["mode_used"] == seg.MN_MODE_EXACT and st["tie_order_used"] == seg.MN_TIES_LOWEST_ID
    assert st["tied_steps"] > 0 and st["tied_conflicts"] > 0 and st["proof"] == seg.MN_PROOF_SEQUENTIAL_TIES
    assert len(classes) == len(g["object_class"])
    assert labels.agreement(mask, g["mask"]) >= 0.999 * mask.size, st


@pytest.mark.xfail(strict=True, reason="a tie-decided input above MN_TIE_LIMIT_RECORDS: the default keeps the exact "
                   "engine's own order among bit-equal priorities, which ends 468 pixels away from the reference here; "
                   "the result says proof == 3 (test above)")
def test_default_mode_on_the_tie_decided_map_at_1024x2048(oracle):
    g, (mask, classes, part, st) = _blurred_1024x2048_default()
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


@pytest.mark.xfail(strict=True, reason="the exact engine's OWN tie rule (lowest record id first: MN_TIES_LOWEST_ID, and "
                   "what images above MN_TIE_LIMIT_RECORDS get) differs from the reference on these inputs; the CPU model "
                   "of the reference's semantics with that rule gives the engine's event count exactly (DESIGN.md section "
                   "5): the difference is the tie rule and nothing else")
@pytest.mark.parametrize("name", TIE_DECIDED)
def test_lowest_id_tie_rule_on_inputs_decided_by_tie_order(oracle, name):
    g = gu.load(name)
    mask, classes, part, st = _run(g, seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_LOWEST_ID)
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


@pytest.mark.parametrize("name", TIE_DECIDED)
def test_lowest_id_tie_rule_stays_close_where_tie_order_decides(oracle, name):
    """... and a floor under it: the partitions agree on at least 98.5 % of the pixels (measured 98.9-99.99 %
    on eight such images, tests/tools/gpu_exact_campaign.py) and differ by at most one instance."""
    from mergenet_amd import labels
    g = gu.load(name)
    mask, classes, part, st = _run(g, seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_LOWEST_ID)
    assert st["tie_order_used"] == seg.MN_TIES_LOWEST_ID
    assert abs(len(classes) - len(g["object_class"])) <= 1
    assert labels.agreement(mask, g["mask"]) >= 0.985 * mask.size, st


@pytest.mark.parametrize("name", ["cseg_adv_48x48_o0", "cseg_synth_32x64_n60", "cseg_checker_96x128_b015",
                                  "cseg_blur_64x128_r2", "cseg_synth_48x80_c81", "cseg_adv_16x16_o2"])
def test_reference_order_loop_pops_what_the_reference_pops(oracle, name):
    """MN_TIES_REFERENCE: the same partition AND the same number of pops as the oracle's run of the reference's
    loop on std::priority_queue / std::unordered_map -- stale queue entries included, so the queue itself went
    through the same states (with other sameness biases and 81 classes among the cases)."""
    g = gu.load(name)
    ref = oracle.run_csegment(g["class_probs"], g["sameness_probs"], g["spec"]["C"], g["offsets"], *g["spec"]["opts"])
    mask, classes, part, st = _run(g, seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_REFERENCE)
    assert st["tie_order_used"] == seg.MN_TIES_REFERENCE
    assert oracle.same_partition(part, ref.partition), name
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st
    assert st["finisher_steps"] == ref.stats["n_pops"], (st["finisher_steps"], ref.stats["n_pops"])
    assert st["merges"] == ref.stats["n_merges"]
    assert abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob)


def test_reference_order_loop_across_launches_and_with_a_small_workspace(oracle, monkeypatch, capfd):
    """The loop's state lives in memory: a launch that has used up its pop budget is followed by another
    (MN_X_BUDGET); a run that fills the bucket arena or the queue is repeated with twice as much."""
    g = gu.load("cseg_synth_32x64_n35")
    monkeypatch.setenv("MN_X_BUDGET", "3000")
    monkeypatch.setenv("MN_RO_ARENA_PER_PIXEL", "8")
    monkeypatch.setenv("MN_RO_HEAP_PER_RECORD", "1")
    monkeypatch.setenv("MN_TRACE_EXACT", "1")
    mask, classes, part, st = _run(g, seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_REFERENCE)
    err = capfd.readouterr().err
    assert "status 3" in err and "status 1" in err          # (arena full, workspace doubled, ... done)
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


def test_tie_limit_keeps_large_images_with_the_exact_engine(oracle, monkeypatch):
    """MN_TIES_DEFAULT redoes an image with tied pops only up to MN_TIE_LIMIT_RECORDS initial records (the
    reference-order loop is sequential); above, the exact engine's answer stands and the statistics say so."""
    monkeypatch.setenv("MN_TIE_LIMIT", "1000")
    g = gu.load("cseg_synth_32x64_n35")
    mask, classes, part, st = _run(g, seg.MN_MODE_EXACT)
    assert st["tie_order_used"] == seg.MN_TIES_LOWEST_ID and st["tied_steps"] > 0
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


def test_exact_engine_blurred_256x512_within_ten_seconds(oracle):
    """The size at which round 2's exact mode gave up (its arg-max scanned every record per step): the
    reference needs 12-15 s here; the bar set for the engine was 10 s."""
    import time
    g = gu.load("cseg_blur_256x512_r2")
    t = time.time()
    mask, classes, part, st = _run(g, seg.MN_MODE_EXACT)
    dt = time.time() - t
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st
    assert dt < 10.0, dt
    assert st["proof"] == gu.sequential_proof(st)


def test_exact_engine_follows_the_oracles_event_sequence(oracle):
    """Not only the same partition: on inputs without bit-equal priorities (continuous random maps,
    blurred maps) the same NUMBER of live pops and merges as the oracle's run of the reference loop
    (segment.cc:542-566) -- the engine executes the same sequence of events -- and the same
    log-likelihood.  (On inputs full of ties -- synth-v1 clips a third of its values to 0.99 -- equal
    priorities are popped in another order and the count of re-scored pops differs by a few per cent
    while partition and merges stay the reference's: last case.)"""
    for name in ("cseg_blur_64x128_r2", "cseg_adv_48x48_o1", "cseg_adv_64x64_o2", "cseg_blur_64x128_r2_s8001",
                 "cseg_synth_128x256"):
        g = gu.load(name)
        ref = oracle.run_csegment(g["class_probs"], g["sameness_probs"], g["spec"]["C"], g["offsets"],
                                  *g["spec"]["opts"])
        mask, classes, part, st = _run(g, seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_LOWEST_ID)
        assert oracle.same_partition(part, ref.partition), name
        assert st["merges"] == ref.stats["n_merges"], name
        if "synth" not in name:
            assert st["finisher_steps"] == ref.stats["n_live_pops"], (name, st["finisher_steps"], ref.stats["n_live_pops"])
        assert abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob)


def test_exact_engine_relaunch_after_a_step_budget(oracle, monkeypatch):
    """The loop kernel comes back when its step budget is used up and is launched again (block maxima
    rebuilt from the leaves): the result must not depend on where it was cut."""
    g = gu.load("cseg_blur_64x128_r2")
    a = _run(g, seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_LOWEST_ID)
    monkeypatch.setenv("MN_X_BUDGET", "777")
    b = _run(g, seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_LOWEST_ID)
    assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2])
    assert a[3]["finisher_steps"] == b[3]["finisher_steps"]
    assert oracle.masks_equivalent(b[0], b[1], g["mask"], g["object_class"])
    # ... nor may the tie tracking (its nesting stack and counters travel through memory between launches): this
    # vector has 86 tied pops and no conflict, cut or not
    g = gu.load("cseg_blur_64x128_r2_s8001")
    c = _run(g, seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_LOWEST_ID)
    monkeypatch.delenv("MN_X_BUDGET")
    d = _run(g, seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_LOWEST_ID)
    for k in ("finisher_steps", "merges", "tied_steps", "tied_merges", "tied_conflicts", "proof"):
        assert c[3][k] == d[3][k], (k, c[3][k], d[3][k])
    assert c[3]["tied_steps"] > 0 and c[3]["tied_conflicts"] == 0 and c[3]["proof"] == seg.MN_PROOF_SEQUENTIAL
    assert np.array_equal(c[0], d[0])


def test_exact_engine_python_variant(oracle):
    """The Python variant's formulae (n1 * n2 denominator, bias inside, merge on >=: segmenter.py:179-193,
    470) through the same engine, against vectors from the reference's utils/segmenter.py."""
    for name in ("py_adv_24x24_o0_raw", "py_adv_32x32_o1_prune", "py_synth_64x128_n60_raw"):
        g = gu.load(name)
        s = seg.ObjectSegmenter(g["class_probs"], g["sameness_probs"], g["spec"]["C"], g["offsets"],
                                seg.SegmenterOptions(*g["spec"]["opts"]))
        thr = 200.0 if g["spec"]["prune"] else -np.inf
        if g["error"]:
            with pytest.raises(NameError):
                s.run_segmentation(prune_threshold=thr, mode=seg.MN_MODE_EXACT)
            continue
        mask, classes = s.run_segmentation(prune_threshold=thr, mode=seg.MN_MODE_EXACT)
        assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), (name, s.stats)


# ---- the default path hands back the reference's result -------------------------------------------------
@pytest.mark.parametrize("name", ["cseg_blur_64x128_r2", "cseg_blur_64x128_r2_s8001", "cseg_blur_256x512_r2",
                                  "cseg_checker_96x128_b015", "cseg_crowd48_256x512_s6408",
                                  "cseg_crowd48_256x512_s6400", "cseg_synth_32x64_n60", "cseg_synth_256x512"])
def test_auto_mode_equals_the_reference_on_order_dependent_inputs(oracle, name):
    """MN_MODE_AUTO with default options: a result the fast path cannot certify is redone in the
    sequential order, so the answer is the reference's and says why (proof 1 = certificate, 2 = the order
    itself was run with nothing left to a tie rule, 3 = the order was run and tied choices conflicted: above
    MN_TIE_LIMIT_RECORDS the engine's own rule among equals stands).  These are the vectors the fast path alone gets wrong (strict xfails of round 2)."""
    g = gu.load(name)
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO)
    assert st["proof"] == (seg.MN_PROOF_CERTIFICATE if st["certified"] else gu.sequential_proof(st)), st
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


def test_stale_record_witness_through_auto(oracle):
    """The witness of the fast path's second-phase deviation (fresh instead of stale priorities,
    test_components_mode_known_deviation_stale_record): AUTO returns the reference's result."""
    offs = [(1, 0), (-1, 5)]
    s = synth.synth_v1(47, 72, 9, offs, 9004, noise=0.1)
    ref = oracle.run_csegment(s.class_probs, s.sameness_probs, 9, offs, 0.0, 1.0, 0.03)
    ctx = seg.HostContext(47, 72, 9, 2)
    try:
        o = seg.default_options(merge_logprob_bias=0.03, clip_inputs=1, exact_limit=1)   # (force the large-image route)
        mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
    finally:
        ctx.close()
    assert st["proof"] >= 1
    assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), st


def test_unproven_fast_path_is_opt_in_and_says_so(oracle):
    """require_proof = -1: the speculative fast path's own answer comes back, with proof == 0 where it is
    not certified; explicit ROUNDS / COMPONENTS requests behave the same by default."""
    g = gu.load("cseg_blur_64x128_r2")
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO, require_proof=seg.MN_PROVE_NEVER, exact_limit=1)
    assert st["proof"] == seg.MN_PROOF_NONE and st["mode_used"] != seg.MN_MODE_EXACT
    mask, classes, part, st = _run(g, seg.MN_MODE_ROUNDS)
    assert st["proof"] == seg.MN_PROOF_NONE and st["mode_used"] == seg.MN_MODE_ROUNDS
    mask, classes, part, st = _run(g, seg.MN_MODE_ROUNDS, require_proof=seg.MN_PROVE_ALWAYS)
    assert st["proof"] == seg.MN_PROOF_SEQUENTIAL and st["mode_used"] == seg.MN_MODE_EXACT
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


def test_a_batch_in_one_launch_equals_the_single_calls(oracle):
    """mn_segment_exact_batch: several images, a workgroup each, in ONE launch of the loop (and relaunched on a
    small step budget while some images have already finished): the reference's mask and classes for each."""
    import torch
    gs = [gu.load(n) for n in ("cseg_crowd48_256x512_s6400", "cseg_blur_256x512_r2", "cseg_synth_256x512",
                               "cseg_crowd48_256x512_s6408")]
    g0 = gs[0]
    H, W, C = g0["spec"]["H"], g0["spec"]["W"], g0["spec"]["C"]
    for g in gs:
        assert (g["spec"]["H"], g["spec"]["W"], g["spec"]["C"], g["spec"]["opts"]) == (H, W, C, g0["spec"]["opts"])
        assert np.array_equal(np.asarray(g["offsets"]), np.asarray(g0["offsets"]))
    sdb, omf, bias = g0["spec"]["opts"]
    o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias,
                            mode=seg.MN_MODE_EXACT, clip_inputs=1)
    cps = [torch.from_numpy(np.ascontiguousarray(g["class_probs"], dtype=np.float32)).cuda() for g in gs]
    sps = [torch.from_numpy(np.ascontiguousarray(g["sameness_probs"], dtype=np.float32)).cuda() for g in gs]
    batch = seg.ExactBatch(H, W, C, len(g0["offsets"]), len(gs))
    old = os.environ.get("MN_X_BUDGET")
    try:
        for budget in (None, "40000"):
            if budget:
                os.environ["MN_X_BUDGET"] = budget
            res = batch.segment(cps, sps, g0["offsets"], o, want_partition=True)
            for g, (mask, table, part, st) in zip(gs, res):
                assert st["mode_used"] == seg.MN_MODE_EXACT and st["proof"] >= 1
                assert st["merges"] == H * W - st["num_objects"]
                assert oracle.masks_equivalent(mask.cpu().numpy(), seg._class_list(table.cpu().numpy()),
                                               g["mask"], g["object_class"]), st
        with pytest.raises(ValueError):
            batch.segment(cps + cps, sps + sps, g0["offsets"], o)
    finally:
        if old is None:
            os.environ.pop("MN_X_BUDGET", None)
        else:
            os.environ["MN_X_BUDGET"] = old
        batch.close()


def test_a_full_workspace_grows_and_the_run_is_repeated(oracle, monkeypatch, capfd):
    """The adjacency arena is sized for what images use (with slack); a run that fills it stops with a
    consistent state, the workspace doubles and the image is run again -- same result."""
    monkeypatch.setenv("MN_X_ARENA_EXTRA", "20")
    monkeypatch.setenv("MN_TRACE_EXACT", "1")
    g = gu.load("cseg_crowd48_256x512_s6400")
    mask, classes, part, st = _run(g, seg.MN_MODE_EXACT)
    assert "workspace grows" in capfd.readouterr().err
    assert st["mode_used"] == seg.MN_MODE_EXACT and st["proof"] >= 1
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


def test_pair_table_slots_stay_consistent_when_inserts_move_occupants(oracle, monkeypatch, capfd):
    """Advisor (round 3): a slow insert of a merge pass may move an occupant of a full bucket that a lane of the
    same pass placed a moment ago; that lane's record must already hold its slot, or the old slot comes back and
    a later delete frees another pair's entry.  MN_X_FORCE_RELOCATE makes every slow insert move an occupant
    when it can; MN_X_CHECK_SLOTS verifies at the end of the run that every live record and its table slot
    point at each other (and fails the call otherwise)."""
    import re
    monkeypatch.setenv("MN_X_FORCE_RELOCATE", "1")
    monkeypatch.setenv("MN_X_CHECK_SLOTS", "1")
    g = gu.load("cseg_crowd48_256x512_s6400")
    mask, classes, part, st = _run(g, seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_LOWEST_ID)
    m = re.search(r"pair-table check: (\d+) errors \(slow inserts (\d+)\)", capfd.readouterr().err)
    assert m and int(m.group(1)) == 0 and int(m.group(2)) > 1000, m
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


def test_paranoid_build_gives_the_same_masks_and_step_counts():
    """The exact engine's loop issues its global loads and stores from ONE wave without fences and relies on the
    memory pipeline keeping accesses to one address in program order (mn_kernels_exact.h).
    libmergenet_hip_paranoid.so (-DMN_X_PARANOID, built by the same Makefile) waits for every outstanding
    vector-memory operation after each pass and step instead: the two builds must give identical masks,
    partitions, class lists and step / merge / tie counts."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    paranoid = os.path.join(root, "mergenet_amd", "libmergenet_hip_paranoid.so")
    assert os.path.exists(paranoid), "make -C mergenet_amd/csrc builds it"
    names = ["cseg_blur_64x128_r2", "cseg_crowd48_256x512_s6400"]
    outs = []
    for lib in (None, paranoid):
        env = dict(os.environ)
        env.pop("MN_LIB", None)
        if lib:
            env["MN_LIB"] = lib
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "gpu_exact_digest.py")] + names,
                           capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert outs[0]["lib"] == "libmergenet_hip.so" and outs[1]["lib"] == "libmergenet_hip_paranoid.so"
    for n in names:
        assert outs[0][n] == outs[1][n], (n, outs[0][n], outs[1][n])


def test_tied_pops_are_counted():
    """stats.tied_steps: pops at which a second live record held the bit-equal stored priority.  Continuous
    (unclipped, network-like) values give none or a handful; the clipped generator's plateaus give thousands."""
    adv = _run(gu.load("cseg_adv_64x64_o0"), seg.MN_MODE_EXACT)[3]
    assert adv["tied_steps"] <= 4 and adv["tied_merges"] <= adv["tied_steps"]
    syn = _run(gu.load("cseg_synth_64x128_n60"), seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_LOWEST_ID)[3]
    assert syn["tied_steps"] > 1000 and 0 < syn["tied_merges"] <= syn["tied_steps"]
    assert syn["tie_order_used"] == seg.MN_TIES_LOWEST_ID
    fast = _run(gu.load("cseg_synth_64x128_n15"), seg.MN_MODE_COMPONENTS, require_proof=-1)[3]
    assert fast["tied_steps"] == 0 and fast["tied_merges"] == 0


@pytest.mark.parametrize("name", ["cseg_adv_48x48_o0", "cseg_adv_64x64_o1", "cseg_adv_64x64_o0", "cseg_blur_64x128_r2",
                                  "cseg_blur_64x128_r2_s8001", "cseg_synth_64x128_n15", "cseg_blur4_128x256_s5100"])
def test_tie_conflict_verdict_equals_the_cpu_models(name):
    """stats.tied_conflicts: do the choices among bit-equal priorities matter?  tests/tools/exact_model.cpp is an
    independent CPU implementation of the engine's semantics and of the criterion (nesting stack of the popped
    priorities, last-toucher stamps per object): same number of pops, merges and tied pops, the same yes / no --
    and proof == 2 exactly where every pop was forced or no tied choice conflicted (two of these vectors have
    tied pops that provably commute: cseg_adv_48x48_o0, cseg_blur_64x128_r2_s8001 with 86 of them)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
    import exact_model
    g = gu.load(name)
    sdb, omf, bias = g["spec"]["opts"]
    assert sdb == 0.0
    part, ocls, m = exact_model.run(g["class_probs"], g["sameness_probs"], g["offsets"], omf, bias)
    mask, classes, gpart, st = _run(g, seg.MN_MODE_EXACT, tie_order=seg.MN_TIES_LOWEST_ID)
    assert labels_same_partition(gpart, part)
    assert (st["finisher_steps"], st["merges"], st["tied_steps"]) == (m["steps"], m["merges"], m["tied_steps"]), (st, m)
    assert (st["tied_conflicts"] > 0) == (m["tied_conflicts"] > 0), (st, m)
    if st["certified"]:               # (a separable map: the certificate holds whatever the order)
        assert st["proof"] == seg.MN_PROOF_CERTIFICATE
    else:
        assert st["proof"] == gu.sequential_proof(st)
        assert (st["proof"] == seg.MN_PROOF_SEQUENTIAL) == (m["tied_steps"] == 0 or m["tied_conflicts"] == 0)


def labels_same_partition(a, b):
    from mergenet_amd import labels
    return labels.same_partition(np.asarray(a).reshape(-1), np.asarray(b).reshape(-1))


def test_a_batch_honours_the_tie_policy(oracle):
    """mn_segment_exact_batch: an image whose exact run met tied pops and is small enough is redone in the
    reference's order, as a single call would do -- the tie-decided vectors come out as the reference's."""
    import torch
    gs = [gu.load(n) for n in TIE_DECIDED]
    g0 = gs[0]
    H, W, C = g0["spec"]["H"], g0["spec"]["W"], g0["spec"]["C"]
    sdb, omf, bias = g0["spec"]["opts"]
    o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias,
                            mode=seg.MN_MODE_EXACT, clip_inputs=1)
    cps = [torch.from_numpy(np.ascontiguousarray(g["class_probs"], dtype=np.float32)).cuda() for g in gs]
    sps = [torch.from_numpy(np.ascontiguousarray(g["sameness_probs"], dtype=np.float32)).cuda() for g in gs]
    batch = seg.ExactBatch(H, W, C, len(g0["offsets"]), len(gs))
    try:
        res = batch.segment(cps, sps, g0["offsets"], o)
        for g, (mask, table, part, st) in zip(gs, res):
            assert st["tie_order_used"] == seg.MN_TIES_REFERENCE and st["tied_steps"] > 0
            assert oracle.masks_equivalent(mask.cpu().numpy(), seg._class_list(table.cpu().numpy()),
                                           g["mask"], g["object_class"]), st
    finally:
        batch.close()


def test_a_batch_redoes_large_tied_images_together(oracle):
    """Inside a batch the reference-order loop runs a workgroup per image in ONE launch, so the tie policy reaches
    further than in a single call (MN_TIE_LIMIT_BATCH_RECORDS: 256x512 at ten offsets): the tie-decided 256x512
    vector -- which a single default call leaves to the engine's own rule (proof 3, and NOT the reference's
    partition) -- comes out as the reference's, beside two other 256x512 images redone in the same launch."""
    import torch
    names = TIE_DECIDED_LARGE + ["cseg_blur_256x512_r2", "cseg_crowd48_256x512_s6408"]
    gs = [gu.load(n) for n in names]
    g0 = gs[0]
    H, W, C = g0["spec"]["H"], g0["spec"]["W"], g0["spec"]["C"]
    sdb, omf, bias = g0["spec"]["opts"]
    o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias,
                            mode=seg.MN_MODE_EXACT, clip_inputs=1)
    cps = [torch.from_numpy(np.ascontiguousarray(g["class_probs"], dtype=np.float32)).cuda() for g in gs]
    sps = [torch.from_numpy(np.ascontiguousarray(g["sameness_probs"], dtype=np.float32)).cuda() for g in gs]
    batch = seg.ExactBatch(H, W, C, len(g0["offsets"]), len(gs))
    try:
        res = batch.segment(cps, sps, g0["offsets"], o)
        for n, g, (mask, table, part, st) in zip(names, gs, res):
            assert st["tied_conflicts"] > 0 and st["tie_order_used"] == seg.MN_TIES_REFERENCE, (n, st)
            assert st["proof"] == seg.MN_PROOF_SEQUENTIAL
            assert oracle.masks_equivalent(mask.cpu().numpy(), seg._class_list(table.cpu().numpy()),
                                           g["mask"], g["object_class"]), (n, st)
    finally:
        batch.close()


def test_a_batch_with_more_images_than_compute_units():
    """More images than the chip has compute units: mn_segment_exact_batch then gives every context at most 6 K queue
    blocks (instead of 16 K), so that two workgroups share a unit's LDS -- the blocks are twice as long at this
    size (320x640: 2 M records), the results must not change: every image equals the single call of its inputs."""
    import torch
    H, W, C = 320, 640, 9
    offs = synth.generate_offsets(40, 10)
    ims = [synth.synth_v1(H, W, C, offs, 1000 + i) for i in range(2)]
    cps = [torch.from_numpy(im.class_probs).cuda() for im in ims]
    sps = [torch.from_numpy(im.sameness_probs).cuda() for im in ims]
    o = seg.default_options(merge_logprob_bias=0.03, mode=seg.MN_MODE_EXACT, clip_inputs=1, tie_order=seg.MN_TIES_LOWEST_ID)
    one = seg.Merger(H, W, C, len(offs))
    ref = [one.segment(cps[i], sps[i], offs, o, want_partition=True) for i in range(2)]
    one.close()
    n = torch.cuda.get_device_properties(0).multi_processor_count + 8
    batch = seg.ExactBatch(H, W, C, len(offs), n)
    try:
        res = batch.segment([cps[i % 2] for i in range(n)], [sps[i % 2] for i in range(n)], offs, o, want_partition=True)
        for i, (mask, table, part, st) in enumerate(res):
            r = ref[i % 2]
            assert torch.equal(mask, r[0]) and torch.equal(part, r[2]), i
            assert (st["finisher_steps"], st["merges"], st["tied_steps"]) == (r[3]["finisher_steps"], r[3]["merges"], r[3]["tied_steps"]), i
    finally:
        batch.close()


def test_a_batch_of_the_reference_order_loop_with_small_workspaces_and_budgets(oracle, monkeypatch, capfd):
    """run_reforder_batch: images whose bucket arena or queue fills up are repeated with twice as much while the
    others of the launch finish; a launch that has used up its pop budget is followed by another."""
    import torch
    monkeypatch.setenv("MN_X_BUDGET", "20000")
    monkeypatch.setenv("MN_RO_ARENA_PER_PIXEL", "8")
    monkeypatch.setenv("MN_RO_HEAP_PER_RECORD", "1")
    monkeypatch.setenv("MN_TRACE_EXACT", "1")
    names = ["cseg_synth_64x128_n15", "cseg_synth_64x128_n35", "cseg_synth_64x128_n60"]
    gs = [gu.load(n) for n in names]
    g0 = gs[0]
    H, W, C = g0["spec"]["H"], g0["spec"]["W"], g0["spec"]["C"]
    sdb, omf, bias = g0["spec"]["opts"]
    o = seg.default_options(same_different_bias=sdb, object_merge_factor=omf, merge_logprob_bias=bias,
                            mode=seg.MN_MODE_EXACT, clip_inputs=1, tie_order=seg.MN_TIES_REFERENCE)
    cps = [torch.from_numpy(np.ascontiguousarray(g["class_probs"], dtype=np.float32)).cuda() for g in gs]
    sps = [torch.from_numpy(np.ascontiguousarray(g["sameness_probs"], dtype=np.float32)).cuda() for g in gs]
    batch = seg.ExactBatch(H, W, C, len(g0["offsets"]), len(gs))
    try:
        res = batch.segment(cps, sps, g0["offsets"], o)
        err = capfd.readouterr().err
        assert "status 3" in err or "status 4" in err          # (some image ran out of workspace and was repeated)
        for n, g, (mask, table, part, st) in zip(names, gs, res):
            ref = oracle.run_csegment(g["class_probs"], g["sameness_probs"], C, g["offsets"], sdb, omf, bias)
            assert st["tie_order_used"] == seg.MN_TIES_REFERENCE
            assert st["proof"] in (seg.MN_PROOF_CERTIFICATE, seg.MN_PROOF_SEQUENTIAL)      # (separable maps: certified, too)
            assert st["finisher_steps"] == ref.stats["n_pops"], (n, st["finisher_steps"], ref.stats["n_pops"])
            assert oracle.masks_equivalent(mask.cpu().numpy(), seg._class_list(table.cpu().numpy()),
                                           g["mask"], g["object_class"]), (n, st)
    finally:
        batch.close()
