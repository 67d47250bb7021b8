"""Parity of the HIP merger with the CPU oracle / the reference's golden vectors (GPU box)."""
import numpy as np
import pytest

import golden_util as gu
from mergenet_amd import segmenter as seg
from mergenet_amd import synth, labels

pytestmark = pytest.mark.gpu

ORDER_DEPENDENT = tuple(gu.names("cseg_blur_") + gu.names("cseg_crowd48_") + gu.names("cseg_checker_") +
                        gu.names("cseg_blur4_"))          # (blur4: decided by the order among bit-equal priorities,
                                                          #  tests/test_gpu_exact.py)
CSEG = [n for n in gu.names("cseg_") if n not in ORDER_DEPENDENT]
TIE_DOMINATED = {"cseg_synth_32x64_n60"}
PY = gu.names("py_")


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


BIG = [n for n in CSEG if any(t in n for t in ("256x512", "512x1024", "1024x2048", "400x667", "800x1333"))]


@pytest.mark.parametrize("name", [n for n in CSEG if n not in BIG and n not in TIE_DOMINATED])
def test_golden_csegment(oracle, name):
    """Instance ids equal the REFERENCE's (its compiled segment.cc) up to label permutation."""
    g = gu.load(name)
    noisy = g["spec"]["kind"] == "adversarial" or g["spec"].get("noise", 0.15) > 0.35
    mask, classes, part, stats = _run(g, seg.MN_MODE_EXACT if noisy else seg.MN_MODE_AUTO)
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), stats
    if not noisy and stats["mode_used"] == seg.MN_MODE_ROUNDS:
        assert stats["certified"] == 1


@pytest.mark.parametrize("name", sorted(TIE_DOMINATED))
def test_golden_csegment_tie_dominated_exact_mode(oracle, name):
    """~40 % of the sameness values of this input are clipped to exactly 0.99 / 0.01, so thousands of
    records share one priority.  The exact engine gives equal priorities to the lowest record id (the
    reference's creation order) and reproduces the reference's result."""
    g = gu.load(name)
    mask, classes, part, stats = _run(g, seg.MN_MODE_EXACT)
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), stats


# ---- order-dependent inputs (round 2): vectors from the reference's own segment.cc ------------------
# What is PROVEN equal to the reference's sequential order is EXACT mode; the fast modes are
# approximations of that order on such inputs and say so (certified == 0).  The strict xfails
# below pin the known gaps: a fix would turn them into failures of the suite and be noticed.

@pytest.mark.parametrize("name", ["cseg_blur_64x128_r2", "cseg_blur_64x128_r2_s8001", "cseg_checker_96x128_b015"])
def test_order_dependent_goldens_exact_mode_equals_reference(oracle, name):
    """Maps whose certainty fades at the boundaries (values pass through 0.5) and a bias-dominated
    checkerboard: the sequential order on the GPU (MN_MODE_EXACT) gives the reference's result."""
    g = gu.load(name)
    mask, classes, part, stats = _run(g, seg.MN_MODE_EXACT)
    assert stats["mode_used"] == seg.MN_MODE_EXACT
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), stats


@pytest.mark.xfail(strict=True, reason="known gap of the OPT-IN fast path: on maps that are not sign-separable it falls to the "
                   "parallel rounds, which reproduce the reference's instance count but assign boundary "
                   "pixels (0.2-1 % of the image) differently from its sequential order; on the "
                   "bias-dominated checkerboard the second phase starts from fresh instead of stale "
                   "priorities (DESIGN.md section 5).  The result carries certified == 0.")
@pytest.mark.parametrize("name", ["cseg_blur_64x128_r2", "cseg_blur_64x128_r2_s8001", "cseg_blur_256x512_r2",
                                  "cseg_checker_96x128_b015", "cseg_crowd48_256x512_s6408"])
def test_order_dependent_goldens_speculative_path_known_gap(oracle, name):
    """(The default AUTO path redoes these in the sequential order and equals the reference:
    tests/test_gpu_exact.py.  This pins the opt-in fast path, require_proof = -1.)"""
    g = gu.load(name)
    mask, classes, part, stats = _run(g, seg.MN_MODE_AUTO, require_proof=seg.MN_PROVE_NEVER, exact_limit=1)
    assert stats["certified"] == 0           # whatever it returns, it must not claim a proof
    assert stats["proof"] == 0
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), stats


def test_crowded_48_instance_goldens(oracle):
    """256x512 images crowded with 48 overlapping instances (slivers of a few pixels): seed 6400 equals
    the reference in every mode.  (Seed 6408 is sign-separable but order-dependent in its second phase -- which
    small instances the background swallows: the speculative components attempt starts that phase from fresh
    priorities where the reference's records are stale, strict xfail above; AUTO redoes it in the exact
    engine, tests/test_gpu_exact.py.)"""
    g = gu.load("cseg_crowd48_256x512_s6400")
    for mode in (seg.MN_MODE_AUTO, seg.MN_MODE_ROUNDS):
        mask, classes, part, stats = _run(g, mode)
        assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), (mode, stats)


def test_crowded_48_seed_6408_rounds_equal_reference(oracle):
    """The explicit rounds (from the cores, no contraction of object clusters: round 2's contraction started the
    second phase from fresh priorities, lost this vector and was removed in round 4) equal the reference here."""
    g = gu.load("cseg_crowd48_256x512_s6408")
    mask, classes, part, stats = _run(g, seg.MN_MODE_ROUNDS)
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), stats


@pytest.mark.parametrize("name", ["cseg_synth_1024x2048_cfg2", "cseg_synth_1024x2048_s1003", "cseg_synth_512x1024_s1001",
                                  "cseg_synth_400x667_c81", "cseg_crowd48_256x512_s6400", "cseg_crowd48_256x512_s6408"])
def test_rounds_without_cluster_contraction_equal_reference(oracle, name):
    """The parallel rounds themselves (from the cores; order-free clusters NOT contracted -- the default
    since round 3 -- so that the rounds really run on these separable maps) equal the reference's result."""
    g = gu.load(name)
    mask, classes, part, stats = _run(g, seg.MN_MODE_ROUNDS)
    assert stats["rounds"] > 3
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), stats


@pytest.mark.parametrize("name", ["cseg_synth_512x1024_s1000", "cseg_synth_256x512"])
def test_rounds_from_single_pixels_equal_reference(oracle, name):
    """... and so do the rounds from single pixels (no cores: debug_flags bit 2 -- the form that also serves
    options the cores cannot, e.g. the Python variant with a bias)."""
    g = gu.load(name)
    mask, classes, part, stats = _run(g, seg.MN_MODE_ROUNDS, debug_flags=4)
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), stats


def test_golden_csegment_256x512_rounds(oracle):
    g = gu.load("cseg_synth_256x512")
    mask, classes, part, stats = _run(g, seg.MN_MODE_ROUNDS)
    assert stats["initial_records"] == 1254486
    assert stats["certified"] == 1
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), stats


@pytest.mark.parametrize("name", ["cseg_synth_64x128_n15", "cseg_synth_64x128_n35", "cseg_synth_128x256"])
def test_rounds_and_exact_agree_with_oracle_partition_and_loglik(oracle, name):
    g = gu.load(name)
    ref = oracle.run_csegment(g["class_probs"], g["sameness_probs"], g["spec"]["C"], g["offsets"],
                              *g["spec"]["opts"])
    for mode in (seg.MN_MODE_ROUNDS, seg.MN_MODE_EXACT):
        if mode == seg.MN_MODE_EXACT and g["spec"]["H"] > 64:
            continue
        mask, classes, part, stats = _run(g, mode)
        assert oracle.same_partition(part, ref.partition), (mode, stats)
        # tolerance stated by BASELINE.json: log-likelihood within 1e-5 (relative, float64
        # accumulation of float32 terms on both sides)
        assert abs(stats["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob)
        assert stats["merges"] == ref.stats["n_merges"]


@pytest.mark.parametrize("name", [n for n in BIG if "256x512" not in n])
def test_golden_csegment_large_images(oracle, name):
    """512x1024 (the size the reference's caller uses, segment.py:93) and 1024x2048
    (BASELINE.json configs[1..2]): the result equals the REFERENCE's own result -- its segment.cc
    needed up to 536 s and 6.5 GB per image -- up to label permutation."""
    g = gu.load(name)
    mask, classes, part, stats = _run(g, seg.MN_MODE_ROUNDS)
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), stats
    assert mask.min() == 0 and mask.max() == len(classes)
    assert stats["merges"] == g["spec"]["H"] * g["spec"]["W"] - stats["num_objects"]


def test_golden_csegment_1024x2048_full_size(oracle):
    """BASELINE.json configs[1]: size-independent properties on top of the golden comparison."""
    g = gu.load("cseg_synth_1024x2048_cfg2")
    mask, classes, part, stats = _run(g, seg.MN_MODE_ROUNDS)
    assert stats["initial_records"] == 20745558
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), stats
    # size-independent properties of the output: every pixel labelled, labels dense 0..K,
    # label-0 pixels are exactly the class-0 objects, merges = pixels - objects
    assert mask.min() == 0 and mask.max() == len(classes)
    assert stats["merges"] == 1024 * 2048 - stats["num_objects"]
    assert len(np.unique(part)) == stats["num_objects"]


def test_determinism_same_input_twice():
    g = gu.load("cseg_synth_128x256")
    a = _run(g, seg.MN_MODE_ROUNDS)
    b = _run(g, seg.MN_MODE_ROUNDS)
    assert np.array_equal(a[0], b[0]) and a[1] == b[1] and np.array_equal(a[2], b[2])
    assert a[3]["total_logprob"] == b[3]["total_logprob"]


def test_drop_in_run_segmentation_signature(oracle):
    """The reference binding's call (c_segment.pyx:30-86) served by c_run_segmentation."""
    g = gu.load("cseg_synth_32x64_n15")
    mask, classes = seg.run_segmentation(g["class_probs"], g["sameness_probs"], 9,
                                         list(g["offsets"]), *g["spec"]["opts"])
    assert mask.dtype == np.int32 and mask.shape == (32, 64)
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"])


def test_offsets_with_negation_rejected():
    cp = np.full((2, 8, 8), 0.5, np.float32)
    sp = np.full((2, 8, 8), 0.5, np.float32)
    ctx = seg.HostContext(8, 8, 2, 2)
    with pytest.raises(seg.MergeNetError) as e:
        ctx.segment(cp, sp, [(0, 1), (0, -1)])
    assert e.value.status == -2
    ctx.close()


@pytest.mark.parametrize("name", [n for n in PY if "256x512" not in n])
def test_golden_pysegmenter(oracle, name):
    """ObjectSegmenter look-alike against vectors from the reference's utils/segmenter.py."""
    g = gu.load(name)
    s = seg.ObjectSegmenter(g["class_probs"], g["sameness_probs"], g["spec"]["C"], g["offsets"],
                            seg.SegmenterOptions(*g["spec"]["opts"]))
    thr = 200.0 if g["spec"]["prune"] else -np.inf
    noisy = g["spec"]["kind"] == "adversarial" or g["spec"].get("noise", 0.15) > 0.35
    mode = seg.MN_MODE_EXACT if noisy else seg.MN_MODE_AUTO
    if g["error"]:
        with pytest.raises(NameError):
            s.run_segmentation(prune_threshold=thr, mode=mode)
        return
    mask, classes = s.run_segmentation(prune_threshold=thr, mode=mode)
    assert mask.dtype == np.int64
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), s.stats


def test_config1_pysegmenter_256x512_vs_reference_python(oracle):
    """BASELINE.json configs[0]: the 256x512 map the reference's utils/segmenter.py needs 169 s
    for (options 0, 1/O, 0, prune 200) -- same instances from the GPU ObjectSegmenter."""
    g = gu.load("py_synth_256x512_cfg1")
    s = seg.ObjectSegmenter(g["class_probs"], g["sameness_probs"], 9, g["offsets"],
                            seg.SegmenterOptions(*g["spec"]["opts"]))
    mask, classes = s.run_segmentation()
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), s.stats
    assert s.stats["certified"] == 1


@pytest.mark.parametrize("shape", [(5, 7, 2, 2), (1, 9, 3, 2), (9, 1, 3, 2), (13, 67, 5, 3), (31, 130, 3, 6)])
def test_ragged_shapes_match_oracle(oracle, shape):
    """Sizes that are not multiples of the tile/vector widths (N % 4 != 0, W < 64, single row or
    column, offsets longer than the image) against the CPU oracle, both modes."""
    H, W, C, O = shape
    offs = synth.generate_offsets(max(2, min(H, W, 6)), max(2, O))[:O]
    offs = [o for o in offs if o != (0, 0)]
    s = synth.adversarial(H, W, C, offs, 4242 + H * W)
    ref = oracle.run_csegment(s.class_probs, s.sameness_probs, C, offs, 0.0, 1.0, 0.03)
    ctx = seg.HostContext(H, W, C, len(offs))
    try:
        o = seg.default_options(mode=seg.MN_MODE_EXACT, clip_inputs=1)
        mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
        assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), st
        assert abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob)
        o = seg.default_options(mode=seg.MN_MODE_ROUNDS, clip_inputs=1)
        mask2, classes2, part2, st2 = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
        # ROUNDS on adversarial data is not expected to reproduce the sequential order; it must
        # still be a valid segmentation: dense labels, consistent counts, deterministic
        assert mask2.min() >= 0 and mask2.max() == len(classes2)
        assert st2["merges"] == H * W - st2["num_objects"]
    finally:
        ctx.close()


def test_all_background_and_empty_class_list(oracle):
    g = gu.load("cseg_closed_all_background")
    mask, classes, part, stats = _run(g)
    assert classes == [] and not mask.any()


def test_same_different_bias_matches_oracle(oracle):
    """same_different_bias != 0 is applied on load (the reference rewrites adj_pred in place,
    segment.cc:183-195); the caller's array must stay untouched."""
    g = gu.load("cseg_adv_24x24_o2")
    before = g["sameness_probs"].copy()
    mask, classes, part, stats = _run(g, seg.MN_MODE_EXACT)
    assert np.array_equal(before, g["sameness_probs"])
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), stats


def test_config4_network_to_merger_device_resident(oracle):
    """BASELINE.json configs[3]: PSPNet-ResNet50-shaped forward in PyTorch-ROCm feeding the HIP
    merger without leaving the GPU; the merger's output on the network's maps equals the oracle's
    (random weights give near-0.5, order-dependent maps: EXACT mode)."""
    import os, sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from pspnet_pipeline import PSPNetResNet50, segment_image
    H, W, C = 40, 56, 5
    offs = synth.generate_offsets(8, 6)
    torch.manual_seed(3)
    model = PSPNetResNet50(C, len(offs), width=16).cuda().eval()
    img = torch.rand(1, 3, H, W, device="cuda")
    merger = seg.Merger(H, W, C, len(offs))
    opts = seg.default_options(clip_inputs=1, mode=seg.MN_MODE_EXACT)
    mask, table, _, st = segment_image(model, img, offs, merger, opts)
    with torch.no_grad():
        probs = torch.sigmoid(model(img)[0]).float().cpu().numpy()
    ref = oracle.run_csegment(probs[:C], probs[C:], C, offs, 0.0, 1.0, 0.03)
    got = [int(c) for c in table.cpu().numpy()[: st["num_instances"]]]
    assert oracle.masks_equivalent(mask.cpu().numpy(), got, ref.mask, ref.object_class), st
    merger.close()


@pytest.mark.parametrize("noise", [0.15, 0.25, 0.35, 0.45])
def test_rounds_mode_equals_oracle_on_many_seeds(oracle, noise):
    """ROUNDS mode against the CPU oracle run on the box: 8 seeds per noise level at 128x256
    (4-5 instances each, C=9, O=10, Cityscapes options); every partition, class list and
    log-likelihood must agree.  Noise <= 0.35 keeps intra-instance edges above p = 0.5 (the
    certified regime); 0.45 is beyond it (tests/tools/gpu_noise_study.py: 8/8 equal up to noise 0.5,
    7/8 at 0.6 where the eighth has a HIGHER likelihood than the reference's result)."""
    offs = synth.generate_offsets(40, 10)
    ctx = seg.HostContext(128, 256, 9, len(offs))
    try:
        for seed in range(2000, 2008):
            s = synth.synth_v1(128, 256, 9, offs, seed, noise=noise)
            ref = oracle.run_csegment(s.class_probs, s.sameness_probs, 9, offs, 0.0, 1.0, 0.03)
            o = seg.default_options(mode=seg.MN_MODE_ROUNDS, clip_inputs=1)
            mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
            assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), (seed, st)
            assert oracle.same_partition(part, ref.partition), (seed, st)
            assert abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob)
    finally:
        ctx.close()


@pytest.mark.parametrize("opts", [(0.5, 1.0, 0.03), (-0.3, 0.5, 0.0), (0.0, 2.0, 0.1)])
def test_rounds_mode_with_other_options_equals_oracle(oracle, opts):
    """same_different_bias != 0 (applied on load), other merge factors and biases, ROUNDS mode."""
    offs = synth.generate_offsets(40, 10)
    s = synth.synth_v1(96, 160, 9, offs, 3100, noise=0.15, num_instances=4)
    ref = oracle.run_csegment(s.class_probs, s.sameness_probs, 9, offs, *opts)
    ctx = seg.HostContext(96, 160, 9, len(offs))
    try:
        o = seg.default_options(same_different_bias=opts[0], object_merge_factor=opts[1],
                                merge_logprob_bias=opts[2], mode=seg.MN_MODE_ROUNDS, clip_inputs=1)
        mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
        assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), st
        assert abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob)
    finally:
        ctx.close()


def test_one_context_serves_different_image_sizes(oracle):
    """A context created for the largest image is reused for smaller ones and other offset sets."""
    ctx = seg.HostContext(128, 256, 9, 10)
    try:
        for (H, W, C, oa, seed) in [(128, 256, 9, (40, 10), 1000), (32, 64, 9, (40, 10), 1000),
                                    (24, 40, 3, (10, 6), 5), (128, 256, 9, (40, 10), 1000)]:
            offs = synth.generate_offsets(*oa)
            s = synth.synth_v1(H, W, C, offs, seed, num_instances=4 if H < 100 else None)
            ref = oracle.run_csegment(s.class_probs, s.sameness_probs, C, offs, 0.0, 1.0, 0.03)
            mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs,
                                                  seg.default_options(clip_inputs=1))
            assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), (H, W, st)
    finally:
        ctx.close()
    big = seg.HostContext(16, 16, 3, 3)
    with pytest.raises(seg.MergeNetError) as e:
        big.segment(np.full((3, 32, 32), 0.5, np.float32), np.full((3, 32, 32), 0.5, np.float32),
                    synth.generate_offsets(4, 3))
    assert e.value.status == -4            # MN_ERR_CAPACITY
    big.close()


def test_fuzz_exact_mode_equals_oracle_on_random_small_inputs(oracle):
    """60 random tiny problems (continuous random maps: no ties): random sizes, class counts,
    offset lists and options; EXACT mode must reproduce the oracle's partition, classes and
    log-likelihood on every one."""
    rng = np.random.default_rng(20261003)
    bad = []
    for it in range(60):
        H, W = int(rng.integers(2, 20)), int(rng.integers(2, 24))
        C = int(rng.integers(1, 7))
        O = int(rng.integers(1, 7))
        offs = []
        while len(offs) < O:
            o = (int(rng.integers(-5, 6)), int(rng.integers(-5, 6)))
            if o == (0, 0) or o in offs or (-o[0], -o[1]) in offs:
                continue
            offs.append(o)
        cp = rng.uniform(0.02, 0.98, (C, H, W)).astype(np.float32)
        sp = rng.uniform(0.02, 0.98, (O, H, W)).astype(np.float32)
        opts = (float(rng.choice([0.0, 0.0, 0.4, -0.3])), float(rng.choice([1.0, 0.25, 0.5, 2.0])),
                float(rng.choice([0.0, 0.03, 0.1, -0.05])))
        ref = oracle.run_csegment(cp, sp, C, offs, *opts)
        ctx = seg.HostContext(H, W, C, O)
        try:
            o = seg.default_options(same_different_bias=opts[0], object_merge_factor=opts[1],
                                    merge_logprob_bias=opts[2], mode=seg.MN_MODE_EXACT)
            mask, classes, part, st = ctx.segment(cp, sp, offs, o)
        finally:
            ctx.close()
        ok = oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class) and \
            oracle.same_partition(part, ref.partition) and \
            abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob)
        if not ok:
            bad.append((it, H, W, C, offs, opts))
    assert not bad, bad


def test_abi_entry_regrows_its_cached_context(oracle):
    """c_run_segmentation keeps one context per thread and re-creates it for larger images."""
    for name in ("cseg_closed_two_halves", "cseg_synth_32x64_n15", "cseg_adv_16x16_o0",
                 "cseg_synth_64x128_n15", "cseg_closed_single_instance"):
        g = gu.load(name)
        mask, classes = seg.run_segmentation(np.ascontiguousarray(g["class_probs"]),
                                             np.ascontiguousarray(g["sameness_probs"]),
                                             g["spec"]["C"], list(g["offsets"]), *g["spec"]["opts"])
        assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), name


# ---- components mode (connected components over the positive edges, then the finisher) ----------

def _segment(cp, sp, offs, mode, opts=(0.0, 1.0, 0.03), variant=seg.MN_VARIANT_CSEGMENT):
    ctx = seg.HostContext(cp.shape[1], cp.shape[2], cp.shape[0], len(offs))
    try:
        o = seg.default_options(same_different_bias=opts[0], object_merge_factor=opts[1],
                                merge_logprob_bias=opts[2], mode=mode, clip_inputs=1, variant=variant)
        return ctx.segment(cp, sp, offs, o)
    finally:
        ctx.close()


def _components_vs_oracle(oracle, H, W, C, seed, noise, opts=(0.0, 1.0, 0.03), python_variant=False):
    offs = synth.generate_offsets(12, 6)
    s = synth.synth_v1(H, W, C, offs, seed, noise=noise)
    if python_variant:
        ref = oracle.run_pysegmenter(s.class_probs, s.sameness_probs, C, offs, *opts)
        variant = seg.MN_VARIANT_PYSEGMENTER
    else:
        ref = oracle.run_csegment(s.class_probs, s.sameness_probs, C, offs, *opts)
        variant = seg.MN_VARIANT_CSEGMENT
    mask, classes, part, st = _segment(s.class_probs, s.sameness_probs, offs, seg.MN_MODE_COMPONENTS,
                                       opts, variant)
    return ref, mask, classes, part, st


@pytest.mark.parametrize("name", [n for n in BIG if "256x512" not in n])
def test_components_mode_equals_reference_on_large_goldens(oracle, name):
    """Every large golden (reference segment.cc output) in components mode, forced."""
    g = gu.load(name)
    mask, classes, part, st = _run(g, seg.MN_MODE_COMPONENTS)
    assert st["mode_used"] == seg.MN_MODE_COMPONENTS, "separable synthetic maps must not fall back"
    assert st["rounds"] == 0
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st
    assert st["merges"] == g["spec"]["H"] * g["spec"]["W"] - st["num_objects"]


def test_auto_mode_picks_components_for_large_images(oracle):
    g = gu.load("cseg_synth_512x1024_s1001")                 # certified: the fast path's answer stands
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO)
    assert st["mode_used"] == seg.MN_MODE_COMPONENTS and st["proof"] == seg.MN_PROOF_CERTIFICATE
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st
    g = gu.load("cseg_synth_512x1024_s1000")                 # second-phase merges: not certified
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO, require_proof=seg.MN_PROVE_NEVER)
    assert st["mode_used"] == seg.MN_MODE_COMPONENTS and st["proof"] == seg.MN_PROOF_NONE
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO)      # default: redone in the sequential order
    assert st["mode_used"] == seg.MN_MODE_EXACT and st["proof"] == gu.sequential_proof(st)
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


@pytest.mark.parametrize("shape", [(48, 96, 3), (61, 100, 4), (33, 67, 2), (40, 41, 5)])
def test_components_mode_ragged_widths_match_oracle(oracle, shape):
    """Widths that are / are not multiples of 4 and 64 (tile borders, 1-pixel-per-lane sweeps)."""
    H, W, C = shape
    ref, mask, classes, part, st = _components_vs_oracle(oracle, H, W, C, 77, 0.15)
    assert st["mode_used"] == seg.MN_MODE_COMPONENTS
    assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), st
    assert oracle.same_partition(part, ref.partition), st
    assert abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob)
    assert st["merges"] == ref.stats["n_merges"]


@pytest.mark.parametrize("shape", [(64, 1), (64, 2), (60, 3), (1, 64), (128, 1)])
def test_images_narrower_than_a_lane_group(oracle, shape):
    """W < 4 with N % 4 == 0 (advisor, round 3): a lane's four linear pixels would span up to four rows, which
    the sweep's straddling lane does not handle -- such shapes take one pixel per lane.  Whatever mode the image
    ends in, the result is the oracle's (vertical offsets only make sense here: (1,0), (-2,0), (3,0) / their
    horizontal twins)."""
    H, W = shape
    offs = [(1, 0), (-2, 0), (3, 0)] if W < 4 else [(0, 1), (0, -2), (0, 3)]
    s = synth.synth_v1(H, W, 3, offs, 31, noise=0.15, num_instances=2)
    ref = oracle.run_csegment(s.class_probs, s.sameness_probs, 3, offs, 0.0, 1.0, 0.03)
    for mode in (seg.MN_MODE_COMPONENTS, seg.MN_MODE_EXACT):
        mask, classes, part, st = _segment(s.class_probs, s.sameness_probs, offs, mode, (0.0, 1.0, 0.03),
                                           seg.MN_VARIANT_CSEGMENT)
        assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), (shape, mode, st)
        assert abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob)


def test_components_mode_falls_back_when_not_separable(oracle):
    """Noise 0.45 flips edge signs inside instances: the check must send the image to the rounds."""
    ref, mask, classes, part, st = _components_vs_oracle(oracle, 96, 160, 4, 5, 0.45)
    assert st["mode_used"] == seg.MN_MODE_ROUNDS
    assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), st


def test_components_mode_needs_nonnegative_bias_and_positive_factor(oracle):
    for opts in [(0.0, 1.0, -0.02), (0.0, -1.0, 0.03)]:
        ref, mask, classes, part, st = _components_vs_oracle(oracle, 64, 96, 3, 9, 0.15, opts=opts)
        assert st["mode_used"] == seg.MN_MODE_ROUNDS
        assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), (opts, st)


def test_components_mode_python_variant_only_with_bias_zero(oracle):
    ref, mask, classes, part, st = _components_vs_oracle(oracle, 64, 96, 3, 11, 0.15, opts=(0.0, 1.0, 0.0),
                                                         python_variant=True)
    assert st["mode_used"] == seg.MN_MODE_COMPONENTS
    assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), st
    # with the Python variant's own bias the two kinds of record are not separated by the bias
    ref, mask, classes, part, st = _components_vs_oracle(oracle, 64, 96, 3, 11, 0.15, opts=(0.0, 1.0, 0.05),
                                                         python_variant=True)
    assert st["mode_used"] == seg.MN_MODE_ROUNDS
    assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), st


def _checkerboard(H, W, C, offs, cell_px, seed):
    rng = np.random.default_rng(seed)
    cell = (np.arange(H)[:, None] // cell_px) * ((W + cell_px - 1) // cell_px) + (np.arange(W)[None, :] // cell_px)
    cmap = rng.integers(0, C, size=cell.max() + 1)[cell]
    cp = np.full((C, H, W), 0.05, np.float32)
    for c in range(C):
        cp[c][cmap == c] = 0.9
    cp += rng.uniform(0, 0.01, size=cp.shape).astype(np.float32)
    sp = np.zeros((len(offs), H, W), np.float32)
    for k, (di, dj) in enumerate(offs):
        other = np.full((H, W), -1)
        r0, r1 = max(0, -di), min(H, H - di)
        c0, c1 = max(0, -dj), min(W, W - dj)
        other[r0:r1, c0:c1] = cell[r0 + di:r1 + di, c0 + dj:c1 + dj]
        sp[k] = np.where(other == cell, 0.9, 0.1) + rng.uniform(-0.05, 0.05, size=(H, W))
    return cp.astype(np.float32), sp.astype(np.float32)


def test_components_mode_many_small_components_overflowing_tables(oracle):
    """A checkerboard of 4x4 cells: ~250 components per 4096-pixel block overflow the 64-slot LDS
    table of the class sweep and crowd the 256-slot one of the edge sweep, so their direct
    global-atomic paths run; the result is still the oracle's."""
    offs = synth.generate_offsets(6, 4)
    cp, sp = _checkerboard(96, 128, 3, offs, 4, 3)
    ref = oracle.run_csegment(cp, sp, 3, offs, 0.0, 1.0, 0.03)
    mask, classes, part, st = _segment(cp, sp, offs, seg.MN_MODE_COMPONENTS)
    assert st["mode_used"] == seg.MN_MODE_COMPONENTS
    assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), st
    assert abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob)


def test_components_mode_values_inside_rounding_margin_fall_back(oracle):
    """An edge whose sameness value sits within the float32 rounding margin of 0.5 fails the
    separability check even though its sign is right (sep_hi / sep_lo in fill_params)."""
    offs = synth.generate_offsets(6, 4)
    s = synth.synth_v1(48, 64, 2, offs, 21, noise=0.1)
    sp = s.sameness_probs.copy()
    inside = np.argwhere(sp[1] > 0.6)
    r, c = inside[len(inside) // 2]
    sp[1, r, c] = np.float32(0.5) + np.float32(1e-7)        # positive log-odds of ~4e-7
    ref = oracle.run_csegment(s.class_probs, sp, 2, offs, 0.0, 1.0, 0.03)
    mask, classes, part, st = _segment(s.class_probs, sp, offs, seg.MN_MODE_COMPONENTS)
    assert st["mode_used"] == seg.MN_MODE_ROUNDS
    assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), st


def test_components_mode_record_table_overflow_falls_back(oracle):
    """Every edge negative: each pixel is its own component and the ~N*O records between them do
    not fit the small table of this mode; the bounded insert must report it (not spin) and the
    image is redone by the rounds -- where nothing merges, as in the reference."""
    H, W, C = 96, 128, 3
    offs = synth.generate_offsets(6, 4)
    rng = np.random.default_rng(8)
    cp = rng.uniform(0.1, 0.9, size=(C, H, W)).astype(np.float32)
    sp = rng.uniform(0.05, 0.3, size=(len(offs), H, W)).astype(np.float32)
    ref = oracle.run_csegment(cp, sp, C, offs, 0.0, 1.0, 0.03)
    mask, classes, part, st = _segment(cp, sp, offs, seg.MN_MODE_COMPONENTS)
    assert st["mode_used"] == seg.MN_MODE_ROUNDS
    assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), st
    assert oracle.same_partition(part, ref.partition), st


@pytest.mark.parametrize("name", ["cseg_synth_512x1024_s1000", "cseg_synth_512x1024_s1001",
                                  "cseg_synth_1024x2048_cfg2", "cseg_synth_1024x2048_s1002"])
def test_components_certificate_equals_the_sweep_over_the_final_partition(oracle, name):
    """Components mode takes certificate and log-likelihood from the contraction's own sweep plus
    what the finisher merged; the rounds recompute them by a sweep over the final partition.  Same
    partition => same violation counts, log-likelihood equal to float32-sum accuracy."""
    g = gu.load(name)
    mask_c, classes_c, part_c, st_c = _run(g, seg.MN_MODE_COMPONENTS)
    mask_r, classes_r, part_r, st_r = _run(g, seg.MN_MODE_ROUNDS)
    assert st_c["mode_used"] == seg.MN_MODE_COMPONENTS and st_r["mode_used"] == seg.MN_MODE_ROUNDS
    assert oracle.same_partition(part_c, part_r)
    assert st_c["cert_edge_violations"] == st_r["cert_edge_violations"]
    assert st_c["cert_class_violations"] == st_r["cert_class_violations"]
    assert st_c["cert_record_violations"] == st_r["cert_record_violations"]
    assert st_c["certified"] == st_r["certified"]
    assert abs(st_c["total_logprob"] - st_r["total_logprob"]) <= 1e-7 * abs(st_r["total_logprob"])


def test_fuzz_components_mode_on_random_shapes_offsets_and_options(oracle):
    """Random image sizes (tile and 4-pixel-lane borders everywhere), class counts, offset sets
    (with and without the unit offsets the tile stage uses) and options; components mode forced.
    Whatever path it ends on, the result must be the oracle's."""
    rng = np.random.default_rng(20261004)
    used = {seg.MN_MODE_COMPONENTS: 0, seg.MN_MODE_ROUNDS: 0}
    for trial in range(24):
        H = int(rng.integers(8, 72))
        W = int(rng.integers(8, 150))
        C = int(rng.integers(2, 13))
        offs = synth.generate_offsets(int(rng.integers(3, 14)), int(rng.integers(4, 8)))
        if trial % 3 == 2:                       # an offset list without (0,1) / (1,0)
            offs = [o for o in offs if abs(o[0]) + abs(o[1]) > 1]
        noise = float(rng.choice([0.1, 0.2, 0.3]))
        opts = [(0.0, 1.0, 0.03), (0.3, 1.0, 0.03), (-0.2, 0.7, 0.0), (0.0, 2.0, 0.1)][trial % 4]
        s = synth.synth_v1(H, W, C, offs, 9000 + trial, noise=noise)
        ref = oracle.run_csegment(s.class_probs, s.sameness_probs, C, offs, *opts)
        mask, classes, part, st = _segment(s.class_probs, s.sameness_probs, offs, seg.MN_MODE_COMPONENTS, opts)
        used[st["mode_used"]] += 1
        ctx = (trial, H, W, C, offs, noise, opts, st)
        assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), ctx
        assert oracle.same_partition(part, ref.partition), ctx
        assert abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob), ctx
    assert used[seg.MN_MODE_COMPONENTS] >= 12, used     # the fast path must be what is mostly tested


@pytest.mark.xfail(strict=True, reason="known deviation of the fast modes' second phase: the reference "
                   "keeps a record at the priority of its last re-score (taken when the objects were "
                   "smaller), components mode scores the records between components afresh; here one "
                   "record is non-negative afresh but was last scored negative in the reference, so it is "
                   "merged here and never looked at there (DESIGN.md section 5).  EXACT mode and, on this "
                   "input, the rounds give the reference's result.")
def test_components_mode_known_deviation_stale_record(oracle):
    """Found by the fuzz test above with only two offsets [(1,0), (-1,5)]: the instances fall
    into interleaved column components that no offset links.  Kept as a witness."""
    offs = [(1, 0), (-1, 5)]
    s = synth.synth_v1(47, 72, 9, offs, 9004, noise=0.1)
    ref = oracle.run_csegment(s.class_probs, s.sameness_probs, 9, offs, 0.0, 1.0, 0.03)
    for mode in (seg.MN_MODE_EXACT, seg.MN_MODE_ROUNDS):
        mask, classes, part, st = _segment(s.class_probs, s.sameness_probs, offs, mode)
        assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), (mode, st)
    mask, classes, part, st = _segment(s.class_probs, s.sameness_probs, offs, seg.MN_MODE_COMPONENTS)
    assert st["certified"] == 0          # the flag that says "not proven": it must not claim more
    assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), st



@pytest.mark.parametrize("shape", [(1, 9, 3, 2), (9, 1, 3, 2), (3, 5, 2, 2), (2, 130, 3, 3), (17, 65, 4, 5),
                                   (16, 64, 2, 4), (15, 63, 3, 4), (20, 24, 127, 3), (24, 40, 3, 32),
                                   (24, 40, 3, 31)])
def test_components_mode_extreme_shapes(oracle, shape):
    """One-pixel-wide images, sizes just off the 16x64 tile and the 4-pixel lane, the maximum
    class count (127) and offset count (32): components mode forced, result = the oracle's."""
    H, W, C, O = shape
    if O <= 8:
        offs = synth.generate_offsets(max(3, min(H, W) // 2 + 2), O)[:O]
    else:                                      # 32 distinct offsets of one half-plane (no negations)
        offs = [(di, dj) for di in range(0, 5) for dj in range(-4, 5) if di > 0 or dj > 0][:O]
    s = synth.synth_v1(H, W, C, offs, 31 + H * W, noise=0.12, num_instances=2)
    ref = oracle.run_csegment(s.class_probs, s.sameness_probs, C, offs, 0.0, 1.0, 0.03)
    mask, classes, part, st = _segment(s.class_probs, s.sameness_probs, offs, seg.MN_MODE_COMPONENTS)
    assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), (shape, offs, st)
    assert oracle.same_partition(part, ref.partition), (shape, st)
    assert abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob), (shape, st)
    if O >= 31:
        # the sweep takes offsets five at a time: offsets 32..34 of the last group do not exist and must
        # not set bits 0..2 (they did: every O >= 31 image then failed the separability check)
        assert st["mode_used"] == seg.MN_MODE_COMPONENTS, st
        for flags in (0, seg.MN_DEBUG_NO_CORES):
            ctx = seg.HostContext(H, W, C, len(offs))
            try:
                o = seg.default_options(merge_logprob_bias=0.03, mode=seg.MN_MODE_ROUNDS, clip_inputs=1, debug_flags=flags)
                m2, c2, p2, st2 = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
            finally:
                ctx.close()
            assert oracle.masks_equivalent(m2, c2, ref.mask, ref.object_class), (flags, st2)


def test_compute_logprob_off_skips_certificate_but_not_the_result(oracle):
    g = gu.load("cseg_synth_128x256")
    for mode in (seg.MN_MODE_COMPONENTS, seg.MN_MODE_ROUNDS):
        mask, classes, part, st = _run(g, mode, compute_logprob=0)
        assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st
        assert np.isnan(st["total_logprob"]) and st["certified"] == 0
        mask1, classes1, part1, st1 = _run(g, mode)
        assert np.array_equal(mask, mask1) and classes == classes1
        assert np.isfinite(st1["total_logprob"])


def test_require_proof_routes_to_the_sequential_order(oracle):
    """mn_options.require_proof = 1: a result that is only an approximation of the reference's order is
    redone in MN_MODE_EXACT -- at any image size since the exact engine (round 2 returned MN_ERR_UNPROVEN
    at 256x512); a certified result passes as it is."""
    g = gu.load("cseg_blur_64x128_r2")                       # not sign-separable, 80 k records
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO, require_proof=1)
    assert st["mode_used"] == seg.MN_MODE_EXACT and st["proof"] == seg.MN_PROOF_SEQUENTIAL
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st
    g = gu.load("cseg_blur_256x512_r2")                      # 1.25 M records
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO, require_proof=1)
    assert st["mode_used"] == seg.MN_MODE_EXACT and st["proof"] == seg.MN_PROOF_SEQUENTIAL
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st
    g = gu.load("cseg_synth_256x512")                        # separable and certified: nothing to redo
    mask, classes, part, st = _run(g, seg.MN_MODE_AUTO, require_proof=1)
    assert st["proof"] == seg.MN_PROOF_CERTIFICATE and st["certified"] == 1
    assert oracle.masks_equivalent(mask, classes, g["mask"], g["object_class"]), st


def test_replay_of_recorded_graphs_gives_the_same_result():
    """debug_flags bit 5: from the third call through the same buffers the launches after the sweep are
    replayed from two recorded hipGraphs; the result must not change, and a call with other buffers
    must not use the old recording."""
    import torch
    offs = synth.generate_offsets(40, 10)
    H, W, C = 256, 512, 9
    imgs = []
    for sd in (1000, 1001):
        s = synth.synth_v1(H, W, C, offs, sd)
        imgs.append((torch.from_numpy(s.class_probs).cuda(), torch.from_numpy(s.sameness_probs).cuda()))
    m = seg.Merger(H, W, C, len(offs))
    plain = seg.default_options(merge_logprob_bias=0.03, mode=seg.MN_MODE_COMPONENTS)
    refs = [m.segment(cp, sp, offs, plain) for cp, sp in imgs]
    assert refs[0][3]["mode_used"] == seg.MN_MODE_COMPONENTS
    o = seg.default_options(merge_logprob_bias=0.03, mode=seg.MN_MODE_COMPONENTS,
                            debug_flags=seg.MN_DEBUG_LEAN_EVENTS | seg.MN_DEBUG_REPLAY)
    out = (torch.empty((H, W), dtype=torch.int32, device="cuda"), torch.empty((H * W,), dtype=torch.int32, device="cuda"))
    for which in (0, 0, 0, 0, 1, 1, 1, 0, 0, 0):
        out[0].zero_()
        mask, table, _, st = m.segment_async(imgs[which][0], imgs[which][1], offs, o, out=out).result()
        rm, rt, _, rs = refs[which]
        assert st["num_instances"] == rs["num_instances"] and st["total_logprob"] == rs["total_logprob"]
        assert torch.equal(mask, rm) and torch.equal(table[:st["num_instances"]], rt[:rs["num_instances"]])


def test_core_with_a_non_positive_edge_inside_falls_apart(oracle):
    """Two same-class instances side by side whose SHORT edges are made positive across a stretch of
    their common boundary: the clean pixels chain across, the core holds long edges that are negative,
    mn_core_check condemns it, and the rounds start from single pixels there.  The result must stay a
    valid answer close to the reference's; without the bridge no core is condemned."""
    offs = synth.generate_offsets(20, 10)
    H, W, C = 48, 96, 3
    inst = np.zeros((H, W), np.int32)
    inst[:, : W // 2] = 1
    inst[:, W // 2:] = 2
    cp = np.full((C, H, W), 0.05, np.float32)
    cp[1] = 0.9                                             # both halves of class 1
    rng = np.random.default_rng(5)
    short = [k for k, (di, dj) in enumerate(offs) if abs(di) <= 6 and abs(dj) <= 6]

    def maps(bridge):
        sp = np.ones((len(offs), H, W), np.float32)
        for k, (di, dj) in enumerate(offs):
            r0, r1 = max(0, -di), min(H, H - di)
            c0, c1 = max(0, -dj), min(W, W - dj)
            same = inst[r0:r1, c0:c1] == inst[r0 + di:r1 + di, c0 + dj:c1 + dj]
            sp[k, r0:r1, c0:c1] = np.where(same, 0.9, 0.1)
        if bridge:
            for k in short:
                sp[k, 16:32, W // 2 - 8: W // 2 + 8] = 0.8    # every short edge near the boundary says "same"
        sp += rng.uniform(-0.03, 0.03, sp.shape).astype(np.float32)
        return np.clip(sp, 0.01, 0.99).astype(np.float32)

    for bridge in (False, True):
        sp = maps(bridge)
        ref = oracle.run_csegment(cp, sp, C, offs, 0.0, 1.0, 0.03)
        ctx = seg.HostContext(H, W, C, len(offs))
        o = seg.default_options(merge_logprob_bias=0.03, mode=seg.MN_MODE_ROUNDS, clip_inputs=1)
        mask, classes, part, st = ctx.segment(cp, sp, offs, o)
        ctx.close()
        assert st["status"] == 0 and st["cores_condemned"] == (1 if bridge else 0), st
        assert mask.min() >= 0 and mask.max() == len(classes) == st["num_instances"]
        assert labels.agreement(mask, ref.mask) >= 0.97 * mask.size, (bridge, st, len(ref.object_class))


@pytest.mark.parametrize("name,floor", [("cseg_blur_64x128_r2", 0.985), ("cseg_blur_64x128_r2_s8001", 0.985),
                                        ("cseg_blur_256x512_r2", 0.995)])
def test_blurred_maps_general_path_stays_close_to_the_reference(oracle, name, floor):
    """Not equality (strict xfail above) but a floor under the approximation: the general path (cores of
    radius 6, band 0.05, hand-over at 2048) finds the reference's instance count and agrees with its
    partition on 99.0-99.7 % of the pixels of these vectors; a change that moves it further away --
    cores from a radius of 3 lose an instance on the 64x128 vectors -- turns this red."""
    g = gu.load(name)
    mask, classes, part, stats = _run(g, seg.MN_MODE_ROUNDS)
    assert stats["mode_used"] == seg.MN_MODE_ROUNDS and stats["proof"] == 0
    assert len(classes) == len(g["object_class"])
    assert labels.agreement(mask, g["mask"]) >= floor * mask.size, stats


@pytest.mark.parametrize("shape", [(33, 47), (50, 70), (61, 96), (64, 101), (97, 130)])
@pytest.mark.parametrize("flags", [seg.MN_DEBUG_NO_CORES, 0])
def test_general_path_on_odd_shapes_equals_oracle(oracle, shape, flags):
    """Widths and pixel counts that are not multiples of 4 (the one-pixel-per-lane forms of the sweep,
    the separate class sweep, tail lanes of every 4-pixel kernel), through the general path from the
    cores (flags 0) and from single pixels (flags 4): separable maps, so the reference's result is expected
    exactly."""
    H, W = shape
    offs = synth.generate_offsets(12, 8)
    for seed in (4100, 4101, 4102):
        s = synth.synth_v1(H, W, 5, offs, seed, noise=0.2, num_instances=4)
        ref = oracle.run_csegment(s.class_probs, s.sameness_probs, 5, offs, 0.0, 1.0, 0.03)
        ctx = seg.HostContext(H, W, 5, len(offs))
        try:
            o = seg.default_options(merge_logprob_bias=0.03, mode=seg.MN_MODE_ROUNDS, clip_inputs=1, debug_flags=flags)
            mask, classes, part, st = ctx.segment(s.class_probs, s.sameness_probs, offs, o)
        finally:
            ctx.close()
        assert st["status"] == 0
        assert oracle.masks_equivalent(mask, classes, ref.mask, ref.object_class), (shape, seed, flags, st)
        assert abs(st["total_logprob"] - ref.total_logprob) <= 1e-5 * abs(ref.total_logprob)
