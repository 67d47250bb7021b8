"""Phase A (affinity scoring) of the HIP library against the oracle's phase-A arrays (GPU box).

Reference: the constructor of utils/csegment/segment.cc:153-232 -- per-pixel arg-max class
(:5-21), per in-bounds edge log-odds (:24-46) and initial merge priority (:107-150).  The HIP
edge pass keeps, per pixel, only the best incident record (priority, partner); that is compared
with the arg-max over the oracle's per-edge priorities.  cls is integer-exact; priorities are
float32 computed with the GPU's logf / log(1-p) (<= 2 ulp from glibc's), tolerance 1e-5 relative
(the tolerance BASELINE.json states for log-likelihood terms).
"""
import numpy as np
import pytest

from mergenet_amd import segmenter as seg
from mergenet_amd import synth

pytestmark = pytest.mark.gpu


def _decode(best):
    b = best.astype(np.uint64)
    bits = (b >> np.uint64(32)).astype(np.uint32)
    prio = bits.view(np.float32)
    partner = (np.uint64(0x7FFFFFFF) - (b & np.uint64(0x7FFFFFFF))).astype(np.int64)
    has = b != 0
    return np.where(has, prio, np.nan).astype(np.float32), np.where(has, partner, -1)


def _score(merger, s, offs, opts):
    import torch
    cp = torch.from_numpy(np.ascontiguousarray(s.class_probs)).cuda()
    sp = torch.from_numpy(np.ascontiguousarray(s.sameness_probs)).cuda()
    _, _, cls, best = merger.score(cp, sp, offs, opts, want_arrays=True)
    torch.cuda.synchronize()
    return cls.cpu().numpy(), best.cpu().numpy()


@pytest.mark.parametrize("shape", [(256, 512, 9, (40, 10), 0.15), (1024, 2048, 9, (40, 10), 0.15),
                                   (400, 667, 81, (80, 16), 0.15), (96, 160, 5, (12, 6), 0.45)])
def test_phase_a_arrays_equal_the_oracle(oracle, shape):
    H, W, C, oa, noise = shape
    offs = synth.generate_offsets(*oa)
    s = synth.synth_v1(H, W, C, offs, 1000, noise=noise, occlusion=(C == 81))
    merger = seg.Merger(H, W, C, len(offs))
    try:
        opts = seg.default_options(clip_inputs=1)
        cls, best = _score(merger, s, offs, opts)
        ref_cls, ref_oml, ref_prio = oracle.phase_a(s.class_probs, s.sameness_probs, C, offs, 0.0, 1.0, 0.03)
        assert np.array_equal(cls.astype(np.int32), ref_cls)             # integer-exact arg-max
        prio, partner = _decode(best)
        rprio, rpartner = oracle.best_initial_record(ref_prio, offs)
        assert np.array_equal(np.isnan(prio), np.isnan(rprio))           # same pixels have a record >= 0
        ok = ~np.isnan(rprio)
        rel = np.abs(prio[ok] - rprio[ok]) / np.maximum(1.0, np.abs(rprio[ok]))
        assert rel.max() <= 1e-5, rel.max()
        same = partner[ok] == rpartner[ok]
        # a different partner is only acceptable on a near-tie: the oracle's priority of the edge
        # the GPU chose must equal the oracle's maximum to the same tolerance
        if not same.all():
            pg = partner.copy()
            idx = np.argwhere(ok & (partner != rpartner))
            for (r, c) in idx[:2000]:
                p, q = r * W + c, int(pg[r, c])
                alt = None
                for k, (di, dj) in enumerate(offs):
                    if q == p + di * W + dj and 0 <= c + dj < W:
                        alt = ref_prio[k, r, c]
                    if q == p - di * W - dj and 0 <= c - dj < W and 0 <= r - di < H:
                        alt = ref_prio[k, r - di, c - dj]
                assert alt is not None and abs(alt - rprio[r, c]) <= 1e-5 * max(1.0, abs(rprio[r, c]))
        assert same.mean() >= 0.999, same.mean()
        # the fast form of the edge pass and the generic one are the same function
        cls_g, best_g = _score(merger, s, offs, seg.default_options(clip_inputs=1, debug_flags=1))
        assert np.array_equal(cls_g, cls)
        pg_, qg_ = _decode(best_g)
        assert np.array_equal(np.isnan(pg_), np.isnan(prio))
        assert np.array_equal(pg_[ok], prio[ok])
        assert (qg_[ok] == partner[ok]).mean() >= 0.9999
    finally:
        merger.close()


def test_phase_a_with_same_different_bias_and_other_options(oracle):
    """same_different_bias != 0 (applied on load; the reference rewrites the plane, segment.cc:183-195),
    merge factor 0.5, bias 0: generic edge pass."""
    H, W, C = 64, 96, 4
    offs = synth.generate_offsets(10, 6)
    s = synth.synth_v1(H, W, C, offs, 5, noise=0.3)
    merger = seg.Merger(H, W, C, len(offs))
    try:
        opts = seg.default_options(clip_inputs=1, same_different_bias=0.4, object_merge_factor=0.5,
                                   merge_logprob_bias=0.0)
        cls, best = _score(merger, s, offs, opts)
        ref_cls, _, ref_prio = oracle.phase_a(s.class_probs, s.sameness_probs, C, offs, 0.4, 0.5, 0.0)
        assert np.array_equal(cls.astype(np.int32), ref_cls)
        prio, partner = _decode(best)
        rprio, rpartner = oracle.best_initial_record(ref_prio, offs)
        assert np.array_equal(np.isnan(prio), np.isnan(rprio))
        ok = ~np.isnan(rprio)
        assert (np.abs(prio[ok] - rprio[ok]) / np.maximum(1.0, np.abs(rprio[ok]))).max() <= 1e-5
        assert (partner[ok] == rpartner[ok]).mean() >= 0.999
    finally:
        merger.close()


# ---- the sweep of the DEFAULT path (mn_cc_sign) pinned on its own --------------------------------------
# Round 2 pinned phase A through mn_score_device, i.e. through the round-1 kernels; the default path's
# sweep was covered end to end only.  mn_sweep_device exports what that sweep leaves: edge-sign masks,
# the negative-edge list, per-lane class log-products, arg-max classes, the certificate's log sum.

@pytest.mark.parametrize("shape", [(256, 512, 9, (40, 10), 0.15, 0.0, 0),
                                   (1024, 2048, 9, (40, 10), 0.15, 0.0, 0),
                                   (400, 667, 81, (80, 16), 0.15, 0.0, 0),      # W % 4 != 0
                                   (96, 160, 5, (12, 6), 0.45, 0.0, 1),         # clip fused into the loads
                                   (128, 256, 9, (40, 10), 0.25, 0.4, 1),       # same_different_bias != 0
                                   (24, 40, 3, None, 0.12, 0.0, 1)])            # 32 offsets
def test_default_path_sweep_equals_the_oracles_phase_a(oracle, shape):
    import torch
    H, W, C, oa, noise, sdb, clip = shape
    if oa is None:
        offs = [(di, dj) for di in range(0, 5) for dj in range(-4, 5) if di > 0 or dj > 0][:32]
    else:
        offs = synth.generate_offsets(*oa)
    s = synth.synth_v1(H, W, C, offs, 1000, noise=noise, occlusion=(C == 81), num_instances=2 if oa is None else None)
    merger = seg.Merger(H, W, C, len(offs))
    try:
        o = seg.default_options(same_different_bias=sdb, clip_inputs=clip)
        cp = torch.from_numpy(np.ascontiguousarray(s.class_probs)).cuda()
        sp = torch.from_numpy(np.ascontiguousarray(s.sameness_probs)).cuda()
        out = merger.sweep(cp, sp, offs, o)
        ref_cls, ref_oml, ref_prio = oracle.phase_a(s.class_probs, s.sameness_probs, C, offs, sdb, 1.0, 0.03)
        ok = ~np.isnan(ref_oml)
        # edges inside the float32 rounding margin of 0.5 (|log-odds| below ~2 N ulp(bias)) are neither
        # positive nor negative for the sweep: it counts them (the image then fails the separability
        # check); a noisy map holds a few, they are left out of the sign comparisons
        with np.errstate(invalid="ignore"):
            margin = ok & (np.abs(ref_oml) < 1e-4)
        assert out["margin_edges"] <= int(margin.sum())
        assert noise > 0.4 or out["margin_edges"] == 0
        ok = ok & ~margin
        bits = out["bits"].cpu().numpy().view(np.uint32)
        # bit k set <=> the edge is in bounds and the oracle's log-odds are > 0
        for k in range(len(offs)):
            got = ((bits >> np.uint32(k)) & np.uint32(1)).astype(bool)
            assert np.array_equal(got & ~margin[k], ok[k] & (ref_oml[k] > 0)), k
        if len(offs) < 32:
            assert not (bits >> np.uint32(len(offs))).any()          # no bit of an offset that does not exist
        # every negative in-bounds edge is listed with the oracle's log-odds (1e-5 relative), none else
        neg = out["neg"].cpu().numpy()
        want = ok & (ref_oml < 0)
        assert np.array_equal(~np.isnan(neg) & ~margin, want)
        # (1e-5 relative; plus 1e-6 absolute, a few float32 ulps of the two logs whose difference the
        #  log-odds are: near v = 0.5 they cancel and a purely relative bound is not meaningful)
        err = np.abs(neg[want] - ref_oml[want]) - 1e-5 * np.abs(ref_oml[want])
        assert err.size == 0 or err.max() <= 1e-6, err.max()
        # certificate's sum: log max(v, 1 - v) over the in-bounds edges = -log(1 + exp(-|log-odds|))
        want_sum = float(-np.log1p(np.exp(-np.abs(ref_oml[ok | margin].astype(np.float64)))).sum())
        assert abs(out["logsum"] - want_sum) <= 1e-5 * abs(want_sum), (out["logsum"], want_sum)
        assert out["pixels_per_lane"] == 4 or W % 4 != 0
        if out["fused_class"]:
            assert np.array_equal(out["cls"].cpu().numpy().astype(np.int32), ref_cls)      # integer-exact arg-max
            gsum = out["gsum"].cpu().numpy().astype(np.float64) / 16777216.0               # 2^-24 fixed point
            cpc = np.clip(s.class_probs.astype(np.float64), np.finfo(np.float32).eps, 1 - np.finfo(np.float32).eps)
            for c in range(C):
                want_c = float(np.log(cpc[c]).sum())
                assert abs(gsum[c].sum() - want_c) <= 1e-6 * abs(want_c), (c, gsum[c].sum(), want_c)
        else:
            assert W % 4 != 0 or (H * W) % 4 != 0
    finally:
        merger.close()
