"""Producer hand-off and mask post-processing kernels against a plain PyTorch fp32 reference.

Float kernel => torch reference (cv2 is not installed here): F.interpolate(mode="bilinear",
align_corners=False) uses cv2.resize/INTER_LINEAR's coordinate mapping, mode="nearest" is cv2's
INTER_NEAREST.  Tolerance: 2e-6 absolute on probabilities (float32 blend order differs).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [((19, 64, 96), (32, 48)), ((5, 50, 70), (25, 35)),
                                   ((3, 33, 47), (64, 90)), ((2, 17, 9), (17, 9))])
@pytest.mark.parametrize("sigmoid", [False, True])
def test_prepare_matches_torch_bilinear(shape, sigmoid):
    import torch
    import torch.nn.functional as F
    from mergenet_amd import segmenter as seg
    (K, Hin, Win), (Ho, Wo) = shape
    g = torch.Generator().manual_seed(7)
    x = torch.randn((K, Hin, Win), generator=g) * 3 if sigmoid else torch.rand((K, Hin, Win), generator=g)
    x = x.cuda().contiguous()
    m = seg.Merger(max(Hin, Ho), max(Win, Wo), 4, 4)
    got = m.prepare(x, Ho, Wo, apply_sigmoid=sigmoid, clip=True)
    src = torch.sigmoid(x) if sigmoid else x
    ref = F.interpolate(src[None], size=(Ho, Wo), mode="bilinear", align_corners=False)[0]
    eps = float(np.finfo(np.float32).eps)
    ref = ref.clamp(eps, 1.0 - eps)
    assert got.shape == ref.shape
    assert float((got - ref).abs().max()) <= 2e-6
    m.close()


@pytest.mark.parametrize("shape", [((32, 48), (64, 96)), ((25, 35), (50, 70)), ((40, 56), (37, 129))])
def test_upsample_mask_matches_torch_nearest(shape):
    import torch
    import torch.nn.functional as F
    from mergenet_amd import segmenter as seg
    (Hin, Win), (Ho, Wo) = shape
    g = torch.Generator().manual_seed(9)
    mask = torch.randint(0, 50, (Hin, Win), generator=g, dtype=torch.int32).cuda()
    m = seg.Merger(max(Hin, Ho), max(Win, Wo), 4, 4)
    got = m.upsample_mask(mask, Ho, Wo)
    ref = F.interpolate(mask[None, None].float(), size=(Ho, Wo), mode="nearest")[0, 0].to(torch.int32)
    assert torch.equal(got, ref)
    m.close()


def test_prepare_then_segment_equals_host_pipeline(oracle):
    """logits -> (device) sigmoid + resize + clip -> merger, against the same steps done with
    torch on the host feeding the CPU oracle (the reference caller's order, segment.py:110-138)."""
    import torch
    import torch.nn.functional as F
    from mergenet_amd import segmenter as seg, synth
    offs = synth.generate_offsets(8, 6)
    C, O = 4, len(offs)
    s = synth.synth_v1(96, 128, C, offs, 77, num_instances=3)
    probs = np.concatenate([s.class_probs, s.sameness_probs]).clip(1e-4, 1 - 1e-4)
    logits = torch.from_numpy(np.log(probs / (1 - probs)).astype(np.float32)).cuda()
    m = seg.Merger(96, 128, C, O)
    maps = m.prepare(logits, 48, 64, apply_sigmoid=True, clip=True)
    # the halved maps no longer match the offsets' geometry -> order-dependent input: EXACT mode
    o = seg.default_options(mode=seg.MN_MODE_EXACT)
    mask, table, _, st = m.segment(maps[:C].contiguous(), maps[C:].contiguous(), offs, o)
    big = m.upsample_mask(mask, 96, 128)
    host = maps.cpu().numpy()      # same bits as the merger saw (kernel already checked vs torch)
    ref = oracle.run_csegment(host[:C], host[C:], C, offs, 0.0, 1.0, 0.03)
    got = [int(c) for c in table.cpu().numpy()[: st["num_instances"]]]
    assert oracle.masks_equivalent(mask.cpu().numpy(), got, ref.mask, ref.object_class), st
    ref_big = F.interpolate(torch.from_numpy(ref.mask)[None, None].float(), size=(96, 128), mode="nearest")[0, 0]
    assert oracle.same_partition(big.cpu().numpy(), ref_big.numpy().astype(np.int32))
    m.close()


def test_sameness_targets_match_numpy_roll_definition():
    """np.roll compare with the border forced to 1, as utils/dataset.py:259-277 defines it."""
    import torch
    from mergenet_amd import segmenter as seg, synth
    rng = np.random.default_rng(5)
    H, W = 37, 53
    mask = rng.integers(0, 4, (H, W)).astype(np.int32)
    offs = synth.generate_offsets(10, 6)
    expect = np.zeros((len(offs), H, W), np.float32)
    for n, (i, j) in enumerate(offs):
        rolled = np.roll(np.roll(mask, -i, axis=0), -j, axis=1)
        t = (rolled == mask)
        if i < 0:
            t[:-i, :] = 1
        elif i > 0:
            t[-i:, :] = 1
        if j < 0:
            t[:, :-j] = 1
        elif j > 0:
            t[:, -j:] = 1
        expect[n] = t
    m = seg.Merger(H, W, 4, len(offs))
    got = m.sameness_targets(torch.from_numpy(mask).cuda(), offs).cpu().numpy()
    assert np.array_equal(got, expect)
    m.close()


def test_instance_scores_are_the_class_margin(oracle):
    import torch
    from mergenet_amd import segmenter as seg, synth
    offs = synth.generate_offsets(40, 10)
    s = synth.synth_v1(64, 128, 9, offs, 1001, num_instances=4)
    m = seg.Merger(64, 128, 9, len(offs))
    cp, sp = torch.from_numpy(s.class_probs).cuda(), torch.from_numpy(s.sameness_probs).cuda()
    mask, table, _, st = m.segment(cp, sp, offs, seg.default_options(clip_inputs=1))
    K = st["num_instances"]
    scores = m.instance_scores(K).cpu().numpy()
    lp = np.log(s.class_probs.clip(np.finfo(np.float32).eps, 1 - np.finfo(np.float32).eps).astype(np.float64))
    mk = mask.cpu().numpy()
    tb = table.cpu().numpy()
    for k in range(1, K + 1):
        sel = mk == k
        want = lp[tb[k - 1]][sel].sum() - lp[0][sel].sum()
        assert abs(scores[k - 1] - want) <= 1e-4 * abs(want)
    m.close()


def test_pack_wire_matches_layout():
    """mn_pack_wire_device: int16 [labels | K | classes padded with -1] (MaskExchange wire format)."""
    import torch
    from mergenet_amd import distributed as mnd
    from mergenet_amd import segmenter as seg
    for (H, W, K) in [(6, 10, 3), (7, 9, 0), (64, 128, 40)]:       # n % 4 != 0 included
        g = torch.Generator().manual_seed(H * W + K)
        mask = torch.randint(0, K + 1, (H, W), generator=g, dtype=torch.int32).cuda()
        table = torch.full((H * W,), -1, dtype=torch.int32)
        table[:K] = torch.randint(1, 100, (K,), generator=g, dtype=torch.int32)
        table = table.cuda()
        n = H * W
        wire = torch.full((n + 1 + mnd.MAX_INSTANCES + 4,), 12345, dtype=torch.int16, device="cuda")
        seg.pack_wire(mask, table, K, wire, mnd.MAX_INSTANCES, -12345.6789)
        torch.cuda.synchronize()
        w = wire.cpu()
        assert w[:n].tolist() == mask.cpu().reshape(-1).tolist()
        assert int(w[n]) == K
        assert w[n + 1: n + 1 + K].tolist() == table[:K].cpu().tolist()
        assert bool((w[n + 1 + K: n + 1 + mnd.MAX_INSTANCES] == -1).all())
        assert float(w[n + 1 + mnd.MAX_INSTANCES:].clone().view(torch.float64)[0]) == -12345.6789
    with pytest.raises(seg.MergeNetError):
        seg.pack_wire(mask, table, mnd.MAX_INSTANCES + 1, wire, mnd.MAX_INSTANCES)


@pytest.mark.parametrize("fmt", ["runs", "int16"])
def test_mask_exchange_single_gpu_uses_the_pack_kernel(fmt):
    """MaskExchange on device tensors without a process group: wire packed (and, for the run-length
    format, unpacked) by the HIP kernels; both formats deliver the same mask, table, count, log-likelihood."""
    import torch
    from mergenet_amd import distributed as mnd
    from mergenet_amd import segmenter as seg
    H, W, K = 33, 47, 5
    mask = torch.zeros((H, W), dtype=torch.int32)
    mask[3:8, 5:30] = 2                 # ~35 row-major label changes: within the 64 a 33x47 wire holds
    mask[10:20, 25:47] = 5
    mask[0, 0] = 1
    mask = mask.cuda()
    table = torch.full((H * W,), -1, dtype=torch.int32)
    table[:K] = torch.tensor([3, 1, 4, 1, 5], dtype=torch.int32)
    table = table.cuda()
    merger = seg.Merger(H, W, 3, 2)
    ex = mnd.MaskExchange(H, W, torch.device("cuda", 0), fmt=fmt, merger=merger)
    slots = [ex.submit(mask, table, K, -77.125) for _ in range(3)]     # reuses both buffers
    masks, tabs, counts = ex.result(slots[-1])
    torch.cuda.synchronize()
    assert masks.shape == (1, H, W) and bool((masks[0].to(torch.int32) == mask).all())
    assert tabs[0, :K].tolist() == [3, 1, 4, 1, 5] and bool((tabs[0, K:] == -1).all())
    assert int(counts[0]) == K and ex.logprobs(slots[-1]).tolist() == [-77.125]
    ex.drain()
    merger.close()


def test_merger_pool_equals_serial_merger_and_keeps_order():
    """Four contexts / streams / host threads: every result equals the single-context result of
    the same image, in submission order; inputs produced on the caller's stream just before."""
    import torch
    from mergenet_amd import segmenter as seg, synth
    offs = synth.generate_offsets(12, 6)
    H, W, C = 96, 160, 4
    images = []
    for seed in range(6):
        s = synth.synth_v1(H, W, C, offs, 500 + seed)
        images.append((torch.from_numpy(s.class_probs).cuda(), torch.from_numpy(s.sameness_probs).cuda()))
    opts = seg.default_options(clip_inputs=1)
    single = seg.Merger(H, W, C, len(offs))
    want = [single.segment(cp, sp, offs, opts) for cp, sp in images]
    pool = seg.MergerPool(H, W, C, len(offs), depth=4)
    # inputs "produced" right before submission on the current stream: a scaled copy
    fresh = [((cp * 1.0).contiguous(), (sp * 1.0).contiguous()) for cp, sp in images] * 3
    got = pool.map(fresh, offs, opts)
    assert len(got) == 18
    for k, (mask, table, _, st) in enumerate(got):
        wmask, wtable, _, wst = want[k % 6]
        assert torch.equal(mask, wmask) and st["num_instances"] == wst["num_instances"]
        assert torch.equal(table[: st["num_instances"]], wtable[: wst["num_instances"]])
        assert st["total_logprob"] == wst["total_logprob"]
    pool.close()
    single.close()
    with pytest.raises(RuntimeError):
        pool.submit(images[0][0], images[0][1], offs, opts)


def test_segment_async_two_contexts_one_stream_equals_segment():
    """mn_segment_launch / mn_segment_finish: launches of image i+1 ahead of the read-back of
    image i give the results of the blocking call, in every mode (fallbacks included)."""
    import torch
    from mergenet_amd import segmenter as seg, synth
    offs = synth.generate_offsets(12, 6)
    H, W, C = 96, 160, 4
    cases = []
    for seed, noise in [(700, 0.15), (701, 0.45), (702, 0.15), (703, 0.15)]:     # 0.45: falls back
        s = synth.synth_v1(H, W, C, offs, seed, noise=noise)
        cases.append((torch.from_numpy(s.class_probs).cuda(), torch.from_numpy(s.sameness_probs).cuda()))
    a, b = seg.Merger(H, W, C, len(offs)), seg.Merger(H, W, C, len(offs))
    for mode in (seg.MN_MODE_COMPONENTS, seg.MN_MODE_ROUNDS, seg.MN_MODE_AUTO):
        opts = seg.default_options(mode=mode, clip_inputs=1)
        want = [a.segment(cp, sp, offs, opts) for cp, sp in cases]
        got, prev = [], None
        for i, (cp, sp) in enumerate(cases):
            cur = (a, b)[i % 2].segment_async(cp, sp, offs, opts)
            if prev is not None:
                got.append(prev.result())
            prev = cur
        got.append(prev.result())
        for (m, t, _, st), (wm, wt, _, wst) in zip(got, want):
            assert torch.equal(m, wm) and st["num_instances"] == wst["num_instances"]
            assert st["mode_used"] == wst["mode_used"] and st["total_logprob"] == wst["total_logprob"]
            assert torch.equal(t[: st["num_instances"]], wt[: wst["num_instances"]])
    # a busy context refuses a second launch and the blocking call
    p = a.segment_async(cases[0][0], cases[0][1], offs, opts)
    with pytest.raises(seg.MergeNetError):
        a.segment_async(cases[1][0], cases[1][1], offs, opts)
    with pytest.raises(seg.MergeNetError):
        a.segment(cases[1][0], cases[1][1], offs, opts)
    assert p.result()[3]["num_instances"] == want[0][3]["num_instances"]
    a.close(); b.close()


def test_run_length_wire_device_equals_cpu_twin_and_round_trips():
    """mn_pack_runs_device / mn_unpack_runs_device (the default wire of the multi-GPU mask exchange)
    against the CPU twins in mergenet_amd.distributed, on a real 1024x2048 result mask (24 instances)
    and on a small ragged one; plus the overflow report for a mask with too many label changes."""
    import os
    import torch
    import golden_util as gu
    from mergenet_amd import distributed as mnd
    from mergenet_amd import segmenter as seg
    z = np.load(os.path.join(gu.GOLDEN, "cseg_synth_1024x2048_cfg2.npz"))
    cases = [(z["mask"].astype(np.int32), [int(c) for c in z["object_class"]])]
    rng = np.random.default_rng(3)
    small = np.repeat(np.repeat(rng.integers(0, 5, (5, 9)), 7, 0), 11, 1)[:33, :97].astype(np.int32)
    cases.append((small, [1, 2, 3, 4]))
    for mask_np, classes in cases:
        H, W = mask_np.shape
        merger = seg.Merger(H, W, 9, 10)
        try:
            cap = max(mnd.runs_capacity(H * W), 2048)
            k = len(classes)
            mask = torch.from_numpy(mask_np).cuda()
            table = torch.full((H * W,), -1, dtype=torch.int32, device="cuda")
            table[:k] = torch.tensor(classes, dtype=torch.int32, device="cuda")
            wire = torch.zeros(seg.runs_wire_words(cap, mnd.MAX_INSTANCES), dtype=torch.int32, device="cuda")
            seg.pack_runs(merger, mask, table, k, wire, cap, mnd.MAX_INSTANCES, -4321.125)
            ref = mnd.pack_runs_cpu(torch.from_numpy(mask_np), table.cpu(), k, -4321.125, cap)
            n = int(ref[0])
            got = wire.cpu()
            assert int(got[0]) == n and n > 0
            assert torch.equal(got[:4 + n], ref[:4 + n])                                   # header + positions
            lab_g = got[4 + cap: 4 + cap + (cap + 1) // 2].view(torch.int16)[:n]
            lab_r = ref[4 + cap: 4 + cap + (cap + 1) // 2].view(torch.int16)[:n]
            assert torch.equal(lab_g, lab_r)
            assert torch.equal(got[4 + cap + (cap + 1) // 2:], ref[4 + cap + (cap + 1) // 2:])   # classes
            m2, t2 = seg.unpack_runs(wire, H, W, cap, mnd.MAX_INSTANCES)
            assert torch.equal(m2.cpu(), torch.from_numpy(mask_np))
            assert t2[:k].cpu().tolist() == classes and bool((t2[k:] == -1).all())
            if H * W > 100000:
                assert 10 * wire.numel() * 4 <= 2 * H * W + 2 * (mnd.MAX_INSTANCES + 5) + 40 * 1024   # >= 10x below int16
        finally:
            merger.close()
    # too many changes for the capacity: reported in the header, not written past the buffer
    H, W = 64, 64
    merger = seg.Merger(H, W, 3, 2)
    try:
        noisy = torch.from_numpy(rng.integers(0, 7, (H, W)).astype(np.int32)).cuda()
        cap = 128
        wire = torch.zeros(seg.runs_wire_words(cap, 16) + 8, dtype=torch.int32, device="cuda")
        wire[-8:] = 777
        seg.pack_runs(merger, noisy, torch.zeros(16, dtype=torch.int32, device="cuda"), 3, wire, cap, 16, 0.0)
        assert int(wire[0]) == -1 and bool((wire[-8:] == 777).all())
    finally:
        merger.close()


@pytest.mark.gpu
def test_run_length_wires_of_all_ranks_unpack_in_one_launch():
    """mn_unpack_runs_batch_device: the gathered wires of several ranks (rows of a strided 2-D buffer)
    restored by ONE launch equal the one-by-one results; a wire whose header claims more change points
    than the capacity is clamped (the search stays inside the wire's own sections)."""
    import torch
    from mergenet_amd import distributed as mnd
    from mergenet_amd import segmenter as seg
    rng = np.random.default_rng(9)
    H, W = 40, 72
    cap = 1024                                # (room for every change point of the block masks below)
    words = seg.runs_wire_words(cap, mnd.MAX_INSTANCES)
    merger = seg.Merger(H, W, 3, 2)
    try:
        wires = torch.zeros((3, 2, words), dtype=torch.int32, device="cuda")     # [rank, batch, words]
        masks = []
        for r in range(3):
            m = np.repeat(np.repeat(rng.integers(0, 4, (5, 6)), 8, 0), 12, 1)[:H, :W].astype(np.int32)
            masks.append(m)
            table = torch.full((H * W,), -1, dtype=torch.int32, device="cuda")
            table[:3] = torch.tensor([1 + r, 2 + r, 3 + r], dtype=torch.int32, device="cuda")
            seg.pack_runs(merger, torch.from_numpy(m).cuda(), table, 3, wires[r, 1], cap, mnd.MAX_INSTANCES, -1.0 * r)
        view = wires[:, 1]                                                         # strided rows
        got_m, got_t = seg.unpack_runs_batch(view, H, W, cap, mnd.MAX_INSTANCES)
        for r in range(3):
            assert np.array_equal(got_m[r].cpu().numpy(), masks[r])
            assert got_t[r, :3].cpu().tolist() == [1 + r, 2 + r, 3 + r] and bool((got_t[r, 3:] == -1).all())
            one_m, one_t = seg.unpack_runs(view[r].contiguous(), H, W, cap, mnd.MAX_INSTANCES)
            assert torch.equal(one_m, got_m[r]) and torch.equal(one_t, got_t[r])
        view[0, 0] = cap + 1000                                                    # damaged header
        got_m, got_t = seg.unpack_runs_batch(view, H, W, cap, mnd.MAX_INSTANCES)   # must not fault
        assert np.array_equal(got_m[1].cpu().numpy(), masks[1])
    finally:
        merger.close()
