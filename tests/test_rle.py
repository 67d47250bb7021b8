"""COCO RLE: host format code (CPU) and the one-pass device extraction (GPU)."""
import numpy as np
import pytest

from mergenet_amd import rle


def test_known_answers_of_the_published_format():
    # all-ones 2x2 mask: runs [0 zeros, 4 ones]; small counts map to single characters '0'+c
    assert rle.counts_to_string([0, 4]) == b"04"
    assert rle.counts_to_string([5]) == b"5"
    # 5-bit groups, low group first, 0x20 = continuation: 40 = 0b01000 | (1 << 5)
    assert rle.counts_to_string([40]) == bytes([8 + 0x20 + 48, 1 + 48])
    # from the third count on the difference to the count two places back is stored
    assert rle.string_to_counts(rle.counts_to_string([3, 9, 2, 9, 1])) == [3, 9, 2, 9, 1]


def test_round_trip_random_masks():
    rng = np.random.default_rng(1)
    for _ in range(100):
        H, W = rng.integers(1, 20, 2)
        b = (rng.random((H, W)) < 0.4).astype(np.uint8)
        c = rle.binary_mask_counts(b)
        assert sum(c) == H * W
        assert rle.string_to_counts(rle.counts_to_string(c)) == c
        assert np.array_equal(rle.decode(c, H, W), b)


def test_change_point_grouping_equals_per_instance_encoding():
    rng = np.random.default_rng(2)
    for _ in range(50):
        H, W = rng.integers(1, 12, 2)
        K = 4
        m = rng.integers(0, K + 1, (H, W)).astype(np.int32)
        flat = m.reshape(-1, order="F")
        prev = np.concatenate([[0], flat[:-1]])
        j = np.flatnonzero(flat != prev)
        res = rle.from_change_points(j, prev[j], flat[j], H, W, K)
        for k in range(1, K + 1):
            assert res[k - 1]["counts"] == rle.counts_to_string(rle.binary_mask_counts(m == k))
            assert res[k - 1]["size"] == [H, W]


def test_native_host_encoder_equals_the_python_one():
    """mn_rle_encode_host (C, in libmergenet_hip.so; needs no GPU) against rle.from_change_points:
    same strings, plus the pixel count per instance used for the zero-area drop
    (egs/cityscape/local/evaluate.py:52-54)."""
    import ctypes
    from mergenet_amd import segmenter as seg
    lib = seg.load_library()
    rng = np.random.default_rng(4)
    for it in range(120):
        H, W = (int(x) for x in rng.integers(1, 16, 2))
        K = 6
        m = rng.integers(0, K + 1, (H, W)).astype(np.int32)
        if it % 3 == 0:
            m[m == 2] = 0                                   # an instance without pixels
        flat = m.reshape(-1, order="F")
        prev = np.concatenate([[0], flat[:-1]])
        j = np.flatnonzero(flat != prev)
        n, cap = len(j), len(j) + 5
        pts = np.zeros((3, cap), np.int32)
        pts[0, :n], pts[1, :n], pts[2, :n] = j, prev[j], flat[j]
        offs = (ctypes.c_longlong * (K + 1))()
        areas = (ctypes.c_int * K)()
        out = ctypes.create_string_buffer(8 * n + 16 * K + 64)
        need = lib.mn_rle_encode_host(pts.ctypes.data_as(ctypes.POINTER(ctypes.c_int)), cap, n, H, W, K,
                                      ctypes.cast(out, ctypes.c_void_p), len(out), offs, areas)
        assert 0 < need <= len(out)
        ref = rle.from_change_points(j, prev[j], flat[j], H, W, K)
        for k in range(1, K + 1):
            assert out.raw[offs[k - 1]:offs[k]] == ref[k - 1]["counts"]
            assert areas[k - 1] == int((m == k).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(7, 5), (64, 96), (250, 333)])
def test_device_rle_equals_per_instance_host_encoding(shape):
    import torch
    from mergenet_amd import segmenter as seg
    H, W = shape
    rng = np.random.default_rng(3)
    K = 9
    # blobs rather than noise: realistic run lengths
    m = np.zeros((H, W), np.int32)
    for k in range(1, K + 1):
        y, x = rng.integers(0, H), rng.integers(0, W)
        m[max(0, y - H // 6):y + H // 6 + 1, max(0, x - W // 6):x + W // 6 + 1] = k
    merger = seg.Merger(H, W, 4, 4)
    res = merger.encode_rle(torch.from_numpy(m).cuda(), K)
    assert len(res) == K and [e["label"] for e in res] == list(range(1, K + 1))
    kept = merger.encode_rle(torch.from_numpy(m).cuda(), K, drop_zero_area=True)
    assert [e["label"] for e in kept] == [k for k in range(1, K + 1) if (m == k).any()]
    assert all(e["area"] == int((m == e["label"]).sum()) for e in res)
    for k in range(1, K + 1):
        assert res[k - 1]["counts"] == rle.counts_to_string(rle.binary_mask_counts(m == k))
        assert np.array_equal(rle.decode(rle.string_to_counts(res[k - 1]["counts"]), H, W), m == k)
    merger.close()
