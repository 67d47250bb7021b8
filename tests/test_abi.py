"""The C-ABI library loads and exports every symbol include/mergenet_hip.h declares (no GPU)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "mergenet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\(", text)
    return sorted({n for n in names if n.startswith("mn_") or n == "c_run_segmentation"})


def test_header_declares_the_reference_entry_point():
    assert "c_run_segmentation" in declared_functions()


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from mergenet_amd import segmenter as seg
    lib = seg.load_library()
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, "not exported: %s" % missing
    assert set(declared_functions()) == set(seg.EXPORTS)


def test_option_struct_layout_matches_header():
    import ctypes
    from mergenet_amd import segmenter as seg
    o = seg.default_options()
    assert abs(o.merge_logprob_bias - 0.03) < 1e-7 and o.object_merge_factor == 1.0
    assert o.variant == seg.MN_VARIANT_CSEGMENT and o.prune_threshold == 200.0
    assert ctypes.sizeof(seg.MnOptions) == 17 * 4      # + tie_order
    assert ctypes.sizeof(seg.MnStats) == 10 * 4 + 2 * 8 + 8 + 10 * 4 + 6 * 4   # + proof, cores_condemned, tied_steps, tied_merges, tie_order_used, reserved


def test_binding_rejects_bad_buffers_like_the_reference_binding():
    import numpy as np
    from mergenet_amd import segmenter as seg
    good = np.zeros((2, 4, 4), np.float32)
    with pytest.raises(ValueError):
        seg.run_segmentation(good.astype(np.float64), good, 2, [(0, 1), (1, 0)], 0.0, 1.0, 0.0)
    with pytest.raises(ValueError):
        seg.run_segmentation(good[:, :, ::2], good, 2, [(0, 1), (1, 0)], 0.0, 1.0, 0.0)
    with pytest.raises(TypeError):
        seg.run_segmentation(None, good, 2, [(0, 1), (1, 0)], 0.0, 1.0, 0.0)
    with pytest.raises(AssertionError):
        seg.ObjectSegmenter(good, np.zeros((3, 4, 4), np.float32), 2, [(0, 1), (1, 0)])
