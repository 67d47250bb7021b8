"""Input tooling: the offset list contract and the generators (CPU)."""
import numpy as np
import pytest

from mergenet_amd import synth


def test_generate_offsets_literal_lists():
    """The two lists the reference's callers use, as measured from utils/train_utils.py:317-328
    (SURVEY.md section 8): generate_offsets(40, 10) (egs/cityscape/local/segment.py:77) and
    generate_offsets(80, 16) (the extended set of configs[4])."""
    assert synth.generate_offsets(40, 10) == [(1, 0), (0, 1), (-2, -1), (2, -3), (4, 3), (-6, 5),
                                              (-6, -10), (17, -6), (5, 26), (-40, 0)]
    assert synth.generate_offsets(80, 16) == [(1, 0), (0, 1), (-2, -1), (1, -2), (3, 2), (-3, 3),
                                              (-3, -5), (8, -3), (2, 11), (-15, 0), (4, -20), (26, 9),
                                              (-19, 32), (-39, -32), (52, -44), (46, 80)]


def test_offset_lists_with_negations_are_rejected():
    with pytest.raises(ValueError):
        synth.validate_offsets([(1, 0), (-1, 0)])
    with pytest.raises(ValueError):
        synth.validate_offsets([(0, 0)])
    synth.validate_offsets(synth.generate_offsets(40, 10))


def test_blurred_maps_are_not_sign_separable_but_deterministic():
    offs = synth.generate_offsets(40, 10)
    a = synth.blurred_v1(48, 64, 9, offs, 8000)
    b = synth.blurred_v1(48, 64, 9, offs, 8000)
    assert np.array_equal(a.sameness_probs, b.sameness_probs) and np.array_equal(a.class_probs, b.class_probs)
    inside = (a.sameness_probs > 0.4) & (a.sameness_probs < 0.6)
    assert inside.any()                      # values pass through 0.5 next to the boundaries
    assert a.sameness_probs.dtype == np.float32 and a.class_probs.shape == (9, 48, 64)


def test_checkerboard_cells_have_one_class_each():
    offs = synth.generate_offsets(6, 4)
    s = synth.checkerboard(24, 32, 3, offs, 4, 3)
    arg = s.class_probs.argmax(0)
    for cell in np.unique(s.instances):
        assert len(np.unique(arg[s.instances == cell])) == 1
