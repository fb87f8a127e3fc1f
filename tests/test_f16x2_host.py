"""Host logic of precision "f16x2" (engine_f16x2.PlaneScales): the rule that places and moves the per-tensor powers of two.
No GPU: the scales are host floats handed to the kernels as arguments."""
import math

from cdml_amd import engine_f16x2 as h2


def test_pow2_for_places_a_maximum_below_its_top():
    for amax in (1e-9, 3.3e-4, 0.026, 1.0, 7.9, 6.0e4):
        s = h2.pow2_for(amax, 2.0 ** 11)
        assert s == 2.0 ** round(math.log2(s))                       # a power of two
        if 2.0 ** -30 < s < 2.0 ** 40:
            assert 2.0 ** 10 < amax * s <= 2.0 ** 11
    assert h2.pow2_for(1e-30, 2.0 ** 11) == 2.0 ** 40                # clamped: products of two scales stay finite in fp32
    assert h2.pow2_for(1e30, 2.0 ** 11) == 2.0 ** -30
    for bad in (0.0, -1.0, float("nan"), float("inf")):
        assert h2.pow2_for(bad, 2.0 ** 11) is None                   # nothing to place: the scale stays where it is


def test_scales_move_only_outside_their_window():
    s = h2.PlaneScales()
    assert s.x == 2.0 ** 14 and not s.calibrated
    assert s._move("w1", 0.03, s.TOP_MAX) and 2.0 ** 10 < 0.03 * s.w1 <= 2.0 ** 11      # calibration always places
    s.calibrated = True
    w = s.w1
    for amax in (0.03, 0.06, 0.1, 0.02, 0.03 / 200):                 # 4 x up, 200 x down: inside [top / 256, 4 top)
        assert not s._move("w1", amax, s.TOP_MAX) and s.w1 == w
    assert s.changes == 0
    assert s._move("w1", 0.2, s.TOP_MAX) and s.w1 < w and s.changes == 1                # 0.2 * w >= 4 top: re-placed
    assert 2.0 ** 10 < 0.2 * s.w1 <= 2.0 ** 11
    w = s.w1
    assert s._move("w1", 0.2 / 1000, s.TOP_MAX) and s.w1 > w and s.changes == 2          # fell out of the bottom
    assert not s._move("w1", 0.0, s.TOP_MAX) and not s._move("w1", float("nan"), s.TOP_MAX)
    # a tensor that sits at the top of its window is still 8 x (bounds) / 32 x (maxima) below fp16's largest value
    assert 4.0 * s.TOP_BOUND <= 65504.0 / 1.99 and 4.0 * s.TOP_MAX * 8 <= 65504.0 + 32


def test_check_steps():
    s = h2.PlaneScales(check_every=64)
    assert all(s.due(t) for t in range(200))                         # until calibrated: every step
    s.calibrated = True
    due = [t for t in range(300) if s.due(t)]
    assert due == [0, 1, 2, 4, 8, 16, 32, 64, 128, 192, 256]
    assert set(s.state()) == {"x", "w1", "w2", "h1", "dz2", "dz1"}
