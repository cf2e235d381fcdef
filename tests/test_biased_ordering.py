"""The arithmetic fact the packed-int16 cells rest on (sw_kernels.hip, "biased int16"): for 16-bit patterns in
[1024, 0x7C00) -- the normal, finite, positive fp16 numbers -- the fp16 ordering is the integer ordering, so gfx950's
v_pk_maximum3_f16 can serve as an integer max3; below 1024 sit the denormals, from 0x7C00 on inf / NaN."""
import numpy as np


def test_fp16_order_equals_integer_order_on_the_biased_range():
    pat = np.arange(1024, 0x7C00, dtype=np.uint16)
    f = pat.view(np.float16).astype(np.float64)
    assert np.isfinite(f).all() and (f > 0).all()
    assert (np.diff(f) > 0).all()                       # strictly increasing with the bit pattern
    tiny = np.finfo(np.float16).tiny                    # smallest normal number = pattern 1024
    assert np.array([1024], np.uint16).view(np.float16)[0] == tiny
    below = np.arange(1, 1024, dtype=np.uint16).view(np.float16).astype(np.float64)
    assert (below < float(tiny)).all() and (below > 0).all()   # denormals: smaller than every biased value even if kept
    assert np.isinf(np.array([0x7C00], np.uint16).view(np.float16)[0])
    assert np.isnan(np.array([0x7C01, 0x7E00, 0xFFFB], np.uint16).view(np.float16)).all()   # incl. "-5" as int16


def test_thresholds_leave_room_for_the_largest_sum():
    # plain biased cell: unflagged scores < 31600 (biased); a diagonal sum adds at most 127
    assert 31599 + 127 < 0x7C00
    # column-frame cell: true < 22256, bias 1024, frame offset <= 8192, sum adds at most 127 + ge with ge <= 64
    assert 22255 + 1024 + 8192 + 127 + 64 < 0x7C00
