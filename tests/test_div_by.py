"""The IOC kernels' neighbour search divides by the (fixed) window sizes through a reciprocal: y = RN(1 / b), q0 = RN(a y),
r = a - q0 b (one fma, exact), q = RN(q0 + r y)  (csrc/common.h: div_by / div_rn).  "Bit-exact neighbour-grid indexing" rests
on q == RN(a / b) for every a, so the restatement below is checked against IEEE fp32 division over EVERY significand of a in
several binades, for the window sizes the tests and the bench use -- and the divisors the device code excludes (significand all
ones, where the theorem behind it does not hold) are shown to be exactly the ones that fail."""
import numpy as np
import pytest

LD = np.longdouble          # 64-bit significand: a product of two fp32 values and its sum with a third are exact or correctly rounded


def _fma32(a, b, c):
    return (a.astype(LD) * b.astype(LD) + c.astype(LD)).astype(np.float32)


def div_rn_restated(a, b):
    b32 = np.float32(b)
    y = np.float32(1.0) / b32
    bb, yy = np.full_like(a, b32), np.full_like(a, y)
    q0 = (a * y).astype(np.float32)
    return _fma32(_fma32(-q0, bb, a), yy, q0)


def _fast(b):                # common.h: div_by().fast
    u = int(np.float32(b).view(np.uint32))
    e = (u >> 23) & 0xff
    return (u & 0x7fffff) != 0x7fffff and 64 < e < 190


def _all_significands(e):
    m = np.arange(0, 1 << 23, dtype=np.uint32)
    return (m | np.uint32((127 + e) << 23)).view(np.float32)


@pytest.mark.parametrize("b", [0.1, 0.25, 0.3, 0.04, 0.05, 0.08, 0.5, 0.6, 1.0 / 3.0, 32.0 / 1400.0, 0.12345])
def test_reciprocal_division_is_ieee_division(b):
    assert _fast(b)
    for e in (-9, -5, -3, -2, -1):
        a = _all_significands(e)
        a = a[a <= np.float32(2 * b)]                    # a = x_j - low lies in [0, window)
        if a.size == 0:
            continue
        np.testing.assert_array_equal(div_rn_restated(a, b), (a / np.float32(b)).astype(np.float32))


def test_the_excluded_divisors_are_the_ones_that_fail():
    for b in (np.float32(1.0) - np.float32(2.0 ** -24), np.nextafter(np.float32(2.0), np.float32(0.0))):
        assert not _fast(b)                              # the device code keeps the real division for these
        a = _all_significands(-1)
        assert (div_rn_restated(a, b) != (a / np.float32(b)).astype(np.float32)).any()
    assert not _fast(np.float32(1e-30)) and not _fast(np.float32(1e30))
