"""Division by a divisor that many divisions share, against the division itself.

The reference forms the off-diagonal entries of the implicit-Euler operator
with four divisions by the mesh widths per cell
(SecondOrderCentered::Dx_m1 / Dx_p1 / Dy_m1 / Dy_p1,
include/space_disc.hpp:83-113).  The kernels that evaluate those entries on
the fly take a correctly rounded reciprocal of each width once and then five
multiply-adds per quotient, which yields the correctly rounded quotient inside
a guarded range; outside it (and for divisors outside theirs) they divide.
Every quotient must have the bits of the division."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ONES = float(np.float64(2.0) - np.finfo(np.float64).eps)      # significand of all ones
DIVISORS = [
    (40.0 / 8192, 1.0 / 8192),          # the headline meshes' widths
    (1.0 / 3.0, 1e-3),
    (ONES, -ONES * 2.0 ** -70),
    (np.nextafter(1.0, 2.0), np.nextafter(1.0, 0.0)),
    (3.0, -7.0),
    (2.0 ** -100, 2.0 ** 100 * ONES),   # the ends of the divisors' range
    (0.1, 1.2345678901234567e-5),
    (np.pi, np.e * 1e7),
]


@pytest.mark.parametrize("mode", [0, 3])
@pytest.mark.parametrize("dx,dy", DIVISORS)
def test_quotients_have_the_bits_of_the_division(ctx, dx, dy, mode):
    count = 1 << 29
    bad, fast = ctx.selftest_shared_division(dx, dy, 12345 + mode, count, mode)
    assert bad == 0
    assert fast == count  # every group took the five multiply-adds


@pytest.mark.parametrize("dx,dy", DIVISORS[:4])
def test_guard_sends_everything_else_to_the_division(ctx, dx, dy):
    count = 1 << 26
    # arbitrary bit patterns: zeros, subnormals, infinities, NaNs, huge and tiny numbers
    bad, fast = ctx.selftest_shared_division(dx, dy, 777, count, 1)
    assert bad == 0
    assert fast < count // 100  # nearly every wavefront holds a lane outside the range
    # quotients around 2^-800 and 2^800: both sides of the guard
    bad, fast = ctx.selftest_shared_division(dx, dy, 778, count, 2)
    assert bad == 0


@pytest.mark.parametrize("dx,dy", [(2.0 ** -101, 1.0), (1.0, 2.0 ** 101), (0.0, 1.0), (1.0, float("inf")),
                                   (float("nan"), 1.0), (5e-324, 1.0), (1.0, 1e-310)])
def test_divisors_outside_their_range_always_divide(ctx, dx, dy):
    bad, fast = ctx.selftest_shared_division(dx, dy, 99, 1 << 22, 0)
    assert bad == 0
    assert fast == 0
