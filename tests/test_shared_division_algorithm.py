"""The division by a shared divisor, restated in exact rational arithmetic.

The device evaluates the quotients in the band entries
(SecondOrderCentered::Dx_m1 / Dx_p1 / Dy_m1 / Dy_p1, include/space_disc.hpp:83-113)
as q0 = RN(a y), q1 = RN(q0 + RN(a - d q0) y), q = RN(q1 + RN(a - d q1) y) with
y = RN(1 / d) and fused residuals (div_shared_fast in
krylov_amd/csrc/kry_device.h).  Here every fused operation is one rounding of an
exact rational value, so this checks the algorithm itself, without a GPU:
q must equal the correctly rounded a / d (which is what Python's float division
returns) for numerators inside the guarded range."""
import random
from fractions import Fraction

import numpy as np


def fma(a, b, c):
    """RN(a b + c): one rounding of the exact value."""
    return float(Fraction(a) * Fraction(b) + Fraction(c))


def div_shared_fast(a, d, y):
    q = a * y
    r = fma(-d, q, a)
    q = fma(r, y, q)
    r = fma(-d, q, a)
    return fma(r, y, q)


def numerators(rng, d, count):
    """Uniform mantissas over +-40 binades, and quotients next to powers of two and to midpoints."""
    for _ in range(count):
        yield rng.choice((-1.0, 1.0)) * rng.uniform(1.0, 2.0) * 2.0 ** rng.randint(-40, 40)
    for _ in range(count):
        q = 2.0 ** rng.randint(-30, 30) if rng.random() < 0.5 else rng.uniform(1.0, 2.0) * 2.0 ** rng.randint(-30, 30)
        for _ in range(rng.randint(0, 4)):
            q = float(np.nextafter(q, rng.choice((0.0, np.inf))))
        if rng.random() < 0.5:  # half a unit in the last place further: next to a midpoint
            q_mid = Fraction(q) + (Fraction(float(np.nextafter(q, np.inf))) - Fraction(q)) / 2
            yield float(q_mid * Fraction(d))
        else:
            yield q * d


def test_two_corrections_give_the_correctly_rounded_quotient():
    rng = random.Random(2024)
    ones = 2.0 - 2.0 ** -52
    divisors = [40.0 / 8192, 1.0 / 8192, 1.0 / 3.0, 1e-3, ones, -ones * 2.0 ** -70, float(np.nextafter(1.0, 2.0)),
                float(np.nextafter(1.0, 0.0)), 3.0, -7.0, 2.0 ** -100, 2.0 ** 100 * ones, 0.1, np.pi]
    divisors += [rng.choice((-1.0, 1.0)) * rng.uniform(1.0, 2.0) * 2.0 ** rng.randint(-100, 100) for _ in range(40)]
    checked = 0
    for d in divisors:
        d = float(d)
        y = 1.0 / d
        for a in numerators(rng, d, 400):
            assert div_shared_fast(a, d, y) == a / d, (a, d)
            checked += 1
    assert checked == len(divisors) * 800


def test_one_correction_is_not_enough():
    """The reason for the second correction: after one, the quotient can still be the wrong neighbour."""
    rng = random.Random(7)
    wrong_q0 = wrong_q1 = 0
    for _ in range(20000):
        d = rng.uniform(1.0, 2.0)
        a = rng.uniform(1.0, 2.0)
        y = 1.0 / d
        q0 = a * y
        q1 = fma(fma(-d, q0, a), y, q0)
        wrong_q0 += q0 != a / d
        wrong_q1 += q1 != a / d
    assert wrong_q0 > 0          # the plain product with the reciprocal is off by a unit now and then
    assert wrong_q1 <= wrong_q0  # one correction repairs (nearly) all of them; the second is the proven one
