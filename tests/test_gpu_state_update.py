"""The state update of ImplicitEuler::timestep (include/time_disc.hpp:281-286)
on its own: T += delta on interior cells and the returned maximum.

The reference folds `max_delta = std::max(std::abs(delta), max_delta)` over the
interior cells in (i, j) order.  std::max(a, b) is `(a < b) ? b : a`; with a NaN
on either side the comparison is false, so a NaN |delta| replaces the running
maximum and the next finite |delta| replaces a NaN maximum: the value is the
largest |delta| BEHIND the last NaN cell, NaN only if the last cell is NaN.
The device (a parallel reduction) must return exactly that value.
"""
import math

import numpy as np
import pytest

from krylov_amd import host, problems

pytestmark = pytest.mark.gpu


def reference_fold(delta_interior):
    """include/time_disc.hpp:279-286, literally."""
    m = 0.0
    for a in np.abs(delta_interior).reshape(-1):
        m = m if a < m else a  # std::max(a, m)
    return float(m)


def same(a, b):
    return a == b or (math.isnan(a) and math.isnan(b))


@pytest.mark.parametrize("cx,cy", [(37, 23), (300, 70), (1, 1), (2, 129), (1500, 3)])
def test_apply_delta_matches_the_reference_fold(ctx, cx, cy):
    rng = np.random.default_rng(cx * 1000 + cy)
    dev = host.ImplicitEuler(ctx, **problems.s_iso(cx, cy), m=3)
    nint = cx * cy
    cases = [[], [0], [nint - 1], [nint // 2], [nint // 2, nint - 1], [0, nint // 3], [cy - 1], [max(nint - 2, 0)],
             sorted(set(rng.integers(0, nint, max(1, nint // 100)).tolist()))]
    for nan_at in cases:
        d = np.zeros((cx + 2, cy + 2))
        d[:, :] = rng.uniform(-3.0, 3.0, d.shape)
        inner = d[1:-1, 1:-1]
        for p in nan_at:
            inner[p // cy, p % cy] = np.nan
        # a large value in front of the last NaN must not survive, ghost entries must be ignored
        if nan_at and nan_at[-1] > 0:
            inner[0, 0] = 1e30
        d[0, :] = np.nan
        d[:, -1] = 1e300
        T0 = dev.field(0)
        got = dev.apply_delta(d)
        want = reference_fold(inner)
        assert same(got, want), (nan_at[:5], got, want)
        T1 = dev.field(0)
        assert np.array_equal(T1[1:-1, 1:-1], T0[1:-1, 1:-1] + inner, equal_nan=True)
        # ghost cells and corners untouched
        mask = np.ones(T0.shape, bool)
        mask[1:-1, 1:-1] = False
        assert np.array_equal(T1[mask], T0[mask], equal_nan=True)
        # keep the field finite for the next case
        dev = host.ImplicitEuler(ctx, **problems.s_iso(cx, cy), m=3) if nan_at else dev
    assert dev.cur_time() == 0.0


def test_infinite_entries_are_ordinary_values(ctx):
    cx, cy = 20, 12
    dev = host.ImplicitEuler(ctx, **problems.s_iso(cx, cy), m=3)
    d = np.zeros((cx + 2, cy + 2))
    d[5, 5] = -np.inf
    assert dev.apply_delta(d) == np.inf
    d[7, 7] = np.nan
    d[9, 9] = 2.0
    assert dev.apply_delta(d) == 2.0
