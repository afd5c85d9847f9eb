"""Problem definitions shared by the tests and the benchmark.

Each function returns the keyword arguments of an ``ImplicitEuler``
constructor (corner1, corner2, cx, cy, init, params, bcs); all inputs are
closed-form (no RNG, no files), so every process builds identical data.
"""
import numpy as np

from .capi import BOTTOM, TOP, LEFT, RIGHT, FIELD_T, FIELD_U, FIELD_V

PROB1_PARAMS = (1.0, 25.0, 0.7, 0.1)  # diffusion, Reynolds, Prandtl, Eckert


def _zero(x, y, t):
    return 0.0 * x


def prob1_bcs():
    """Boundary conditions of scripts/prob1.py:75-88."""
    return {
        (FIELD_T, BOTTOM): (True, _zero),
        (FIELD_T, TOP): (True, lambda x, y, t: 0.0 * x + 1.0),
        (FIELD_T, LEFT): (True, lambda x, y, t: y),
        (FIELD_T, RIGHT): (False, _zero),
        (FIELD_U, BOTTOM): (True, _zero),
        (FIELD_U, TOP): (True, _zero),
        (FIELD_U, LEFT): (False, _zero),
        (FIELD_U, RIGHT): (False, _zero),
        (FIELD_V, BOTTOM): (True, _zero),
        (FIELD_V, TOP): (True, _zero),
        (FIELD_V, LEFT): (True, _zero),
        (FIELD_V, RIGHT): (True, _zero),
    }


def prob1(cx=38, cy=18):
    """The channel-flow configuration of scripts/prob1.py:65-88 (dt = 0.25)."""
    avg_u = 3.0

    def init(x, y):
        return 0.0 * x * y, 6.0 * avg_u * y * (1.0 - y) + 0.0 * x, 0.0 * x * y

    return dict(corner1=(0.0, 0.0), corner2=(40.0, 1.0), cx=cx, cy=cy, init=init,
                params=PROB1_PARAMS, bcs=prob1_bcs())


def rough_field(cx=18, cy=8):
    """The rougher initial field of the reference's LU smoke runs
    (src/test.cpp:48-81): T = cos(pi x) sin(pi y), u = 3 y sin(pi x),
    v = 0.5 x cos(pi y); only three temperature conditions are non-default."""
    def init(x, y):
        return (1.0 * np.cos(np.pi * x) * np.sin(np.pi * y), 3.0 * y * np.sin(np.pi * x),
                0.5 * x * np.cos(np.pi * y))

    bcs = {
        (FIELD_T, BOTTOM): (True, _zero),
        (FIELD_T, TOP): (True, lambda x, y, t: 0.0 * x + 1.0),
        (FIELD_T, LEFT): (True, lambda x, y, t: y),
    }
    return dict(corner1=(0.0, 0.0), corner2=(40.0, 1.0), cx=cx, cy=cy, init=init,
                params=PROB1_PARAMS, bcs=bcs)


def s_iso(n, ny=None):
    """Synthetic isotropic benchmark mesh: n x ny cells of size h = 0.1, smooth
    closed-form fields, cell Peclet number below one.  With dt = 0.25 the
    system matrix is strongly diagonally dominant, so a fixed-m GMRES cycle is
    well conditioned at any mesh size."""
    ny = n if ny is None else ny
    h = 0.1
    Lx, Ly = n * h, ny * h

    def init(x, y):
        T = np.sin(2.0 * np.pi * x / Lx) * np.sin(np.pi * y / Ly)
        u = 0.45 * 4.0 * (y / Ly) * (1.0 - y / Ly) + 0.0 * x
        v = 0.1 * np.sin(2.0 * np.pi * x / Lx) + 0.0 * y
        return T, u, v

    bcs = prob1_bcs()
    bcs[(FIELD_T, LEFT)] = (True, lambda x, y, t: y / Ly)  # 0..1 across the inlet, as in prob1
    return dict(corner1=(0.0, 0.0), corner2=(Lx, Ly), cx=n, cy=ny, init=init,
                params=PROB1_PARAMS, bcs=bcs)


def s_aniso(n, ny=None):
    """prob1's 40 x 1 channel on an n x ny mesh: strongly anisotropic, high CFL
    along y, the case the line preconditioner is meant for."""
    return prob1(n, n if ny is None else ny)


def time_dependent(cx=12, cy=10):
    """A small case whose boundary values move in time (exercises the
    per-step boundary upload and the 'conditions are taken at the old time'
    rule of include/time_disc.hpp:287,292)."""
    def init(x, y):
        return np.sin(x) * np.cos(y), 0.3 * np.cos(x + y), 0.2 * np.sin(x - y)

    bcs = {
        (FIELD_T, BOTTOM): (True, lambda x, y, t: np.sin(x + t)),
        (FIELD_T, TOP): (False, lambda x, y, t: 0.1 * t + 0.0 * x),
        (FIELD_T, LEFT): (True, lambda x, y, t: y * (1.0 + t)),
        (FIELD_T, RIGHT): (False, lambda x, y, t: np.cos(y - t)),
        (FIELD_U, BOTTOM): (False, lambda x, y, t: 0.05 * x),
        (FIELD_U, TOP): (True, lambda x, y, t: 0.2 + 0.0 * x),
        (FIELD_U, LEFT): (True, lambda x, y, t: 0.1 * y),
        (FIELD_U, RIGHT): (False, _zero),
        (FIELD_V, BOTTOM): (True, lambda x, y, t: 0.01 * x * t),
        (FIELD_V, TOP): (False, lambda x, y, t: 0.02 + 0.0 * x),
        (FIELD_V, LEFT): (False, lambda x, y, t: 0.03 * y),
        (FIELD_V, RIGHT): (True, lambda x, y, t: 0.04 * y * t),
    }
    return dict(corner1=(2.0, 1.0), corner2=(0.0, -1.0), cx=cx, cy=cy, init=init,
                params=(0.8, 10.0, 0.9, 0.2), bcs=bcs)
