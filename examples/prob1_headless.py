#!/usr/bin/env python3
"""Headless replay of the reference driver script (scripts/prob1.py:65-98):
same imports, same configuration call, same time loop, without the plots.

    python examples/prob1_headless.py [solver-class-name]

The default solver of the reference script, ImplicitEulerGMRESTDPSolver10,
diverges (its preconditioned solve returns u instead of M^-1 u) and ends on
NaN exactly as the reference does; pass ImplicitEulerGMRESSolver10 for the
converging run (147 steps).
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "krylov_amd"))

import krylov  # noqa: E402
from krylov import SecondOrderCentered, BCType  # noqa: E402


def def_configuration(timedisc):
    diffusion = 1.0
    reynolds = 25.0
    prandtl = 0.7
    eckert = 0.1
    avg_u = 3.0
    return timedisc((0.0, 0.0), (40.0, 1.0), 38, 18,
                    lambda x, y: (0.0, 6.0 * avg_u * y * (1.0 - y), 0.0),
                    SecondOrderCentered(diffusion, reynolds, prandtl, eckert),
                    (BCType.dirichlet, lambda x, y, t: 0.0),  # Temperature bottom
                    (BCType.dirichlet, lambda x, y, t: 1.0),  # Temperature top
                    (BCType.dirichlet, lambda x, y, t: y),    # Temperature left
                    (BCType.neumann, lambda x, y, t: 0.0),    # Temperature right
                    (BCType.dirichlet, lambda x, y, t: 0.0),  # Vel u bottom
                    (BCType.dirichlet, lambda x, y, t: 0.0),  # Vel u top
                    (BCType.neumann, lambda x, y, t: 0.0),    # Vel u left
                    (BCType.neumann, lambda x, y, t: 0.0),    # Vel u right
                    (BCType.dirichlet, lambda x, y, t: 0.0),  # Vel v bottom
                    (BCType.dirichlet, lambda x, y, t: 0.0),  # Vel v top
                    (BCType.dirichlet, lambda x, y, t: 0.0),  # Vel v left
                    (BCType.dirichlet, lambda x, y, t: 0.0))  # Vel v right


def run(name="ImplicitEulerGMRESTDPSolver10", verbose=True, max_steps=1000):
    solver = def_configuration(getattr(krylov, name))
    max_delta = float("inf")
    history = []
    i = 0
    while max_delta > 1e-9 and i < max_steps:
        if verbose:
            print("Timestep {}, delta: {}".format(i, max_delta))
        max_delta = solver.timestep(0.25)
        history.append(max_delta)
        i += 1
    return solver, history


if __name__ == "__main__":
    s, h = run(sys.argv[1] if len(sys.argv) > 1 else "ImplicitEulerGMRESTDPSolver10")
    T = s.mesh().Temp_array()
    print("steps: {}, last delta: {}, t = {}, sum of interior T = {!r}".format(
        len(h), h[-1], s.cur_time(), float(T[1:-1, 1:-1].sum())))
