"""Prints digests of a few one-sweep solves (T after two steps, Hessenberg matrix); run it with
different KRY_SPLIT_FROM settings and compare the lines."""
import hashlib
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from krylov_amd import host, problems  # noqa: E402

ctx = host.Context(0)
for name, shape, m in (("s_iso", (300, 260), 50), ("s_iso", (40, 30), 35), ("s_iso", (5, 700), 50), ("prob1", None, 30),
                       ("s_iso", (700, 250), 24), ("s_iso", (33, 253), 14), ("s_iso", (1100, 520), 48)):
    cfg = problems.prob1() if name == "prob1" else getattr(problems, name)(*shape)
    for gen in (True, False):
        dev = host.ImplicitEuler(ctx, **cfg, m=m)
        dev.set_generated_bands(gen)
        d = [dev.timestep(0.25) for _ in range(2)]
        h = hashlib.sha1(dev.field(0).tobytes() + dev.solver().hessenberg_matrix().tobytes()).hexdigest()[:16]
        print(name, shape, m, gen, repr(d[-1]), h)
