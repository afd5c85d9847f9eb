"""Exploration: errors of several device variants against ONE CPU run at full size.
usage: python tools/full_size_probe.py MESH M [tdp]"""
import contextlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import test_gpu_full_size as F  # noqa: E402
from krylov_amd import host, problems  # noqa: E402
from oracle import oracle  # noqa: E402


class Cap:
    @contextlib.contextmanager
    def disabled(self):
        yield


n, m = int(sys.argv[1]), int(sys.argv[2])
tdp = len(sys.argv) > 3
oracle.build()
ctx = host.Context(0)
variants = [{}] if tdp else [{}, {"one_sweep": -1, "mgs_block": 1}, {"one_sweep": -1, "mgs_block": 32}]
prob = problems.s_aniso(n) if tdp else problems.s_iso(n)
res = F.one_step_errors(ctx, oracle, Cap(), "%d^2 m=%d" % (n, m), prob, m, tdp, steps=3 if tdp else 1, variants=variants)
for kw, r in zip(variants, res):
    print(kw, "max|delta| %.3e delta %.3e T %.3e" % (r["max_delta"], r["delta"], r["T"]))
    print("   H col errs:", " ".join("%.1e" % v for v in r["H_cols"]))
    print("   d", r["d"], "d_ref", r["d_ref"])
