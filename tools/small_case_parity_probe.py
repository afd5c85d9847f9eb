"""Actual errors of the small parity cases (the three tests of tests/test_gpu_parity.py
whose bars were looser than 1e-10) next to the conditioning of their reduced
normal equations: the data behind the per-case bars."""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from krylov_amd import host, problems  # noqa: E402
from oracle import oracle  # noqa: E402
import test_gpu_parity as tp  # noqa: E402


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


def kappa(ref):
    H = ref.solver().hessenberg_matrix()
    s = np.linalg.svd(H, compute_uv=False)
    return (s[0] / s[-1]) ** 2 if s[-1] > 0 else float("inf")


ctx = host.Context(0)
print("== timesteps_match_reference")
cases = [("timedep", problems.time_dependent, (12, 10), 8, False), ("timedep", problems.time_dependent, (12, 10), 8, True),
         ("rough", problems.rough_field, (18, 8), 40, False), ("siso", lambda a, b: problems.s_iso(a, b), (64, 48), 30, False),
         ("siso", lambda a, b: problems.s_iso(a, b), (33, 65), 20, True), ("tiny", problems.prob1, (1, 1), 2, False),
         ("tiny", problems.prob1, (2, 3), 4, True)]
for name, make, shape, m, tdp in cases:
    ref = oracle.ImplicitEuler(**make(*shape), m=m, tdp=tdp)
    dev = host.ImplicitEuler(ctx, **make(*shape), m=m, tdp=tdp)
    for step in range(4):
        dr, d = ref.timestep(0.25), dev.timestep(0.25)
        if not (math.isfinite(dr) and dr < 1e100):
            break
        print("%s %s m=%d tdp=%d step %d: kappa(HtH) %.2e  max_delta %.2e delta %.2e T %.2e" % (
            name, shape, m, tdp, step, kappa(ref), abs(d - dr) / abs(dr), rel(dev.delta(), ref.delta()),
            rel(dev.field(0)[1:-1, 1:-1], ref.field(0)[1:-1, 1:-1])), flush=True)
print("== recorded reference runs")
with open(os.path.join(ROOT, "tests", "golden", "prob1_reference_runs.json")) as f:
    runs = json.load(f)["runs"]
for idx in range(5):
    run = runs[idx]
    tdp = run["solver"] == "GMRESTDPSolver"
    dev = host.ImplicitEuler(ctx, **problems.prob1(run["cx"], run["cy"]), m=run["m"], tdp=tdp)
    ref = oracle.ImplicitEuler(**problems.prob1(run["cx"], run["cy"]), m=run["m"], tdp=tdp)
    for k in range(5):
        d = dev.timestep(run["dt"])
        ref.timestep(run["dt"])
        want = run["max_delta"][str(k)]
        print("run %d %s m=%d %dx%d step %d: kappa(HtH) %.2e  max_delta err %.2e" % (
            idx, run["solver"], run["m"], run["cx"], run["cy"], k, kappa(ref), abs(d - want) / abs(want)), flush=True)
print("== random configurations")
for seed in range(24):
    prob, dt = tp._random_problem(seed)
    ref = oracle.ImplicitEuler(**prob, m=6)
    dev = host.ImplicitEuler(ctx, **prob, m=6)
    worst, kmax = 0.0, 0.0
    for _ in range(3):
        d, dr = dev.timestep(dt), ref.timestep(dt)
        worst = max(worst, abs(d - dr) / max(abs(dr), 1e-300))
        kmax = max(kmax, kappa(ref))
    print("seed %d (%dx%d): kappa(HtH) %.2e  max_delta %.2e |dr| %.2e T %.2e" % (
        seed, prob["cx"], prob["cy"], kmax, worst, abs(dr), rel(dev.field(0)[1:-1, 1:-1], ref.field(0)[1:-1, 1:-1])), flush=True)
