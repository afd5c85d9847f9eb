"""Wall time of one approximate-factorisation step (two tridiagonal sweeps) on a square S-iso mesh."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krylov_amd import host, problems
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
ctx = host.Context(0)
sim = host.ApproxFact(ctx, **problems.s_iso(n), bc_time_dependent=False)
sim.timestep(0.25)
ctx.set_kernel_timing(False)
t0 = time.perf_counter()
steps = 5
for _ in range(steps):
    d = sim.timestep(0.25)
el = (time.perf_counter() - t0) / steps
print("mesh %d^2: %.2f ms per approximate-factorisation step, last max delta %.6g" % (n, el * 1e3, d))
