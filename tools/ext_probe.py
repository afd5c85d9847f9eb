"""Residual reached by the restarted solver with the line preconditioner on the
S-aniso mesh for a list of tolerances (at most two cycles each): used to pick
the tolerance of the full-size extended-mode parity test."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krylov_amd import host, problems

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ctx = host.Context(0)
for mm, rtol, cyc in ((m, 1e-12, 1), (m, 1e-12, 2), (m, 2.5e-4, 3), (m, 2e-4, 3), (m, 1.5e-4, 3), (m, 1e-4, 3),
                      (3, 3e-4, 4), (4, 3e-4, 4), (3, 2.5e-4, 6), (4, 2e-4, 8), (10, 2e-4, 4), (10, 1e-4, 6)):
    dev = host.ImplicitEuler(ctx, **problems.s_aniso(n), m=mm, tdp=True, fixed_preconditioning=True,
                             extended=(rtol, cyc), bc_time_dependent=False)
    t = time.time()
    d = dev.timestep(0.25)
    print("m %d rtol %g max_cycles %d: max_delta %.6e stats %s  %.2f s" % (mm, rtol, cyc, d, dev.solver().last_stats(), time.time() - t), flush=True)
    del dev
