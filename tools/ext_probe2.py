"""First step of the restarted GMRES(m) with the line preconditioner on the S-aniso mesh at a given time step:
iterations and cycles per tolerance (used to pick the multi-restart case of the full-size parity test)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from krylov_amd import host, problems

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
m = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dt = float(sys.argv[3]) if len(sys.argv) > 3 else 0.01
ctx = host.Context(0)
for rtol in (1e-4, 1e-5, 1e-6, 1e-8):
    dev = host.ImplicitEuler(ctx, **problems.s_aniso(n), m=m, tdp=True, extended=(rtol, 20), bc_time_dependent=False)
    t = time.time()
    d = dev.timestep(dt)
    print("m %d dt %g rtol %g: max_delta %.6e stats %s  %.2f s" % (m, dt, rtol, d, dev.solver().last_stats(), time.time() - t), flush=True)
    del dev
