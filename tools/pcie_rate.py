"""PCIe-inclusive rate of the stand-alone solver entry point: kry_gmres_solve
takes the right-hand side from host memory and returns the result to host
memory (the benchmark's step loop keeps everything resident instead)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from krylov_amd import host, problems

n_side, m = int(sys.argv[1]) if len(sys.argv) > 1 else 8192, 50
ctx = host.Context(0)
dev = host.ImplicitEuler(ctx, **problems.s_iso(n_side), m=m, bc_time_dependent=False)
dev.assemble_system(0.25)
g, op = dev.solver(), dev.operator()
b = np.random.default_rng(1).uniform(-1, 1, (n_side + 2) ** 2)
b.reshape(n_side + 2, n_side + 2)[[0, 0, -1, -1], [0, -1, 0, -1]] = 0.0
g.solve(op, b)
t0 = time.perf_counter(); g.solve(op, b); t_host = time.perf_counter() - t0
g.solve_device(op, dev.rhs_device(), None); ctx.synchronize()
t0 = time.perf_counter(); g.solve_device(op, dev.rhs_device(), None); ctx.synchronize(); t_dev = time.perf_counter() - t0
print("mesh %d m %d: host-buffer solve %.1f ms (%.1f it/s), resident solve %.1f ms (%.1f it/s), transfer share %.1f %%"
      % (n_side, m, t_host * 1e3, m / t_host, t_dev * 1e3, m / t_dev, 100 * (t_host - t_dev) / t_host))
