"""Through the pybind11 module `krylov` (the reference's Python surface): cost of
sampling the Python boundary callables per step and the wall time of a step.
With the sampling hidden behind the device work a step costs about
max(device step, sampling) instead of their sum."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "krylov_amd"))
import krylov as K

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
h = 0.1
L = n * h
calls = [0]


def bc(val):
    def f(x, y, t):
        calls[0] += 1
        return val(x, y, t)
    return f


D, N = K.BCType.dirichlet, K.BCType.neumann
sim = K.ImplicitEulerGMRESSolver30(
    (0.0, 0.0), (L, L), n, n, lambda x, y: (0.0, 1.8 * (y / L) * (1.0 - y / L), 0.0),
    K.SecondOrderCentered(1.0, 25.0, 0.7, 0.1),
    (D, bc(lambda x, y, t: 0.0)), (D, bc(lambda x, y, t: 1.0)), (D, bc(lambda x, y, t: y / L)), (N, bc(lambda x, y, t: 0.0)),
    (D, bc(lambda x, y, t: 0.0)), (D, bc(lambda x, y, t: 0.0)), (N, bc(lambda x, y, t: 0.0)), (N, bc(lambda x, y, t: 0.0)),
    (D, bc(lambda x, y, t: 0.0)), (D, bc(lambda x, y, t: 0.0)), (D, bc(lambda x, y, t: 0.0)), (D, bc(lambda x, y, t: 0.0)))
sim.timestep(0.25)
calls[0] = 0
t0 = time.perf_counter()
steps = 10
for _ in range(steps):
    d = sim.timestep(0.25)
el = (time.perf_counter() - t0) / steps
per_step_calls = calls[0] // steps
# the sampling alone
fs = [lambda x, y, t: y / L] * 1
t0 = time.perf_counter()
for k in range(per_step_calls):
    fs[0](0.5, 0.5, 0.0)
pure = time.perf_counter() - t0
print("mesh %d^2 GMRES(30) through the module: %.2f ms per step, %d boundary callbacks per step "
      "(a bare Python loop over that many calls takes %.2f ms), last max delta %.6g" % (n, el * 1e3, per_step_calls, pure * 1e3, d))
