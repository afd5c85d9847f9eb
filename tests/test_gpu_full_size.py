"""Parity at BASELINE.json's full sizes: one implicit-Euler step of the
DEFAULT HIP path (through the C ABI) against the single-threaded CPU
restatement on the same synthetic input.

  config 2   2048 x 2048, GMRES(30)                     CPU about 5 s
  config 3   8192 x 8192, GMRES(50)                     CPU about 3 min, 35 GB of host memory
  config 5   8192 x 8192 S-aniso, line preconditioner   CPU about 1 min (GMRES(10), as scripts/prob1.py uses it)

Bars (fixed-m cycle of include/solver.hpp:124-161, step of
include/time_disc.hpp:277-289): relative 2-norm error of the solution vector
delta and of the temperature field <= 1e-10, max|delta| to 1e-10 relative, and
the first six columns of the Hessenberg matrix to 1e-10 of its largest entry.
Later Hessenberg columns are sensitive to the rounding of the dot products
(each column's deviation is about 1.5 times the previous one's): the CPU
reference sums 67 million products one after the other, the device in a tree,
and with the reference's OWN algorithm on the device (the plain modified
Gram-Schmidt chain, one link per sweep) the late columns differ from the CPU
by just as much.  So every column of the default path (one-sweep Arnoldi
steps) is held to the distance of that chain run from the CPU reference (x 2),
which shows that the drift comes from the summation order, not from the
formulation.

The CPU runs are long and silent, so a heartbeat thread reports progress
while they run (to the uncaptured stdout and to gpurun_out/heartbeat.log).
"""
import contextlib
import os
import threading
import time

import numpy as np
import pytest

from krylov_amd import host, problems

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RTOL = 1e-10


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


@contextlib.contextmanager
def heartbeat(label, capfd, every=20.0):
    """Report that the process is alive while a long CPU computation runs."""
    stop = threading.Event()
    t0 = time.time()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    path = os.path.join(ROOT, "gpurun_out", "heartbeat.log")

    def beat():
        while not stop.wait(every):
            line = "[%s: CPU reference running, %.0f s]" % (label, time.time() - t0)
            with open(path, "a") as f:
                f.write(line + "\n")
            print(line, flush=True)

    with capfd.disabled():
        th = threading.Thread(target=beat, daemon=True)
        th.start()
        try:
            yield
        finally:
            stop.set()
            th.join()


def one_step_errors(ctx, oracle, capfd, label, prob, m, tdp, dt=0.25, steps=1, variants=({},)):
    """Run `steps` implicit steps on the CPU restatement (in a worker thread;
    the call releases the GIL) and on the device (once per entry of
    `variants`: extra constructor arguments), and return the errors of each."""
    ref = oracle.ImplicitEuler(**prob, m=m, tdp=tdp)
    out = {}

    def cpu():
        out["d_ref"] = [ref.timestep(dt) for _ in range(steps)]

    th = threading.Thread(target=cpu)
    runs = []
    with heartbeat(label, capfd):
        th.start()
        for kw in variants:
            dev = host.ImplicitEuler(ctx, **prob, m=m, tdp=tdp, bc_time_dependent=False, **kw)
            d = [dev.timestep(dt) for _ in range(steps)]
            runs.append((d, dev.delta(), dev.field(0)[1:-1, 1:-1].copy(), dev.solver().hessenberg_matrix()))
            del dev
        th.join()
    dr = out["d_ref"]
    Href = ref.solver().hessenberg_matrix()
    dref, Tref = ref.delta(), ref.field(0)[1:-1, 1:-1].copy()
    del ref
    results = []
    for d, delta, T, H in runs:
        results.append({
            "max_delta": max(abs(a - b) / abs(b) for a, b in zip(d, dr)),
            "delta": rel(delta, dref),
            "T": rel(T, Tref),
            "H_cols": np.max(np.abs(H - Href), axis=0) / np.max(np.abs(Href)),
            "d": d, "d_ref": dr,
        })
    return results


def report(capfd, label, res):
    with capfd.disabled():
        print("\n[%s] rel. errors vs CPU: max|delta| %.2e, delta %.2e, T %.2e, H worst column %.2e (column %d)"
              % (label, res["max_delta"], res["delta"], res["T"], float(np.max(res["H_cols"])),
                 int(np.argmax(res["H_cols"]))), flush=True)


CHAIN = {"one_sweep": -1, "mgs_block": 1}  # the reference's Gram-Schmidt chain, one link per sweep


def check(res, chain=None, h_cols=6):
    assert res["max_delta"] <= RTOL
    assert res["delta"] <= RTOL
    assert res["T"] <= RTOL
    assert float(np.max(res["H_cols"][:h_cols])) <= RTOL
    if chain is not None:
        assert np.all(res["H_cols"] <= 2.0 * chain["H_cols"] + 1e-12)


def test_config2_2048_gmres30_one_step_vs_cpu(ctx, oracle, capfd):
    """BASELINE config 2 at full size, default path (one-sweep Arnoldi steps)."""
    res, chain = one_step_errors(ctx, oracle, capfd, "2048^2 GMRES(30)", problems.s_iso(2048), 30, False,
                                 variants=({}, CHAIN))
    report(capfd, "config 2: 2048^2, GMRES(30)", res)
    report(capfd, "config 2: 2048^2, GMRES(30), plain chain", chain)
    check(res, chain)


def test_config3_8192_gmres50_one_step_vs_cpu(ctx, oracle, capfd):
    """BASELINE config 3 at full size on one GPU, default path: the headline
    workload of bench.py against the CPU restatement."""
    res, chain = one_step_errors(ctx, oracle, capfd, "8192^2 GMRES(50)", problems.s_iso(8192), 50, False,
                                 variants=({}, CHAIN))
    report(capfd, "config 3: 8192^2, GMRES(50)", res)
    report(capfd, "config 3: 8192^2, GMRES(50), plain chain", chain)
    check(res, chain)


def test_config5_8192_s_aniso_line_preconditioner_vs_cpu(ctx, oracle, capfd):
    """BASELINE config 5 at full size, bug-compatible (the reference's
    GMRESTDPSolver never applies M^-1 to the combination,
    include/solver.hpp:290-293, so the step diverges as scripts/prob1.py's
    default does): three steps of GMRESTDPSolver<10>.  With the line
    preconditioner A M^-1 is close to the identity along the lines: the Krylov
    space captures the right-hand side within two steps, after which
    Gram-Schmidt orthogonalises rounding noise (H(2,1) is ~1e-9 of H(0,0)) in
    the reference as on the device, so only the first Hessenberg column is
    compared; the solution, the field and max|delta| carry the parity bar."""
    res, = one_step_errors(ctx, oracle, capfd, "8192^2 S-aniso TDP GMRES(10)", problems.s_aniso(8192), 10, True,
                           steps=3)
    report(capfd, "config 5: 8192^2 S-aniso, line preconditioner, GMRES(10), 3 steps", res)
    assert all(np.isfinite(res["d_ref"]))
    check(res, h_cols=1)
