"""ImplicitEulerApproxFact (include/time_disc.hpp:388-540).

Pins of the CPU restatement (CPU tests): the reference's own tests of this
class only print (src/test.cpp:591-631) or hold a stale expectation for the
interior diagonal (:649-654, it fails against the reference itself), so what
is used is (a) the boundary rows that :643-648 does assert, and (b) the steady
state: the time loop converges to the fixed point of the same spatial
discretisation as the GMRES runs, whose recorded reference value of the summed
temperature it must reproduce.  The transient itself has no reference
fixture: parity of the transient is between the device path and this
restatement only (bit-exact, GPU tests)."""
import json
import os

import numpy as np
import pytest

from krylov_amd import capi, problems

HERE = os.path.dirname(os.path.abspath(__file__))


def af_default_problem(cx, cy):
    """default_ie_af_solver, src/test.cpp:558-589: three temperature conditions, the rest default."""
    p = problems.prob1(cx, cy)
    p["bcs"] = {
        (capi.FIELD_T, capi.BOTTOM): (True, lambda x, y, t: 0.0 * x),
        (capi.FIELD_T, capi.TOP): (True, lambda x, y, t: 0.0 * x + 1.0),
        (capi.FIELD_T, capi.LEFT): (True, lambda x, y, t: y),
    }
    return p


def test_assemble_sys_boundary_rows(oracle):
    """TEST(assemble_sys, approximate_factorization), src/test.cpp:633-648."""
    sim = oracle.ApproxFact(**af_default_problem(20, 10))
    dx = sim.assemble_system_dx(0.25, 1)
    assert dx.shape == (22, 3)
    assert np.isnan(dx[0, 0]) and dx[0, 1] == 1.0 and dx[0, 2] == 1.0
    assert dx[-1, 0] == 1.0 and dx[-1, 1] == 1.0 and np.isnan(dx[-1, 2])
    # interior rows: Dx_m1 dt, Dx_0 dt + 1, Dx_p1 dt (include/time_disc.hpp:521-526)
    k = 1.0 / (25.0 * 0.7)
    h = 2.0
    assert dx[3, 1] == 2.0 * k / (h * h) * 0.25 + 1.0
    u = sim.field(1)
    assert dx[3, 0] == (-u[2, 2] / 2.0 - k / h) / h * 0.25 and dx[3, 2] == (u[4, 2] / 2.0 - k / h) / h * 0.25
    dy = sim.assemble_system_dy(0.25, 4)
    assert dy.shape == (12, 3) and dy[0, 1] == 1.0 and dy[0, 2] == 1.0 and dy[-1, 0] == 1.0


def test_steady_state_equals_recorded_gmres_steady_state(oracle):
    """Same spatial discretisation, same fixed point: the summed temperature at
    convergence (max delta <= 1e-9) against the reference's recorded GMRES runs."""
    rec = json.load(open(os.path.join(HERE, "golden", "prob1_reference_runs.json")))
    want = {}
    for r in rec["runs"]:
        if r.get("sum_T_interior"):
            want.setdefault((r["cx"], r["cy"]), []).append(r["sum_T_interior"])
    assert set(want) == {(38, 18), (62, 62)}
    for (cx, cy), sums in want.items():
        sim = oracle.ApproxFact(**problems.prob1(cx, cy))
        d, steps = np.inf, 0
        while d > 1e-9 and steps < 2000:
            d = sim.timestep(0.25)
            steps += 1
        assert steps < 2000
        total = float(np.sum(sim.field(0)[1:-1, 1:-1]))
        for w in sums:
            assert abs(total - w) <= 2e-9 * abs(w), (cx, cy, total, w)


@pytest.mark.parametrize("cx,cy", [(18, 8), (38, 18)])
def test_reference_smoke_runs_converge(oracle, cx, cy):
    """TEST(solve_20_10_approxfact / solve_40_20_approxfact), src/test.cpp:591-631."""
    sim = oracle.ApproxFact(**af_default_problem(cx, cy))
    d, steps = np.inf, 0
    while d > 1e-9 and steps < 2000:
        d = sim.timestep(0.25)
        steps += 1
    assert steps < 400 and np.all(np.isfinite(sim.field(0)[1:-1, 1:-1]))


@pytest.mark.gpu
@pytest.mark.parametrize("name,args", [("prob1", (38, 18)), ("af_default", (18, 8)), ("time_dependent", (12, 10)),
                                       ("rough_field", (18, 8)), ("s_iso", (70, 130)), ("prob1", (1, 1)),
                                       ("s_aniso", (130, 67))])
def test_device_steps_bit_exact(ctx, oracle, name, args):
    """Twenty steps on the device against the restatement: every operation is
    element-wise or a sequential line recurrence in the reference's order, so
    temperatures, per-step maxima and the line systems agree bit for bit.

    PARITY UNPINNED for the transient: the reference's own tests of this class
    assert only the boundary rows of the line systems (src/test.cpp:643-648,
    restated in test_assemble_sys_boundary_rows above; its interior check at
    :649-654 is stale and fails against the reference itself) and otherwise
    print (:591-631).  The restatement's transient is therefore pinned by
    nothing but those rows and its steady state; this test shows device ==
    restatement, not restatement == reference."""
    from krylov_amd import host
    prob = af_default_problem(*args) if name == "af_default" else getattr(problems, name)(*args)
    ref = oracle.ApproxFact(**prob)
    dev = host.ApproxFact(ctx, **prob)
    assert np.array_equal(dev.assemble_system_dx(0.25, 0), ref.assemble_system_dx(0.25, 0), equal_nan=True)
    assert np.array_equal(dev.assemble_system_dy(0.25, args[0] - 1), ref.assemble_system_dy(0.25, args[0] - 1),
                          equal_nan=True)
    for _ in range(20):
        a, b = dev.timestep(0.25), ref.timestep(0.25)
        assert a == b
    assert np.array_equal(dev.field(0), ref.field(0), equal_nan=True)
    assert dev.cur_time() == ref.cur_time()


@pytest.mark.gpu
def test_device_needs_solverless_whole_mesh_object(ctx):
    from krylov_amd import host
    import ctypes
    sim = host.ImplicitEuler(ctx, **problems.prob1(), m=4)
    out = ctypes.c_double()
    assert capi.lib().kry_ie_timestep_approx_fact(sim.h, 0.25, ctypes.byref(out)) != 0
    assert b"m = 0" in capi.lib().kry_last_error()
