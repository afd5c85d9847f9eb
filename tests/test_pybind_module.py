"""The pybind11 module `krylov`: same public names as the reference's
src/bindpy.cpp; host-side classes work without a GPU, device classes need one."""
import json
import math
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "krylov_amd"))
sys.path.insert(0, os.path.join(ROOT, "examples"))

REFERENCE_NAMES = sorted(
    ["BCType", "BoundaryConditionHoriz", "BoundaryConditionVert", "LUSolver", "Matrix", "Mesh", "SecondOrderCentered",
     "TriDiagSolver", "Vector", "_TimeDiscBase", "dirichlet", "neumann", "ImplicitEulerLUSolver",
     "ImplicitEulerAFSolver"]
    + ["%s%d" % (p, m) for p in ("GMRESSolver", "GMRESTDPSolver", "ImplicitEulerGMRESSolver",
                                 "ImplicitEulerGMRESTDPSolver") for m in (10, 20, 30, 40, 50)])


@pytest.fixture(scope="module")
def krylov():
    import krylov as k
    return k


def test_public_names_match_reference_module(krylov):
    """src/bindpy.cpp:21-708 defines 34 public names."""
    names = sorted(n for n in dir(krylov) if not n.startswith("__"))
    assert names == REFERENCE_NAMES and len(names) == 34


def test_prob1_import_line(krylov):
    """scripts/prob1.py:4-9"""
    from krylov import SecondOrderCentered, Mesh, BCType  # noqa: F401
    from krylov import (ImplicitEulerLUSolver as IELU, ImplicitEulerAFSolver as IEAF,  # noqa: F401
                        ImplicitEulerGMRESTDPSolver10 as IEGMRES10)
    from krylov import BoundaryConditionHoriz as BCH  # noqa: F401
    from krylov import BoundaryConditionVert as BCV  # noqa: F401
    with pytest.raises(RuntimeError, match="not part of this build"):
        IELU()


def test_mesh_surface(krylov):
    """src/bindpy.cpp:23-97"""
    m = krylov.Mesh((0.0, 0.0), (40.0, 1.0), 38, 18, lambda x, y: (x + y, 18.0 * y * (1.0 - y), 0.5))
    assert (m.cells_x(), m.cells_y()) == (38, 18)
    assert m.dx() == 40.0 / 38 and m.dy() == 1.0 / 18
    T = m.Temp_array()
    assert T.shape == (40, 20) and T.dtype == np.float64
    assert all(np.isnan(T[c]) for c in ((0, 0), (0, -1), (-1, 0), (-1, -1)))
    assert np.isnan(T[0, 5]) and np.isnan(T[7, 0])  # ghosts start as NaN
    assert T[1, 1] == m.median_x(0) + m.median_y(0) == m[0, 0]
    m[3, 4] = 7.5
    assert m.Temp_array()[4, 5] == 7.5
    T[1, 1] = -1.0  # arrays are copies
    assert m[0, 0] != -1.0
    assert m.vel_v_array()[5, 5] == 0.5
    gx, gy = m.grid_x(), m.grid_y()
    assert gx[3, 7] == m.median_x(2) and gy[3, 7] == m.median_y(6)
    empty = krylov.Mesh((0.0, 0.0), (1.0, 1.0), 4, 4)
    assert empty[0, 0] == 0.0 and np.isnan(empty.Temp_array()[0, 1])


def test_boundary_conditions_on_a_mesh(krylov):
    """TEST(boundary_cond, homogeneous) through the Python factories
    (src/test.cpp:1182-1424, src/bindpy.cpp:99-222)."""
    rng = np.random.default_rng(3)
    vals = rng.uniform(-100, 100, (32, 64, 3))
    m = krylov.Mesh((0.0, 0.0), (1.0, 1.0), 32, 64,
                    lambda x, y: tuple(vals[int(round(x * 32 - 0.5)), int(round(y * 64 - 0.5))]))
    V, H, BT = krylov.BoundaryConditionVert, krylov.BoundaryConditionHoriz, krylov.BCType
    zero = lambda x, y, t: 0.0
    H.Temp_bc(m, H.BCSide.top, BT.dirichlet, zero).apply(0.0)
    H.vel_u_bc(m, H.BCSide.bottom, BT.neumann, zero).apply(0.0)
    V.vel_v_bc(m, V.BCSide.right, BT.neumann, zero).apply(0.0)
    V.Temp_bc(m, V.BCSide.left, BT.dirichlet, zero).apply(0.0)
    T, u, v = m.Temp_array(), m.vel_u_array(), m.vel_v_array()
    assert np.array_equal(T[1:-1, 1:-1], vals[:, :, 0])
    assert np.array_equal(T[1:-1, -1], -T[1:-1, -2])
    assert np.array_equal(u[1:-1, 0], u[1:-1, 1])
    assert np.array_equal(v[-1, 1:-1], v[-2, 1:-1])
    assert np.array_equal(T[0, 1:-1], -T[1, 1:-1])  # the stand-alone factory reads the LEFT column
    assert np.isnan(T[0, 0]) and np.isnan(u[-1, -1]) and np.isnan(v[0, 5])
    # inhomogeneous Dirichlet: ghost = 2 bc - boundary at (left_x(0), median_y(j))
    V.vel_u_bc(m, V.BCSide.left, BT.dirichlet, lambda x, y, t: x + 2.0 * y + t).apply(0.5)
    j = 10
    assert m.vel_u_array()[0, j + 1] == 2.0 * (0.0 + 2.0 * m.median_y(j) + 0.5) - vals[0, j, 1]


def test_space_disc_coefficients(krylov):
    """src/bindpy.cpp:229-236 with the formulas of src/test.cpp:861-886"""
    m = krylov.Mesh((0.0, 0.0), (1.0, 1.0), 16, 16, lambda x, y: (0.0, 2.0 * y * math.sin(math.pi * x), x * math.cos(math.pi * y)))
    sd = krylov.SecondOrderCentered(1.0, 50.0, 0.7, 0.1)
    dx, dy = m.dx(), m.dy()
    u, v = m.vel_u_array(), m.vel_v_array()
    i, j = 5, 7
    assert abs(sd.Dx_m1(m, i, j) - (-1.0 / (35.0 * dx * dx) - u[i, j + 1] / (2.0 * dx))) <= 1e-12
    assert abs(sd.Dx_p1(m, i, j) - (-1.0 / (35.0 * dx * dx) + u[i + 2, j + 1] / (2.0 * dx))) <= 1e-12
    assert abs(sd.Dy_m1(m, i, j) - (-1.0 / (35.0 * dy * dy) - v[i + 1, j] / (2.0 * dy))) <= 1e-12
    assert abs(sd.Dy_p1(m, i, j) - (-1.0 / (35.0 * dy * dy) + v[i + 1, j + 2] / (2.0 * dy))) <= 1e-12
    assert abs(sd.Dx_0(m, i, j) - 2.0 / (35.0 * dx * dx)) <= 1e-12 and abs(sd.Dy_0(m, i, j) - 2.0 / (35.0 * dy * dy)) <= 1e-12


def test_host_side_direct_solvers(krylov):
    """TEST(lu_decomp, solvers) src/test.cpp:657-686 (exact) and
    TEST(tridiag, solvers) :209-220 through Matrix / Vector (src/bindpy.cpp:593-667)."""
    m1 = np.array([[2.0, 3.0, 4.0, 20.0], [5.0, 6.0, 7.0, 30.0], [8.0, 9.0, 20.0, 40.0], [11.0, 12.0, 130.0, 160.0]])
    A = krylov.Matrix.from_array(m1)
    assert tuple(A.shape()) == (4, 4) and A[2, 3] == 40.0
    lu = krylov.LUSolver(4)
    l, u = lu.lu_decomp(A)
    assert np.array_equal(np.triu(l.to_array(), 1), np.zeros((4, 4))) and np.array_equal(np.tril(u.to_array(), -1), np.zeros((4, 4)))
    x_expected = np.array([-1.0, 1.0, 2.0, -4.0])
    b = krylov.Vector.from_array(m1 @ x_expected)
    assert np.array_equal(lu.solve(A, b).to_array(), x_expected)
    n = 10
    T = np.zeros((n, n))
    rhs = np.zeros(n)
    rhs[4], rhs[5], rhs[6] = 1.0, 5.0, 1.0
    for i in range(n):
        T[i, i] = -2.0
        if i < n - 1:
            T[i + 1, i] = T[i, i + 1] = 1.0
    x = krylov.TriDiagSolver(n).solve(krylov.Matrix.from_array(T), krylov.Vector.from_array(rhs)).to_array()
    expected = np.array([5.0, 10.0, 15.0, 20.0, 25.0, 199.0 / 7.0, 24.0, 18.0, 12.0, 6.0]) * (-7.0 / 11.0)
    assert np.max(np.abs(x - expected)) <= 1e-14
    v = krylov.Vector([3])
    v[1] = 2.5
    assert v[1] == 2.5 and tuple(v.shape()) == (3,)


# ---- device-backed classes -------------------------------------------------------

def _golden():
    with open(os.path.join(ROOT, "tests", "golden", "prob1_reference_runs.json")) as f:
        return json.load(f)["runs"]


@pytest.mark.gpu
def test_prob1_replay_through_the_module(krylov):
    """scripts/prob1.py:65-98 with GMRESSolver<10>: 147 steps, the reference's
    per-step values and final field; then the script's default solver, which
    diverges like the reference."""
    import prob1_headless
    run = _golden()[2]
    solver, hist = prob1_headless.run("ImplicitEulerGMRESSolver10", verbose=False)
    assert len(hist) == run["steps_to_1e-9"]
    for k, want in run["max_delta"].items():
        assert abs(hist[int(k)] - want) <= 1e-9 * want + 1e-13
    mesh = solver.mesh()
    T = mesh.Temp_array()
    assert T.shape == (40, 20) and np.isnan(T[0, 0])
    assert abs(T[1:-1, 1:-1].sum() - run["sum_T_interior"]) <= 1e-10 * run["sum_T_interior"]
    assert solver.cur_time() == 0.25 * len(hist)
    assert abs(solver.space_disc().Dx_0(mesh, 0, 0) - 2.0 / (17.5 * mesh.dx() ** 2)) <= 1e-15
    A = solver.system().to_array()
    assert A.shape == (796, 796) and np.count_nonzero(A) == 3644
    assert solver.delta().to_array().shape == (796,) and solver.solution().to_array().shape == (796,)
    tdp = _golden()[0]
    _, hist = prob1_headless.run("ImplicitEulerGMRESTDPSolver10", verbose=False, max_steps=120)
    for k in range(5):
        want = tdp["max_delta"][str(k)]
        assert abs(hist[k] - want) <= 1e-9 * want
    assert math.isnan(hist[-1]) and len(hist) == tdp["first_nan_step_1based"]


@pytest.mark.gpu
def test_standalone_gmres_from_python(krylov):
    """GMRESSolver10(n).solve(Matrix, Vector), src/bindpy.cpp:669-707, with the
    reference's arnoldi system (src/test.cpp:187-207, :283-289)."""
    n = 10
    A = np.zeros((n, n))
    b = np.zeros(n)
    b[4], b[5], b[6] = 1.0, 5.0, 1.0
    for i in range(n):
        A[i, i] = -2.0
        if i < n - 1:
            A[i + 1, i] = A[i, i + 1] = 1.0
    x = krylov.GMRESSolver10(n).solve(krylov.Matrix.from_array(A), krylov.Vector.from_array(b)).to_array()
    expected = np.array([5.0, 10.0, 15.0, 20.0, 25.0, 199.0 / 7.0, 24.0, 18.0, 12.0, 6.0]) * (-7.0 / 11.0)
    assert np.max(np.abs(x - expected)) <= 4e-12
    xt = krylov.GMRESTDPSolver10(n).solve(krylov.Matrix.from_array(A), krylov.Vector.from_array(b)).to_array()
    assert np.all(np.isfinite(xt))


@pytest.mark.gpu
def test_approx_fact_through_the_module(krylov, oracle):
    """ImplicitEulerAFSolver (src/bindpy.cpp:270-295) with scripts/prob1.py's
    configuration: the time loop to max delta <= 1e-9 equals the CPU
    restatement step for step, and ends on the steady state of the recorded
    GMRES runs."""
    import prob1_headless
    from krylov_amd import problems
    solver, hist = prob1_headless.run("ImplicitEulerAFSolver", verbose=False)
    ref = oracle.ApproxFact(**problems.prob1())
    want = []
    d = float("inf")
    while d > 1e-9:
        d = ref.timestep(0.25)
        want.append(d)
    assert hist == want
    T = solver.mesh().Temp_array()
    assert np.array_equal(T[1:-1, 1:-1], ref.field(0)[1:-1, 1:-1])
    assert abs(T[1:-1, 1:-1].sum() - _golden()[2]["sum_T_interior"]) <= 2e-9 * T[1:-1, 1:-1].sum()
    dx = solver.assemble_system_dx(0.25, 1).to_array()
    assert dx.shape == (40, 3) and np.isnan(dx[0, 0]) and dx[0, 1] == 1.0 and dx[-1, 1] == 1.0
    assert np.array_equal(dx, ref.assemble_system_dx(0.25, 1), equal_nan=True)
    assert np.array_equal(solver.assemble_system_dy(0.25, 3).to_array(), ref.assemble_system_dy(0.25, 3), equal_nan=True)


def _time_dependent_module_solver(krylov, name, cx=24, cy=16):
    """A configuration whose twelve boundary functions are Python callables that
    depend on time (krylov_amd.problems.time_dependent, argument order of
    src/bindpy.cpp:448-463: T bottom, top, left, right, then u, then v)."""
    from krylov_amd import problems
    from krylov_amd.capi import BOTTOM, TOP, LEFT, RIGHT
    prob = problems.time_dependent(cx, cy)
    calls = {"n": 0}

    def scalar(fn):
        def f(x, y, t):
            calls["n"] += 1
            return float(fn(np.float64(x), np.float64(y), t))
        return f

    bcs = []
    for f in range(3):
        for side in (BOTTOM, TOP, LEFT, RIGHT):
            is_d, fn = prob["bcs"][(f, side)]
            bcs.append((krylov.BCType.dirichlet if is_d else krylov.BCType.neumann, scalar(fn)))
    init = prob["init"]
    sim = getattr(krylov, name)(prob["corner1"], prob["corner2"], cx, cy,
                                lambda x, y: tuple(float(v) for v in init(np.float64(x), np.float64(y))),
                                krylov.SecondOrderCentered(*prob["params"]), *bcs)
    return sim, prob, calls


@pytest.mark.gpu
@pytest.mark.parametrize("graph", [False, True])
@pytest.mark.parametrize("name,m,tdp", [("ImplicitEulerGMRESSolver10", 10, False), ("ImplicitEulerGMRESTDPSolver10", 10, True)])
def test_time_dependent_python_callables_through_the_module(krylov, oracle, name, m, tdp, graph):
    """Six implicit steps driven through the pybind classes with boundary
    conditions that are time-dependent PYTHON callables: the C++ mirror samples
    them for the next step's start time while the device works on the current
    step (pipelined, krylov_amd/host/time_disc.hpp).  Every step must use the
    values of ITS start time (include/time_disc.hpp:287,292): max|delta| per
    step, the temperature field and the velocity ghost cells (which follow the
    moving boundary values exactly) against the CPU restatement, with eagerly
    launched and with graph-replayed steps."""
    sim, prob, calls = _time_dependent_module_solver(krylov, name)
    sim.use_graph(graph)
    ref = oracle.ImplicitEuler(**prob, m=m, tdp=tdp)
    n0 = calls["n"]
    for step in range(6):
        d, dr = sim.timestep(0.25), ref.timestep(0.25)
        if not (math.isfinite(dr) and dr < 1e100):
            break
        assert abs(d - dr) <= 1e-10 * abs(dr), (step, d, dr)
        assert sim.cur_time() == ref.cur_time() == 0.25 * (step + 1)
    assert not sim.failed() and sim.last_error() == ""
    # twelve functions sampled once per step (one sample ahead), on 2 (cx + cy) boundary cells each
    assert calls["n"] - n0 == 6 * 3 * 2 * (24 + 16)
    mesh = sim.mesh()
    T, Tr = mesh.Temp_array(), ref.field(0)
    assert float(np.linalg.norm(T[1:-1, 1:-1] - Tr[1:-1, 1:-1]) / np.linalg.norm(Tr[1:-1, 1:-1])) <= 1e-10
    assert np.array_equal(mesh.vel_u_array(), ref.field(1), equal_nan=True)
    assert np.array_equal(mesh.vel_v_array(), ref.field(2), equal_nan=True)


@pytest.mark.gpu
def test_failed_step_returns_nan_and_python_errors_surface(krylov):
    """timestep() is noexcept as in the reference (include/time_disc.hpp:277): a
    step that cannot run returns NaN, which ends `while max_delta > 1e-9`.  An
    exception raised by a Python boundary callable is raised again in Python."""
    sim, prob, calls = _time_dependent_module_solver(krylov, "ImplicitEulerGMRESSolver10", 8, 6)
    assert math.isfinite(sim.timestep(0.25)) and not sim.failed()

    class Boom(Exception):
        pass

    state = {"armed": False}

    def bad(x, y, t):
        if state["armed"]:
            raise Boom("boundary function failed at t = %g" % t)
        return 0.0

    from krylov_amd import problems
    p = problems.prob1(8, 6)
    zero = (krylov.BCType.dirichlet, lambda x, y, t: 0.0)
    sim2 = krylov.ImplicitEulerGMRESSolver10(p["corner1"], p["corner2"], 8, 6, lambda x, y: (0.0, y * (1.0 - y), 0.0),
                                             krylov.SecondOrderCentered(*p["params"]),
                                             (krylov.BCType.dirichlet, bad), *([zero] * 11))
    assert math.isfinite(sim2.timestep(0.25))
    state["armed"] = True
    with pytest.raises(Boom):
        sim2.timestep(0.25)
    assert sim2.failed()
    state["armed"] = False
    d = sim2.timestep(0.25)
    assert math.isfinite(d) and not sim2.failed()
    # The step whose sampling failed had already been queued: it ran, the time advanced, and the next call
    # sampled the boundary values of the new time again before queueing anything.  The object is where a
    # twin without the failure is after three steps.
    twin = krylov.ImplicitEulerGMRESSolver10(p["corner1"], p["corner2"], 8, 6, lambda x, y: (0.0, y * (1.0 - y), 0.0),
                                             krylov.SecondOrderCentered(*p["params"]),
                                             (krylov.BCType.dirichlet, lambda x, y, t: 0.0), *([zero] * 11))
    dt = [twin.timestep(0.25) for _ in range(3)]
    assert d == dt[2]
    assert sim2.cur_time() == twin.cur_time() == 0.75
    assert np.array_equal(sim2.mesh().Temp_array(), twin.mesh().Temp_array(), equal_nan=True)
