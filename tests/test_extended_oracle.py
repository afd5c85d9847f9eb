"""The CPU statement of the extended solver mode (restarted GMRES(m) with
Givens rotations and a residual monitor).  The reference has no such mode
(include/solver.hpp:124-161 is one fixed cycle), so there are no reference
fixtures for it: it is pinned against an independent implementation,
scipy.sparse.linalg.gmres, whose inner-iteration count for the same restart
length and tolerance it must reproduce, and against the true residual."""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spl

from krylov_amd import problems

CASES = [("prob1", (38, 18), 10), ("prob1", (38, 18), 30), ("s_iso", (64,), 20), ("time_dependent", (12, 10), 8)]


@pytest.mark.parametrize("name,args,m", CASES)
def test_iteration_count_equals_scipy(oracle, name, args, m):
    sim = oracle.ImplicitEuler(**getattr(problems, name)(*args), m=m)
    sim.assemble_system(0.25)
    rp, col, val = sim.csr()
    n = sim.n
    A = sp.csr_matrix((val, col, rp), shape=(n, n))
    b = sim.sol_vec()
    g = oracle.Gmres(n, m)
    g.set_extended(1e-10, 60)
    x = g.solve(sim.op(), b)
    iters, cycles, converged, est = g.last_stats()
    count = [0]

    def cb(_):
        count[0] += 1

    xs, info = spl.gmres(A, b, restart=m, rtol=1e-10, atol=0.0, maxiter=60, callback=cb, callback_type="pr_norm")
    assert info == 0 and converged
    assert iters == count[0]
    assert cycles == (iters + m - 1) // m
    true = np.linalg.norm(b - A @ x) / np.linalg.norm(b)
    assert true <= 1.0001e-10 and abs(true - est) <= 1e-3 * est
    assert np.linalg.norm(x - xs) <= 1e-8 * np.linalg.norm(xs)


@pytest.mark.parametrize("name,args,m", [("prob1", (38, 18), 10), ("s_aniso", (48, 48), 10)])
def test_right_preconditioned_residual_is_true_residual(oracle, name, args, m):
    sim = oracle.ImplicitEuler(**getattr(problems, name)(*args), m=m, tdp=True)
    sim.assemble_system(0.25)
    rp, col, val = sim.csr()
    n = sim.n
    A = sp.csr_matrix((val, col, rp), shape=(n, n))
    b = sim.sol_vec()
    g = oracle.Gmres(n, m, True)
    g.set_extended(1e-11, 40)
    x = g.solve(sim.op(), b)
    iters, cycles, converged, est = g.last_stats()
    assert converged and iters < 60  # the line preconditioner cuts the count by several times
    true = np.linalg.norm(b - A @ x) / np.linalg.norm(b)
    assert true <= 1.01e-11 and abs(true - est) <= 1e-2 * est
    xd = spl.spsolve(A.tocsc(), b)
    assert np.linalg.norm(x - xd) <= 1e-7 * np.linalg.norm(xd)


def test_trivial_cases(oracle):
    n = 12
    eye = sp.identity(n, format="csr")
    op = oracle.Op.csr(eye.indptr.astype(np.int64), eye.indices.astype(np.int32), eye.data.astype(np.float64))
    g = oracle.Gmres(n, 5)
    g.set_extended(1e-12, 5)
    b = np.arange(1.0, n + 1.0)
    x = g.solve(op, b)  # lucky breakdown after one step
    assert g.last_stats()[:3] == (1, 1, True)
    assert np.allclose(x, b, rtol=0, atol=1e-14)
    x = g.solve(op, np.zeros(n))  # zero right-hand side: nothing to do
    assert g.last_stats()[:3] == (0, 1, True)
    assert np.array_equal(x, np.zeros(n))


def test_timestep_with_extended_solver(oracle):
    """An implicit step solved to 1e-12 gives the same update as a direct solve."""
    sim = oracle.ImplicitEuler(**problems.prob1(), m=20, extended=(1e-12, 50))
    sim.assemble_system(0.25)
    rp, col, val = sim.csr()
    A = sp.csr_matrix((val, col, rp), shape=(sim.n, sim.n))
    delta = spl.spsolve(A.tocsc(), sim.sol_vec())
    sim2 = oracle.ImplicitEuler(**problems.prob1(), m=20, extended=(1e-12, 50))
    d = sim2.timestep(0.25)
    interior = [sim2.cell_idx(i, j) for i in range(sim2.cx) for j in range(sim2.cy)]
    assert abs(d - np.max(np.abs(delta[interior]))) <= 1e-9 * d
    assert sim2.solver().last_stats()[2]


def test_stagnation_is_reported(oracle):
    """The strongly convective rough-field case does not converge with short
    cycles: every cycle runs its m steps and the flag stays clear."""
    sim = oracle.ImplicitEuler(**problems.rough_field(18, 8), m=12)
    sim.assemble_system(0.25)
    g = oracle.Gmres(sim.n, 12)
    g.set_extended(1e-10, 4)
    g.solve(sim.op(), sim.sol_vec())
    iters, cycles, converged, est = g.last_stats()
    assert (iters, cycles, converged) == (48, 4, False) and 1e-3 < est < 1.0
