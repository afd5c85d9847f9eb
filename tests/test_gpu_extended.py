"""The extended solver mode on the device (KRY_GMRES_EXTENDED: restarted
GMRES(m), Givens-rotation least squares, residual monitor and stop flag on the
device) against its CPU statement: iteration counts within +-1, solutions to
1e-10.  The reference has no counterpart of this mode; the CPU statement is
pinned against scipy in tests/test_extended_oracle.py."""
import numpy as np
import pytest
import scipy.sparse as sp

from krylov_amd import capi, host, problems

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


# ("chain", K): K Gram-Schmidt links per sweep, basis read twice per step; ("sweep", s):
# one-sweep Arnoldi steps from step s on (the residual monitor then runs one sweep behind)
VARIANTS = [("chain", 1), ("chain", 4), ("chain", 8), ("chain", 16), ("chain", 32), ("sweep", 0), ("sweep", 2),
            ("sweep", 6)]


@pytest.mark.parametrize("variant", VARIANTS, ids=lambda v: "%s%d" % v)
@pytest.mark.parametrize("name,args,m,tdp,rtol", [
    ("prob1", (38, 18), 10, False, 1e-12),
    ("prob1", (38, 18), 30, False, 1e-12),
    ("prob1", (38, 18), 10, True, 1e-12),
    ("s_iso", (64,), 20, False, 1e-12),
    ("s_aniso", (48, 48), 10, True, 1e-12),
    ("time_dependent", (12, 10), 8, False, 1e-12),
    ("prob1", (37, 17), 10, False, 1e-12),  # odd row length: scalar operator kernel
])
def test_solve_matches_cpu_statement(ctx, oracle, name, args, m, tdp, rtol, variant):
    prob = getattr(problems, name)(*args)
    ref = oracle.ImplicitEuler(**prob, m=m, tdp=tdp)
    if variant[0] == "sweep":
        if tdp:
            pytest.skip("the preconditioned solver keeps the chain")
        kw = {"one_sweep": variant[1]}
    else:
        kw = {"one_sweep": -1, "mgs_block": variant[1]}
    dev = host.ImplicitEuler(ctx, **prob, m=m, tdp=tdp, extended=(rtol, 60), **kw)
    if variant[0] == "sweep":
        assert dev.solver().get_one_sweep() == min(variant[1], m - 2)
    ref.assemble_system(0.25)
    dev.assemble_system(0.25)
    b = ref.sol_vec()
    gref = oracle.Gmres(ref.n, m, tdp)
    gref.set_extended(rtol, 60)
    x_ref = gref.solve(ref.op(), b)
    it_ref, cy_ref, cv_ref, rr_ref = gref.last_stats()
    g = dev.solver()
    x = dev.compact(g.solve(dev.operator(), dev.expand(b).reshape(-1)))
    it, cy, cv, rr = g.last_stats()
    assert cv and cv_ref
    assert abs(it - it_ref) <= 1, (it, it_ref)
    assert cy == (it + m - 1) // m
    assert rr <= rtol
    rp, col, val = ref.csr()
    A = sp.csr_matrix((val, col, rp), shape=(ref.n, ref.n))
    assert np.linalg.norm(b - A @ x) <= 1.05 * rtol * np.linalg.norm(b)
    # both iterates solve the system to rtol; they agree far below that when the counts agree
    assert rel(x, x_ref) <= (1e-10 if rtol <= 1e-12 else 50 * rtol)


def test_general_matrix_and_trivial_cases(ctx):
    n = 12
    g = host.GMRESSolver(ctx, n, 5, extended=(1e-12, 5))
    op = host.Operator.dense(ctx, np.eye(n))
    b = np.arange(1.0, n + 1.0)
    x = g.solve(op, b)  # lucky breakdown after one step
    assert g.last_stats()[:3] == (1, 1, True)
    assert np.allclose(x, b, rtol=0, atol=1e-14)
    x = g.solve(op, np.zeros(n))
    assert g.last_stats()[:3] == (0, 1, True)
    assert np.array_equal(x, np.zeros(n))
    # a non-symmetric dense system that needs restarts
    rng = np.random.default_rng(11)
    A = np.eye(40) * 4.0 + rng.uniform(-1, 1, (40, 40))
    b = rng.uniform(-1, 1, 40)
    g = host.GMRESSolver(ctx, 40, 8, extended=(1e-12, 40))
    x = g.solve(host.Operator.dense(ctx, A), b)
    it, cy, cv, rr = g.last_stats()
    assert cv and cy > 1
    assert np.linalg.norm(b - A @ x) <= 1.05e-12 * np.linalg.norm(b)


def test_not_converged_is_reported(ctx, oracle):
    prob = problems.prob1()
    dev = host.ImplicitEuler(ctx, **prob, m=5, extended=(1e-14, 2))
    dev.assemble_system(0.25)
    g = dev.solver()
    ref = oracle.ImplicitEuler(**prob, m=5)
    ref.assemble_system(0.25)
    g.solve(dev.operator(), dev.expand(ref.sol_vec()).reshape(-1))
    it, cy, cv, rr = g.last_stats()
    assert (it, cy, cv) == (10, 2, False) and rr > 1e-14


def test_stagnation_matches_cpu_statement(ctx, oracle):
    prob = problems.rough_field(18, 8)
    ref = oracle.ImplicitEuler(**prob, m=12)
    ref.assemble_system(0.25)
    gref = oracle.Gmres(ref.n, 12)
    gref.set_extended(1e-10, 4)
    x_ref = gref.solve(ref.op(), ref.sol_vec())
    dev = host.ImplicitEuler(ctx, **prob, m=12, extended=(1e-10, 4))
    dev.assemble_system(0.25)
    g = dev.solver()
    x = dev.compact(g.solve(dev.operator(), dev.expand(ref.sol_vec()).reshape(-1)))
    it, cy, cv, rr = g.last_stats()
    assert (it, cy, cv) == (48, 4, False)
    assert abs(rr - gref.last_stats()[3]) <= 1e-8 * rr
    assert rel(x, x_ref) <= 1e-10


def test_mode_is_opt_in(ctx):
    g = host.GMRESSolver(ctx, 10, 4)
    with pytest.raises(RuntimeError):
        g.set_tolerance(1e-8, 3)
    with pytest.raises(RuntimeError):
        g.last_stats()


@pytest.mark.parametrize("name,args,m,tdp", [("prob1", (38, 18), 10, False), ("prob1", (38, 18), 10, True),
                                             ("time_dependent", (12, 10), 8, False)])
def test_timesteps_match_cpu_statement(ctx, oracle, name, args, m, tdp):
    """Ten implicit steps solved to 1e-12: the temperature field and the
    per-step iteration counts against the CPU statement.  With the line
    preconditioner this is the run that the reference's own preconditioned
    solver cannot do (its result lacks M^-1 and diverges)."""
    prob = getattr(problems, name)(*args)
    ref = oracle.ImplicitEuler(**prob, m=m, tdp=tdp, extended=(1e-12, 60))
    dev = host.ImplicitEuler(ctx, **prob, m=m, tdp=tdp, extended=(1e-12, 60))
    for step in range(10):
        d_ref = ref.timestep(0.25)
        d = dev.timestep(0.25)
        it_ref = ref.solver().last_stats()[0]
        it, cy, cv, rr = dev.solver().last_stats()
        assert cv
        assert abs(it - it_ref) <= 1, (step, it, it_ref)
        assert abs(d - d_ref) <= 1e-10 * abs(d_ref) + 1e-13
    T = dev.field(capi.FIELD_T)[1:-1, 1:-1]
    Tr = ref.field(0)[1:-1, 1:-1]
    assert rel(T, Tr) <= 1e-10


def test_early_convergence_after_a_later_hand_over(ctx, oracle):
    """A solve that meets its tolerance within the chain steps never runs the
    one-sweep steps: the scale factors those steps keep per basis slot must not
    survive from an earlier solve (they start every cycle at 1).  First a long
    solve with the default hand-over (slots 3.. get scale factors != 1), then
    the hand-over moves to step 12 and a loose tolerance is met after seven
    chain steps."""
    prob = problems.s_iso(64, 48)
    m = 20
    ref = oracle.ImplicitEuler(**prob, m=m)
    ref.assemble_system(0.25)
    b = ref.sol_vec()
    dev = host.ImplicitEuler(ctx, **prob, m=m, extended=(1e-12, 10))
    dev.assemble_system(0.25)
    g = dev.solver()
    x = dev.compact(g.solve(dev.operator(), dev.expand(b).reshape(-1)))
    assert g.last_stats()[0] > 12
    g.set_one_sweep(12)
    assert g.get_one_sweep() == 12
    g.set_tolerance(5e-2, 10)
    gref = oracle.Gmres(ref.n, m)
    gref.set_extended(5e-2, 10)
    x_ref = gref.solve(ref.op(), b)
    x = dev.compact(g.solve(dev.operator(), dev.expand(b).reshape(-1)))
    it, cy, cv, rr = g.last_stats()
    assert cv and it == gref.last_stats()[0] and it < 12
    assert rel(x, x_ref) <= 1e-10


def test_hand_over_stays_within_the_chains_first_group(ctx, oracle):
    """The first one-sweep step needs the Gram matrix of every stored vector;
    the chain carries Gram entries only within a group of `mgs_block` vectors,
    so the hand-over step in effect is at most mgs_block - 1."""
    prob = problems.s_iso(40, 30)
    m = 20
    ref = oracle.ImplicitEuler(**prob, m=m)
    ref.assemble_system(0.25)
    x_ref = ref.solver().solve(ref.op(), ref.sol_vec())
    dev = host.ImplicitEuler(ctx, **prob, m=m, mgs_block=4, one_sweep=9)
    assert dev.solver().get_one_sweep() == 3
    dev.assemble_system(0.25)
    x = dev.compact(dev.solver().solve(dev.operator(), dev.expand(ref.sol_vec())))
    assert rel(x, x_ref) <= 1e-12
