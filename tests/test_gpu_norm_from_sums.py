"""H(m, m-1) of a cycle's last Arnoldi step from the last sweep's sums.

The reference's gram_schmidt ends with `H(j+1, j) = normalize(next)`
(include/solver.hpp:209); in the last step of a cycle only that norm is used
(the solution combines the first m basis vectors, :153-159).  With one-sweep
steps the solver of a step object takes it as sqrt(||A v||^2 - sum h_k^2) from
sums the last sweep has formed anyway and skips the closing pass over the
basis; the device falls back to the closing pass when the subtraction
cancels.  Checked here against the closing pass itself (switch
kry_gmres_set_norm_from_sums): every other entry of H has the same bits, the
norm agrees to rounding, the vector is formed on demand, the fall-back
triggers at a breakdown, and a replayed graph leaves the host's notes about
the basis as the captured solve did.
"""
import numpy as np
import pytest

from krylov_amd import host, problems

pytestmark = pytest.mark.gpu


def rel(a, b):
    return float(np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300))


@pytest.mark.parametrize("shape,m,graph", [((38, 18), 10, False), ((64, 48), 30, False), ((300, 260), 20, True),
                                           ((33, 65), 50, False), ((12, 10), 8, True), ((520, 40), 43, False)])
def test_last_norm_from_sums_equals_closing_pass(ctx, shape, m, graph):
    prob = problems.s_iso(*shape)
    a = host.ImplicitEuler(ctx, **prob, m=m, use_graph=graph)
    b = host.ImplicitEuler(ctx, **prob, m=m, use_graph=graph)
    b.solver().set_norm_from_sums(False)
    # the sweep with 41 earlier vectors does not deliver <t, t>: a cycle of 43 steps keeps the closing pass
    assert a.solver().get_norm_from_sums() == (m != 43)
    assert not b.solver().get_norm_from_sums()
    for step in range(3):
        da, db = a.timestep(0.25), b.timestep(0.25)
        Ha, Hb = a.solver().hessenberg_matrix(), b.solver().hessenberg_matrix()
        assert a.solver().last_norm_from_sums() == (m != 43)
        assert not b.solver().last_norm_from_sums()
        if step == 0:
            # same input: every entry but the last norm has the same bits
            mask = np.ones_like(Ha, dtype=bool)
            mask[m, m - 1] = False
            assert np.array_equal(Ha[mask], Hb[mask])
            va, vb = a.solver().subspace_vector(m), b.solver().subspace_vector(m)  # formed on demand in `a`
            assert rel(va, vb) <= 1e-13
            assert abs(np.linalg.norm(va) - 1.0) <= 1e-12
        assert abs(Ha[m, m - 1] - Hb[m, m - 1]) <= 1e-13 * abs(Hb[m, m - 1])
        assert rel(a.delta(), b.delta()) <= 1e-13
        assert abs(da - db) <= 1e-13 * abs(db)


def test_cancellation_falls_back_to_the_closing_pass(ctx):
    """As many steps as unknowns: the Krylov space is exhausted in the last step, the norm of the last vector is
    rounding noise and ||A v||^2 - sum h_k^2 cancels completely.  The device notices and forms the vector, so
    both settings give the same bits."""
    prob = problems.s_iso(4, 3)
    m = 6 * 5 - 4
    a = host.ImplicitEuler(ctx, **prob, m=m)
    b = host.ImplicitEuler(ctx, **prob, m=m)
    b.solver().set_norm_from_sums(False)
    assert a.solver().get_norm_from_sums()
    a.timestep(0.25)
    b.timestep(0.25)
    Ha, Hb = a.solver().hessenberg_matrix(), b.solver().hessenberg_matrix()
    assert abs(Hb[m, m - 1]) <= 1e-10 * np.max(np.abs(Hb))
    assert not a.solver().last_norm_from_sums()
    assert np.array_equal(Ha, Hb, equal_nan=True)
    assert np.array_equal(a.delta(), b.delta(), equal_nan=True)


def test_stand_alone_solver_keeps_the_closing_pass(ctx):
    """A stand-alone solver is not bound to an operator that outlives its solves, so it always forms the vector."""
    dev = host.ImplicitEuler(ctx, **problems.s_iso(40, 30), m=12)
    dev.assemble_system(0.25)
    g = host.GMRESSolver(ctx, dev.nloc, 12)
    assert not g.get_norm_from_sums()
    own = dev.solver()
    b = np.linspace(1.0, 2.0, dev.nloc)
    b.reshape(42, 32)[[0, 0, -1, -1], [0, -1, 0, -1]] = 0.0  # the four corner entries are not unknowns
    x = g.solve(dev.operator(), b)
    assert not g.last_norm_from_sums()
    x_own = own.solve(dev.operator(), b)
    assert own.last_norm_from_sums()
    assert rel(x, x_own) <= 1e-13


def test_graph_replay_restores_the_hosts_notes_about_the_basis(ctx):
    """What the host remembers about the basis (how many vectors, un-normalised, strip-major, last vector not yet
    formed) is written when a solve is queued.  A replayed step queues nothing on the host, so after another
    entry point has used the solver in between, the notes are put back to what the captured solve leaves."""
    prob = problems.s_iso(64, 48)
    m = 12
    dev = host.ImplicitEuler(ctx, **prob, m=m, use_graph=True)
    eager = host.ImplicitEuler(ctx, **prob, m=m)
    for _ in range(2):  # capture + launch, then a replay
        dev.timestep(0.25)
        eager.timestep(0.25)
    g = dev.solver()
    assert g.basis_strip_major()
    g.initial_subspace(np.ones(dev.nloc))  # the step-wise interface: one vector, grid order
    assert not g.basis_strip_major()
    with pytest.raises(RuntimeError):
        g.subspace_vector(3)
    d, de = dev.timestep(0.25), eager.timestep(0.25)  # replay
    assert d == de
    assert g.basis_strip_major()
    for k in (0, 3, m):
        assert np.array_equal(g.subspace_vector(k), eager.solver().subspace_vector(k))
