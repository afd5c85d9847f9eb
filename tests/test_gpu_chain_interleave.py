"""Chunk-interleaved basis of chain-only solves against contiguous vectors.

The reference keeps its Krylov basis as a std::vector of vectors
(include/solver.hpp:236).  On one rank, a solve whose every Arnoldi step runs
the Gram-Schmidt chain (the line preconditioner of GMRESTDPSolver,
include/solver.hpp:254-331, more steps than the one-sweep kernels serve, CSR
operators) may store the vectors 512 values at a time, interleaved; it does so
where that pays (from 32 steps per cycle on, or where the preconditioner kernel
also normalises its operand), and with KRY_CHAIN_INTERLEAVE=2 wherever it can.
Only addresses change: every value is produced by the same lane with the same
operations in the same order, so everything must have the same bits as with
KRY_CHAIN_INTERLEAVE=0 (both read when a solver is created).
"""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from krylov_amd import capi, host, problems

pytestmark = pytest.mark.gpu


class environment:
    """Solvers created inside read this setting."""

    def __init__(self, name, value="0"):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = os.environ.get(self.name)
        os.environ[self.name] = self.value

    def __exit__(self, *exc):
        if self.old is None:
            del os.environ[self.name]
        else:
            os.environ[self.name] = self.old


def contiguous_basis():
    return environment("KRY_CHAIN_INTERLEAVE")


def interleaved_basis():
    return environment("KRY_CHAIN_INTERLEAVE", "2")


def pair(ctx, **kw):
    with interleaved_basis():
        a = host.ImplicitEuler(ctx, **kw)
    with contiguous_basis():
        b = host.ImplicitEuler(ctx, **kw)
    return a, b


@pytest.mark.parametrize("shape,m,kw", [
    ((38, 18), 10, dict(tdp=True)),
    ((64, 48), 40, dict(tdp=True, fixed_preconditioning=True)),
    ((33, 65), 50, dict(tdp=True)),
    ((300, 260), 70, dict()),            # beyond the one-sweep range
    ((31, 17), 64, dict()),              # odd number of unknowns, less than a chunk per grid row
    ((130, 120), 12, dict(one_sweep=-1)),
    ((64, 48), 10, dict(tdp=True, use_graph=True)),
    ((96, 80), 8, dict(tdp=True, extended=(1e-9, 30))),
    ((96, 80), 60, dict(extended=(1e-11, 5))),
])
def test_step_has_the_same_bits_as_with_contiguous_vectors(ctx, shape, m, kw):
    prob = problems.s_aniso(*shape) if kw.get("tdp") else problems.s_iso(*shape)
    a, b = pair(ctx, **prob, m=m, **kw)
    for step in range(3):
        da, db = a.timestep(0.25), b.timestep(0.25)
        assert a.solver().basis_chunk_interleaved()
        assert not b.solver().basis_chunk_interleaved()
        assert not a.solver().basis_strip_major()
        assert da == db
        assert np.array_equal(a.delta(), b.delta(), equal_nan=True)
        assert np.array_equal(a.solver().hessenberg_matrix(), b.solver().hessenberg_matrix(), equal_nan=True)
        if "extended" in kw:
            assert a.solver().last_stats() == b.solver().last_stats()
    if "extended" not in kw:
        for k in sorted({0, 1, m // 2, m - 1, m}):
            va, vb = a.solver().subspace_vector(k), b.solver().subspace_vector(k)
            assert np.array_equal(va, vb, equal_nan=True)
            assert abs(np.linalg.norm(va) - 1.0) <= 1e-12


# (the 16-byte-lane form of the preconditioner kernel serves meshes with at least 8177 lines of even length)
@pytest.mark.parametrize("shape,m,kw,fused", [
    ((8190, 600), 10, dict(), True),                              # lines of 602 values: 9 whole chunks of 64 and a tail
    ((8200, 1022), 20, dict(fixed_preconditioning=True), True),   # 16 whole chunks, no tail
    ((8190, 515), 5, dict(use_graph=True), False),                # odd line length
    ((8180, 700), 12, dict(), True),                              # the last workgroup has 6 of its 16 lines
    ((8190, 510), 12, dict(use_graph=True), True),                # exactly 8 chunks
    ((8190, 508), 12, dict(), False),                             # 7 chunks and a tail: too short
    ((300, 600), 12, dict(), False),                              # too few lines
])
def test_normalisation_inside_the_preconditioner_kernel_has_the_same_bits(ctx, shape, m, kw, fused):
    """Between two chain steps of a fixed cycle, GMRESTDPSolver's next vector is normalised (the end of
    gram_schmidt, include/solver.hpp:209) by the kernel that applies M^-1 to it (:290-328), where that kernel's
    16-byte-lane form serves the mesh; KRY_TDP_FUSE_SCALE=0 keeps the separate pass.  Both with and without
    the interleaved basis."""
    prob = problems.s_aniso(*shape)
    a = host.ImplicitEuler(ctx, **prob, m=m, tdp=True, **kw)
    with environment("KRY_TDP_FUSE_SCALE"), interleaved_basis():
        b = host.ImplicitEuler(ctx, **prob, m=m, tdp=True, **kw)
        with contiguous_basis():
            c = host.ImplicitEuler(ctx, **prob, m=m, tdp=True, **kw)
    with contiguous_basis():
        d = host.ImplicitEuler(ctx, **prob, m=m, tdp=True, **kw)
    for step in range(3):
        deltas = [x.timestep(0.25) for x in (a, b, c, d)]
        assert deltas[0] == deltas[1] == deltas[2] == deltas[3]
        for x in (b, c, d):
            assert np.array_equal(a.delta(), x.delta(), equal_nan=True)
            assert np.array_equal(a.solver().hessenberg_matrix(), x.solver().hessenberg_matrix(), equal_nan=True)
    for k in sorted({0, 1, m // 2, m - 1, m}):
        va = a.solver().subspace_vector(k)
        for x in (b, c, d):
            assert np.array_equal(va, x.solver().subspace_vector(k), equal_nan=True)
    # interleaved by default where the preconditioner kernel normalises, or from 32 steps per cycle on
    assert a.solver().basis_chunk_interleaved() == (fused or m >= 32)
    assert b.solver().basis_chunk_interleaved()
    assert not c.solver().basis_chunk_interleaved() and not d.solver().basis_chunk_interleaved()
    # launches of the normalisation per cycle: ||b||, v_0, and one per step unless the preconditioner kernel does it
    normalize = capi.K_NAMES.index("normalize")
    for x, f in ((a, fused), (b, False), (c, False), (d, fused)):
        assert x.solver().cycle_bytes(x.operator())[1][normalize] == (3 if f else m + 2)


def test_one_sweep_solves_keep_their_own_layout(ctx):
    dev = host.ImplicitEuler(ctx, **problems.s_iso(64, 48), m=12)
    dev.timestep(0.25)
    assert dev.solver().basis_strip_major()
    assert not dev.solver().basis_chunk_interleaved()


@pytest.mark.parametrize("n,m,tdp", [(1, 1, False), (511, 5, False), (513, 7, True), (4099, 33, False), (20000, 66, True)])
def test_stand_alone_solver_with_a_csr_operator(ctx, n, m, tdp):
    rng = np.random.default_rng(n + m)
    A = sp.random(n, n, density=min(1.0, 6.0 / n), random_state=rng, format="csr") + sp.diags(np.full(n, 4.0))
    if tdp:  # the preconditioner takes the three central diagonals
        A = A + sp.diags([np.full(n - 1, -1.0), np.full(n - 1, -0.5)], [-1, 1])
    A = sp.csr_matrix(A)
    A.sort_indices()
    b = rng.standard_normal(n)
    op = host.Operator.csr(ctx, A.indptr.astype(np.int64), A.indices.astype(np.int32), A.data)
    with interleaved_basis():
        ga = host.GMRESSolver(ctx, n, m, tdp=tdp)
    with contiguous_basis():
        gb = host.GMRESSolver(ctx, n, m, tdp=tdp)
    xa, xb = ga.solve(op, b), gb.solve(op, b)
    assert ga.basis_chunk_interleaved() and not gb.basis_chunk_interleaved()
    assert np.array_equal(xa, xb, equal_nan=True)
    assert np.array_equal(ga.hessenberg_matrix(), gb.hessenberg_matrix(), equal_nan=True)
    for k in sorted({0, m // 2, m}):
        assert np.array_equal(ga.subspace_vector(k), gb.subspace_vector(k), equal_nan=True)
    # the step-wise interface after a solve: contiguous vectors again, same bits as a fresh solver's
    beta = ga.initial_subspace(b)
    assert not ga.basis_chunk_interleaved()
    assert beta == gb.initial_subspace(b)
    if tdp:
        ga.assemble_preconditioner(op)
        gb.assemble_preconditioner(op)
    for _ in range(min(m, 3)):
        assert ga.add_arnoldi(op) == gb.add_arnoldi(op)
    assert np.array_equal(ga.subspace_vector(1), gb.subspace_vector(1), equal_nan=True)
    # and a solve after that is interleaved again
    assert np.array_equal(ga.solve(op, b), xb, equal_nan=True)
    assert ga.basis_chunk_interleaved()
    # the default: from 32 steps per cycle on
    gd = host.GMRESSolver(ctx, n, m, tdp=tdp)
    assert np.array_equal(gd.solve(op, b), xb, equal_nan=True)
    assert gd.basis_chunk_interleaved() == (m >= 32)
