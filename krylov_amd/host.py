"""Python host mirror of the reference's solver / time-discretisation interface
on top of the C ABI (``include/krylov_amd.h``).

Used by the GPU tests and the benchmark.  The class and method names follow
the reference (``GMRESSolver``: include/solver.hpp:115-239, ``ImplicitEuler``:
include/time_disc.hpp:190-386).  Everything numerical happens on the device;
this module only samples the user's initial-condition and boundary functions
at the reference's coordinates (include/mesh.hpp:20-42,
include/time_disc.hpp:55-76) and moves arrays in and out.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import (BOTTOM, TOP, LEFT, RIGHT, FIELD_T, FIELD_U, FIELD_V,  # noqa: F401
                   GMRES_PLAIN, GMRES_TDP, GMRES_FIXED_PRECOND, check)

_dp = C.POINTER(C.c_double)


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


class Context:
    """Device + stream.  Raises if no HIP device is visible."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        check(capi.lib().kry_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h)))
        self.h = h

    def memory_info(self):
        """(free, total) bytes of the device."""
        f, t = C.c_int64(), C.c_int64()
        check(capi.lib().kry_ctx_memory_info(self.h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def synchronize(self):
        check(capi.lib().kry_ctx_synchronize(self.h))

    @property
    def stream(self):
        return capi.lib().kry_ctx_stream(self.h)

    def set_kernel_timing(self, on):
        check(capi.lib().kry_ctx_set_kernel_timing(self.h, int(on)))

    def kernel_times(self):
        """(milliseconds, launches) per kernel class since the last call."""
        ms = (C.c_double * capi.K_COUNT)()
        cnt = (C.c_int * capi.K_COUNT)()
        check(capi.lib().kry_ctx_get_kernel_times(self.h, ms, cnt))
        return list(ms), list(cnt)

    def selftest_shared_division(self, dx, dy, seed, count, mode):
        """(quotients that differ from the device's division, groups of four served by the shortcut)."""
        bad, fast = C.c_int64(), C.c_int64()
        check(capi.lib().kry_selftest_shared_division(self.h, float(dx), float(dy), int(seed), int(count), int(mode),
                                                      C.byref(bad), C.byref(fast)))
        return bad.value, fast.value

    def __del__(self):
        if getattr(self, "h", None):
            capi.lib().kry_ctx_destroy(self.h)
            self.h = None


class Operator:
    """The `matrix` argument of Solver::solve, resident on the device."""

    def __init__(self, ctx, handle, own=True):
        self.ctx, self.h, self._own = ctx, handle, own
        self.n = capi.lib().kry_op_size(handle)

    @classmethod
    def dense(cls, ctx, a):
        a, pa = _d(a)
        assert a.ndim == 2 and a.shape[0] == a.shape[1]
        h = C.c_void_p()
        check(capi.lib().kry_op_create_dense(ctx.h, a.shape[0], pa, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def csr(cls, ctx, rowptr, col, val):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val, pv = _d(val)
        h = C.c_void_p()
        check(capi.lib().kry_op_create_csr(ctx.h, rowptr.shape[0] - 1, rowptr.ctypes.data,
                                           col.ctypes.data, pv, C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def stencil(cls, ctx, w, s, c, n, e):
        arrs = [_d(a) for a in (w, s, c, n, e)]
        rows, cols = arrs[0][0].shape
        h = C.c_void_p()
        check(capi.lib().kry_op_create_stencil(ctx.h, rows, cols, *[p for _, p in arrs], C.byref(h)))
        return cls(ctx, h)

    def apply(self, x):
        x, px = _d(x)
        y = np.empty(self.n)
        check(capi.lib().kry_op_apply(self.h, px, y.ctypes.data_as(_dp)))
        return y

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "h", None):
            capi.lib().kry_op_destroy(self.h)
            self.h = None


class GMRESSolver:
    """GMRESSolver<m> (tdp=False) / GMRESTDPSolver<m> (tdp=True)."""

    def __init__(self, ctx, n, m, tdp=False, fixed_preconditioning=False, handle=None, extended=None):
        """extended=(rtol, max_cycles): restarted GMRES(m) with Givens rotations
        and a residual monitor instead of the reference's single fixed cycle."""
        self.ctx, self.n, self.m, self.tdp = ctx, n, m, tdp
        self._own = handle is None
        if handle is None:
            flags = ((GMRES_TDP if tdp else 0) | (GMRES_FIXED_PRECOND if fixed_preconditioning else 0)
                     | (capi.GMRES_EXTENDED if extended is not None else 0))
            h = C.c_void_p()
            check(capi.lib().kry_gmres_create(ctx.h, n, m, flags, C.byref(h)))
            handle = h
        self.h = handle
        if extended is not None:
            self.set_tolerance(*extended)

    def set_tolerance(self, rtol, max_cycles):
        check(capi.lib().kry_gmres_set_tolerance(self.h, float(rtol), int(max_cycles)))

    def last_stats(self):
        """(Arnoldi steps, cycles, converged, residual / ||b||) of the last extended solve."""
        it, cy, cv, rr = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        check(capi.lib().kry_gmres_last_stats(self.h, C.byref(it), C.byref(cy), C.byref(cv), C.byref(rr)))
        return it.value, cy.value, bool(cv.value), rr.value

    def initial_subspace(self, b):
        b, pb = _d(b)
        beta = C.c_double()
        check(capi.lib().kry_gmres_initial_subspace(self.h, pb, C.byref(beta)))
        return beta.value

    def assemble_preconditioner(self, op):
        t = np.empty((self.n, 3))
        check(capi.lib().kry_gmres_assemble_preconditioner(self.h, op.h, t.ctypes.data_as(_dp)))
        return t

    def apply_preconditioner(self, v):
        """M^-1 v for the assembled tridiagonal (operator layout)."""
        v, pv = _d(v)
        out = np.empty(self.n)
        check(capi.lib().kry_gmres_apply_preconditioner(self.h, pv, out.ctypes.data_as(_dp)))
        return out

    def add_arnoldi(self, op):
        h = C.c_double()
        check(capi.lib().kry_gmres_add_arnoldi(self.h, op.h, C.byref(h)))
        return h.value

    def solve(self, op, b):
        b, pb = _d(b)
        x = np.empty(self.n)
        check(capi.lib().kry_gmres_solve(self.h, op.h, pb, x.ctypes.data_as(_dp)))
        return x

    def solve_device(self, op, b_ptr, x_ptr=None):
        check(capi.lib().kry_gmres_solve_device(self.h, op.h, b_ptr, x_ptr))

    def subspace_vector(self, i):
        v = np.empty(self.n)
        check(capi.lib().kry_gmres_subspace_vector(self.h, i, v.ctypes.data_as(_dp)))
        return v

    def hessenberg(self, i, j):
        h = C.c_double()
        check(capi.lib().kry_gmres_hessenberg(self.h, i, j, C.byref(h)))
        return h.value

    def hessenberg_matrix(self):
        H = np.empty((self.m + 1, self.m))
        check(capi.lib().kry_gmres_hessenberg_matrix(self.h, H.ctypes.data_as(_dp)))
        return H

    def reduced_solve(self, H, beta):
        """y of the reduced least-squares problem for a given (m+1) x m Hessenberg array."""
        H, pH = _d(H)
        assert H.shape == (self.m + 1, self.m)
        y = np.empty(self.m)
        check(capi.lib().kry_gmres_reduced_solve(self.h, pH, float(beta), y.ctypes.data_as(_dp)))
        return y

    def prev_result(self):
        x = np.empty(self.n)
        check(capi.lib().kry_gmres_prev_result(self.h, x.ctypes.data_as(_dp)))
        return x

    def set_mgs_block(self, k):
        check(capi.lib().kry_gmres_set_mgs_block(self.h, k))

    def get_mgs_block(self):
        return capi.lib().kry_gmres_get_mgs_block(self.h)

    def set_one_sweep(self, first_step):
        """First Arnoldi step of a cycle that uses the one-sweep kernels (< 0: never)."""
        check(capi.lib().kry_gmres_set_one_sweep(self.h, first_step))

    def get_one_sweep(self):
        return capi.lib().kry_gmres_get_one_sweep(self.h)

    def basis_strip_major(self):
        """Whether the basis of the last solve was stored strip by strip."""
        return bool(capi.lib().kry_gmres_basis_strip_major(self.h))

    def basis_chunk_interleaved(self):
        """Whether the basis of the last solve was stored chunk-interleaved (chain-only solves on one rank)."""
        return bool(capi.lib().kry_gmres_basis_chunk_interleaved(self.h))

    def set_norm_from_sums(self, on):
        """H(m, m-1) from the last sweep's sums (default for a step object's solver) or from the closing pass."""
        check(capi.lib().kry_gmres_set_norm_from_sums(self.h, int(bool(on))))

    def get_norm_from_sums(self):
        return bool(capi.lib().kry_gmres_get_norm_from_sums(self.h))

    def last_norm_from_sums(self):
        """Whether the last solve's final cycle took H(m, m-1) from sums (False: it formed the vector)."""
        u = C.c_int()
        check(capi.lib().kry_gmres_last_norm_from_sums(self.h, C.byref(u)))
        return bool(u.value)

    def cycle_bytes(self, op):
        b = (C.c_double * capi.K_COUNT)()
        l = (C.c_int * capi.K_COUNT)()
        check(capi.lib().kry_gmres_cycle_bytes(self.h, op.h, b, l))
        return list(b), list(l)

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "h", None):
            capi.lib().kry_gmres_destroy(self.h)
            self.h = None


class ImplicitEuler:
    """ImplicitEuler<SecondOrderCentered, GMRES[TDP]Solver<m>> on the device.

    ``init(x, y) -> (T, u, v)`` and the boundary functions ``fn(x, y, t)`` are
    evaluated on numpy arrays.  ``bcs[(field, side)] = (is_dirichlet, fn)``;
    missing entries are homogeneous Dirichlet (include/time_disc.hpp:23-50).
    Set ``bc_time_dependent=False`` to sample the boundary functions only once.
    """

    def __init__(self, ctx, corner1, corner2, cx, cy, init, params, bcs, m, tdp=False,
                 fixed_preconditioning=False, use_graph=False, bc_time_dependent=True, mgs_block=None,
                 row_range=None, extended=None, one_sweep=None):
        """row_range=(begin, end): own only these padded rows of the mesh (one
        rank of a row-block decomposition; the context needs collectives).
        extended=(rtol, max_cycles): solve every step with the restarted,
        residual-monitored GMRES(m) instead of the reference's fixed cycle.
        one_sweep: first Arnoldi step served by the one-sweep kernels (< 0: none;
        None: the library's default)."""
        L = capi.lib()
        self.ctx, self.cx, self.cy, self.m, self.tdp = ctx, cx, cy, m, tdp
        self.row_range = (0, cx + 2) if row_range is None else (int(row_range[0]), int(row_range[1]))
        self.bc_time_dependent = bc_time_dependent
        self.bcs = {}
        flags = (C.c_int * 12)()
        for f in range(3):
            for s in range(4):
                is_d, fn = bcs.get((f, s), (True, None))
                self.bcs[(f, s)] = (bool(is_d), fn)
                flags[f * 4 + s] = int(bool(is_d))
        # src/mesh.cpp:11-15
        self.min_x = min(corner1[0], corner2[0])
        max_x = max(corner1[0], corner2[0])
        self.min_y = min(corner1[1], corner2[1])
        max_y = max(corner1[1], corner2[1])
        self.dx = (max_x - self.min_x) / float(cx)
        self.dy = (max_y - self.min_y) / float(cy)
        r0, r1 = self.row_range
        ib, ie = max(r0, 1) - 1, min(r1, cx + 1) - 1  # interior cell rows owned
        self.interior_rows = (ib, max(ie, ib))
        xs = self.median_x(np.arange(ib, max(ie, ib)))
        ys = self.median_y(np.arange(cy))
        T0, u0, v0 = init(xs[:, None], ys[None, :])
        T0, u0, v0 = (np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (len(xs), cy)))
                      for a in (T0, u0, v0))
        gflags = ((GMRES_TDP if tdp else 0) | (GMRES_FIXED_PRECOND if fixed_preconditioning else 0)
                  | (capi.GMRES_EXTENDED if extended is not None else 0))
        h = C.c_void_p()
        check(L.kry_ie_create_slab(ctx.h, corner1[0], corner1[1], corner2[0], corner2[1], cx, cy, r0, r1,
                                   T0.ctypes.data_as(_dp), u0.ctypes.data_as(_dp), v0.ctypes.data_as(_dp),
                                   params[0], params[1], params[2], params[3], flags, m, gflags, C.byref(h)))
        self.h = h
        self.n = L.kry_ie_num_eqns(h)
        self.npad = (cx + 2) * (cy + 2)
        self.nloc = L.kry_ie_local_size(h)
        if use_graph:
            check(L.kry_ie_use_graph(h, 1))
        if mgs_block is not None and m > 0:
            check(L.kry_gmres_set_mgs_block(L.kry_ie_solver(h), mgs_block))
        if extended is not None:
            check(L.kry_gmres_set_tolerance(L.kry_ie_solver(h), float(extended[0]), int(extended[1])))
        if one_sweep is not None and m > 0:
            check(L.kry_gmres_set_one_sweep(L.kry_ie_solver(h), int(one_sweep)))
        self._bc_pushed = False
        self._push_bc_values(0.0)
        check(L.kry_ie_init_offdiag(h))

    # include/mesh.hpp:20-42, same operation order
    def left_x(self, i):
        return self.min_x + i * self.dx

    def right_x(self, i):
        return self.min_x + (i + 1) * self.dx

    def bottom_y(self, j):
        return self.min_y + j * self.dy

    def top_y(self, j):
        return self.min_y + (j + 1) * self.dy

    def median_x(self, i):
        return self.min_x + self.dx / 2 + i * self.dx

    def median_y(self, j):
        return self.min_y + self.dy / 2 + j * self.dy

    def bc_coords(self, side):
        """Boundary-cell coordinates of include/time_disc.hpp:55-76."""
        cx, cy = self.cx, self.cy
        if side == BOTTOM:
            return self.median_x(np.arange(cx)), np.full(cx, self.bottom_y(0))
        if side == TOP:
            return self.median_x(np.arange(cx)), np.full(cx, self.top_y(cy - 1))
        if side == LEFT:
            return np.full(cy, self.left_x(0)), self.median_y(np.arange(cy))
        return np.full(cy, self.right_x(cx - 1)), self.median_y(np.arange(cy))

    def _push_bc_values(self, t):
        if self._bc_pushed and not self.bc_time_dependent:
            return
        for (f, s), (_, fn) in self.bcs.items():
            x, y = self.bc_coords(s)
            vals = np.zeros(x.shape[0]) if fn is None else np.broadcast_to(
                np.asarray(fn(x, y, t), dtype=np.float64), x.shape)
            vals = np.ascontiguousarray(vals)
            check(capi.lib().kry_ie_set_bc_values(self.h, f, s, vals.ctypes.data_as(_dp), vals.shape[0]))
        self._bc_pushed = True

    def cur_time(self):
        return capi.lib().kry_ie_time(self.h)

    def use_graph(self, on):
        """Replay the step through a captured hipGraph (True) or launch its kernels eagerly."""
        check(capi.lib().kry_ie_use_graph(self.h, int(bool(on))))

    def set_generated_bands(self, on):
        """One-sweep steps evaluate the interior band entries from u and v (default) or read the stored bands."""
        check(capi.lib().kry_ie_set_generated_bands(self.h, int(bool(on))))

    def generated_bands(self):
        return bool(capi.lib().kry_ie_get_generated_bands(self.h))

    def apply_bcs(self):
        self._push_bc_values(self.cur_time())
        check(capi.lib().kry_ie_apply_bcs(self.h))

    def assemble_system(self, dt):
        self._push_bc_values(self.cur_time())
        check(capi.lib().kry_ie_assemble_system(self.h, dt))

    def timestep(self, dt):
        self._push_bc_values(self.cur_time())
        d = C.c_double()
        check(capi.lib().kry_ie_timestep(self.h, dt, C.byref(d)))
        return d.value

    def apply_delta(self, delta_local):
        """T += delta on interior cells and the step's maximum for a given delta
        (owned rows, padded layout): the tail of ImplicitEuler::timestep."""
        a = np.ascontiguousarray(delta_local, dtype=np.float64).reshape(-1)
        assert a.shape[0] == self.nloc
        d = C.c_double()
        check(capi.lib().kry_ie_apply_delta(self.h, a.ctypes.data_as(_dp), C.byref(d)))
        return d.value

    def timestep_async(self, dt):
        self._push_bc_values(self.cur_time())
        check(capi.lib().kry_ie_timestep_async(self.h, dt))

    def last_max_delta(self):
        d = C.c_double()
        check(capi.lib().kry_ie_last_max_delta(self.h, C.byref(d)))
        return d.value

    def field(self, f):
        a = np.empty((self.cx + 2, self.cy + 2))
        check(capi.lib().kry_ie_get_field(self.h, f, a.ctypes.data_as(_dp)))
        return a

    def local(self, what):
        """Owned rows of a quantity (capi.LOCAL_*), shape (rows, cy + 2)."""
        a = np.empty((self.row_range[1] - self.row_range[0], self.cy + 2))
        check(capi.lib().kry_ie_get_local(self.h, what, a.ctypes.data_as(_dp)))
        return a

    def bands(self):
        a = np.empty((5, self.cx + 2, self.cy + 2))
        check(capi.lib().kry_ie_get_bands(self.h, a.ctypes.data_as(_dp)))
        return a

    def dense_system(self):
        a = np.empty((self.n, self.n))
        check(capi.lib().kry_ie_get_dense_system(self.h, a.ctypes.data_as(_dp)))
        return a

    def sol_vec(self):
        a = np.empty(self.n)
        check(capi.lib().kry_ie_get_rhs(self.h, a.ctypes.data_as(_dp)))
        return a

    def delta(self):
        a = np.empty(self.n)
        check(capi.lib().kry_ie_get_delta(self.h, a.ctypes.data_as(_dp)))
        return a

    def solver(self):
        return GMRESSolver(self.ctx, self.nloc, self.m, self.tdp,
                           handle=C.c_void_p(capi.lib().kry_ie_solver(self.h)))

    def operator(self):
        return Operator(self.ctx, C.c_void_p(capi.lib().kry_ie_operator(self.h)), own=False)

    def rhs_device(self):
        return capi.lib().kry_ie_rhs_device(self.h)

    def compact(self, padded):
        """Padded-layout vector -> the reference's cell numbering (drops the corners)."""
        p = np.asarray(padded).reshape(self.cx + 2, self.cy + 2)
        return np.concatenate([p[0, 1:-1], p[1:-1, :].reshape(-1), p[-1, 1:-1]])

    def expand(self, compact):
        """Cell-numbered vector -> padded layout with zero corners."""
        cx, cy = self.cx, self.cy
        c = np.asarray(compact)
        p = np.zeros((cx + 2, cy + 2))
        p[0, 1:-1] = c[:cy]
        p[1:-1, :] = c[cy:cy + cx * (cy + 2)].reshape(cx, cy + 2)
        p[-1, 1:-1] = c[cy + cx * (cy + 2):]
        return p.reshape(-1)

    def __del__(self):
        if getattr(self, "h", None):
            capi.lib().kry_ie_destroy(self.h)
            self.h = None


class ApproxFact(ImplicitEuler):
    """ImplicitEulerApproxFact<SecondOrderCentered> (include/time_disc.hpp:388-540)
    on the device: per step one tridiagonal sweep along y and one along x."""

    def __init__(self, ctx, corner1, corner2, cx, cy, init, params, bcs, bc_time_dependent=True):
        super().__init__(ctx, corner1, corner2, cx, cy, init, params, bcs, m=0,
                         bc_time_dependent=bc_time_dependent)
        self.apply_bcs()  # include/time_disc.hpp:440

    def timestep(self, dt):
        self._push_bc_values(self.cur_time())
        d = C.c_double()
        check(capi.lib().kry_ie_timestep_approx_fact(self.h, dt, C.byref(d)))
        return d.value

    def _system(self, along_y, index, dt):
        self.apply_bcs()
        out = np.empty(((self.cy if along_y else self.cx) + 2, 3))
        check(capi.lib().kry_ie_get_af_system(self.h, int(along_y), index, dt, out.ctypes.data_as(_dp)))
        return out

    def assemble_system_dy(self, dt, i):
        return self._system(True, i, dt)

    def assemble_system_dx(self, dt, j):
        return self._system(False, j, dt)
