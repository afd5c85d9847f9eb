"""ctypes front-end of the CPU restatement in ``krylov_oracle.c``.

TEST INFRASTRUCTURE ONLY: imported by ``tests/``, by the smoke check and by the
CPU-baseline leg of ``bench.py``.  Nothing in the ``krylov_amd`` package may
import this module.

The wrapper adds no arithmetic of its own except the sampling of the
user-supplied initial-condition and boundary functions at the reference's
coordinates (``include/mesh.hpp:20-42``, ``include/time_disc.hpp:55-76``).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# KRYLOV_ORACLE_OMP=1 selects the thread-parallel timing build (benchmark's all-cores
# CPU baseline only; its reductions sum in a different order, so it is not a checker).
# variant() below binds a second copy of this module to another build of the same source.
_LIB_NAME = globals().get("_FORCED_LIB_NAME") or (
    "libkrylov_oracle_omp.so" if os.environ.get("KRYLOV_ORACLE_OMP") else "libkrylov_oracle.so")
_LIB_PATH = os.path.join(_HERE, _LIB_NAME)
_lib = None

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

BOTTOM, TOP, LEFT, RIGHT = 0, 1, 2, 3
FIELD_T, FIELD_U, FIELD_V = 0, 1, 2


def variant(kind):
    """A second instance of this module bound to another build of krylov_oracle.c:
    "exact" = dot products and norms summed pairwise in extended precision (the
    yardstick that tells which of two double-precision runs is closer to exactly
    summed reductions; never the parity checker), "omp" = the thread-parallel
    timing build."""
    import importlib.util
    import sys
    name = "oracle_variant_" + kind
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.abspath(__file__))
    mod = importlib.util.module_from_spec(spec)
    mod.__dict__["_FORCED_LIB_NAME"] = {"exact": "libkrylov_oracle_exact.so", "omp": "libkrylov_oracle_omp.so"}[kind]
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def build(force=False):
    """Compile the restatement with the committed Makefile."""
    src = os.path.join(_HERE, "krylov_oracle.c")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= os.path.getmtime(src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-s", "-C", _HERE, _LIB_NAME])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)

    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_dense_matmul", None, _dp, _dp, _dp, C.c_long, C.c_long, C.c_long)
    sig("orc_dense_matvec", None, _dp, _dp, _dp, C.c_long, C.c_long)
    sig("orc_csr_matvec", None, _dp, _vp, _vp, _dp, _dp, C.c_long)
    sig("orc_dot", C.c_double, _dp, _dp, C.c_long)
    sig("orc_l2_norm", C.c_double, _dp, C.c_long)
    sig("orc_normalize", C.c_double, _dp, C.c_long)
    sig("orc_lu_decomp", None, _dp, _dp, _dp, C.c_long)
    sig("orc_lu_solve", None, _dp, _dp, _dp, C.c_long, _dp)
    sig("orc_sparse_tridiag_solve", None, _dp, _dp, _dp, _dp, C.c_long)
    sig("orc_tridiag_solve", None, _dp, _dp, _dp, _dp, C.c_long)
    sig("orc_op_dense", _vp, _dp, C.c_long)
    sig("orc_op_csr", _vp, _vp, _vp, _dp, C.c_long)
    sig("orc_op_free", None, _vp)
    sig("orc_op_apply", None, _vp, _dp, _dp)
    sig("orc_gmres_create", _vp, C.c_long, C.c_int, C.c_int)
    sig("orc_gmres_free", None, _vp)
    sig("orc_gmres_set_fixed_preconditioning", None, _vp, C.c_int)
    sig("orc_gmres_initial_subspace", C.c_double, _vp, _dp)
    sig("orc_gmres_assemble_preconditioner", None, _vp, _vp)
    sig("orc_gmres_add_arnoldi", C.c_double, _vp, _vp)
    sig("orc_gmres_solve", _dp, _vp, _vp, _dp)
    sig("orc_gmres_set_extended", None, _vp, C.c_int, C.c_double, C.c_int)
    sig("orc_gmres_last_stats", None, _vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _dp)
    sig("orc_gmres_subspace_vector", _dp, _vp, C.c_int)
    sig("orc_gmres_hessenberg", C.c_double, _vp, C.c_int, C.c_int)
    sig("orc_gmres_hessenberg_data", _dp, _vp)
    sig("orc_gmres_prev_result", _dp, _vp)
    sig("orc_gmres_tdiags", _dp, _vp)
    sig("orc_gmres_reduced_solution", _dp, _vp)
    sig("orc_ie_create", _vp, C.c_double, C.c_double, C.c_double, C.c_double,
        C.c_int, C.c_int, _dp, _dp, _dp, C.c_double, C.c_double, C.c_double,
        C.c_double, C.POINTER(C.c_int), C.c_int, C.c_int)
    sig("orc_ie_free", None, _vp)
    sig("orc_ie_set_bc_values", None, _vp, C.c_int, C.c_int, _dp)
    sig("orc_ie_apply_bcs", None, _vp)
    sig("orc_ie_init_offdiag", None, _vp)
    sig("orc_ie_assemble_system", None, _vp, C.c_double)
    sig("orc_ie_timestep", C.c_double, _vp, C.c_double)
    sig("orc_af_system_dy", None, _vp, C.c_double, C.c_int, _dp)
    sig("orc_af_system_dx", None, _vp, C.c_double, C.c_int, _dp)
    sig("orc_af_timestep", C.c_double, _vp, C.c_double)
    sig("orc_ie_time", C.c_double, _vp)
    sig("orc_ie_num_eqns", C.c_long, _vp)
    sig("orc_ie_nnz", C.c_int64, _vp)
    sig("orc_ie_field", _dp, _vp, C.c_int)
    sig("orc_ie_field_mut", _dp, _vp, C.c_int)
    sig("orc_ie_rowptr", C.POINTER(C.c_int64), _vp)
    sig("orc_ie_col", C.POINTER(C.c_int32), _vp)
    sig("orc_ie_val", _dp, _vp)
    sig("orc_ie_sol_vec", _dp, _vp)
    sig("orc_ie_solver", _vp, _vp)
    sig("orc_ie_op", _vp, _vp)
    sig("orc_ie_dx", C.c_double, _vp)
    sig("orc_ie_dy", C.c_double, _vp)
    sig("orc_cell_idx", C.c_long, _vp, C.c_int, C.c_int)
    sig("orc_flux_integral", C.c_double, _vp, C.c_int, C.c_int)
    sig("orc_source_fd", C.c_double, _vp, C.c_int, C.c_int)
    sig("orc_ghost_cell", C.c_double, C.c_double, C.c_double, C.c_double, C.c_double)
    sig("orc_eval_cells", C.c_int, _vp, C.c_int, _dp)
    for nm in ("orc_left_x", "orc_right_x", "orc_bottom_y", "orc_top_y",
               "orc_median_x", "orc_median_y"):
        sig(nm, C.c_double, _vp, C.c_int)
    _lib = L
    return L


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _view(ptr, n):
    return np.ctypeslib.as_array(ptr, shape=(int(n),))


# ---- include/matrix.hpp -------------------------------------------------

def dense_matvec(a, x):
    a, pa = _d(a)
    x, px = _d(x)
    y = np.empty(a.shape[0])
    lib().orc_dense_matvec(y.ctypes.data_as(_dp), pa, px, a.shape[0], a.shape[1])
    return y


def dense_matmul(a, b):
    a, pa = _d(a)
    b, pb = _d(b)
    r = np.empty((a.shape[0], b.shape[1]))
    lib().orc_dense_matmul(r.ctypes.data_as(_dp), pa, pb, a.shape[0], a.shape[1], b.shape[1])
    return r


def csr_matvec(rowptr, col, val, x):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val, pv = _d(val)
    x, px = _d(x)
    n = rowptr.shape[0] - 1
    y = np.empty(n)
    lib().orc_csr_matvec(y.ctypes.data_as(_dp), rowptr.ctypes.data, col.ctypes.data, pv, px, n)
    return y


def dot(a, b):
    a, pa = _d(a)
    b, pb = _d(b)
    return lib().orc_dot(pa, pb, a.shape[0])


def l2_norm(a):
    a, pa = _d(a)
    return lib().orc_l2_norm(pa, a.shape[0])


def normalize(a):
    a = np.array(a, dtype=np.float64)
    beta = lib().orc_normalize(a.ctypes.data_as(_dp), a.shape[0])
    return a, beta


# ---- include/solver.hpp: direct solvers ----------------------------------

def lu_decomp(sys):
    sys, ps = _d(sys)
    n = sys.shape[0]
    l = np.empty((n, n))
    u = np.empty((n, n))
    lib().orc_lu_decomp(ps, l.ctypes.data_as(_dp), u.ctypes.data_as(_dp), n)
    return l, u


def lu_solve(sys, rhs):
    sys, ps = _d(sys)
    rhs, pr = _d(rhs)
    n = sys.shape[0]
    x = np.empty(n)
    work = np.empty(2 * n * n + n)
    lib().orc_lu_solve(ps, pr, x.ctypes.data_as(_dp), n, work.ctypes.data_as(_dp))
    return x


def sparse_tridiag_solve(diags, sol):
    diags, pd = _d(diags)
    sol, ps = _d(sol)
    n = sol.shape[0]
    x = np.empty(n)
    ud = np.empty(max(n, 1))
    lib().orc_sparse_tridiag_solve(pd, ps, x.ctypes.data_as(_dp), ud.ctypes.data_as(_dp), n)
    return x


def tridiag_solve(sys, sol):
    sys, pm = _d(sys)
    sol, ps = _d(sol)
    n = sol.shape[0]
    x = np.empty(n)
    ud = np.empty(max(n, 1))
    lib().orc_tridiag_solve(pm, ps, x.ctypes.data_as(_dp), ud.ctypes.data_as(_dp), n)
    return x


# ---- operators and GMRES --------------------------------------------------

class Op:
    """Dense (row-major n x n) or CSR operator; keeps its arrays alive."""

    def __init__(self, handle, n, keep):
        self.h = handle
        self.n = n
        self._keep = keep

    @classmethod
    def dense(cls, a):
        a, pa = _d(a)
        return cls(lib().orc_op_dense(pa, a.shape[0]), a.shape[0], (a,))

    @classmethod
    def csr(cls, rowptr, col, val):
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
        col = np.ascontiguousarray(col, dtype=np.int32)
        val, pv = _d(val)
        n = rowptr.shape[0] - 1
        return cls(lib().orc_op_csr(rowptr.ctypes.data, col.ctypes.data, pv, n), n,
                   (rowptr, col, val))

    def apply(self, x):
        x, px = _d(x)
        y = np.empty(self.n)
        lib().orc_op_apply(self.h, y.ctypes.data_as(_dp), px)
        return y

    def __del__(self):
        if self._keep is not None and self.h:
            lib().orc_op_free(self.h)
            self.h = None


class Gmres:
    """GMRESSolver<m> (tdp=False) / GMRESTDPSolver<m> (tdp=True)."""

    def __init__(self, n, m, tdp=False, handle=None):
        self.n, self.m, self.tdp = n, m, tdp
        self._own = handle is None
        self.h = lib().orc_gmres_create(n, m, int(tdp)) if handle is None else handle

    def set_fixed_preconditioning(self, on):
        lib().orc_gmres_set_fixed_preconditioning(self.h, int(on))

    def set_extended(self, rtol, max_cycles, on=True):
        """Restarted GMRES(m) with Givens rotations and a residual monitor
        (the device library's extended mode; not in the reference)."""
        lib().orc_gmres_set_extended(self.h, int(on), float(rtol), int(max_cycles))

    def last_stats(self):
        it, cy, cv, rr = C.c_int(), C.c_int(), C.c_int(), C.c_double()
        lib().orc_gmres_last_stats(self.h, C.byref(it), C.byref(cy), C.byref(cv), C.byref(rr))
        return it.value, cy.value, bool(cv.value), rr.value

    def initial_subspace(self, b):
        b, pb = _d(b)
        return lib().orc_gmres_initial_subspace(self.h, pb)

    def assemble_preconditioner(self, op):
        lib().orc_gmres_assemble_preconditioner(self.h, op.h)
        return self.tdiags()

    def add_arnoldi(self, op):
        return lib().orc_gmres_add_arnoldi(self.h, op.h)

    def solve(self, op, b):
        b, pb = _d(b)
        p = lib().orc_gmres_solve(self.h, op.h, pb)
        return _view(p, self.n).copy()

    def subspace_vector(self, i):
        return _view(lib().orc_gmres_subspace_vector(self.h, i), self.n).copy()

    def hessenberg(self, i, j):
        return lib().orc_gmres_hessenberg(self.h, i, j)

    def hessenberg_matrix(self):
        return _view(lib().orc_gmres_hessenberg_data(self.h),
                     (self.m + 1) * self.m).reshape(self.m + 1, self.m).copy()

    def prev_result(self):
        return _view(lib().orc_gmres_prev_result(self.h), self.n).copy()

    def tdiags(self):
        return _view(lib().orc_gmres_tdiags(self.h), 3 * self.n).reshape(self.n, 3).copy()

    def reduced_solution(self):
        return _view(lib().orc_gmres_reduced_solution(self.h), self.m).copy()

    def __del__(self):
        if self._own and self.h:
            lib().orc_gmres_free(self.h)
            self.h = None


# ---- the implicit-Euler step ------------------------------------------------

class ImplicitEuler:
    """ImplicitEuler<SecondOrderCentered, GMRES[TDP]Solver<m>>.

    ``init`` maps arrays (x, y) to (T, u, v); ``bcs`` is a dict keyed by
    (field, side) holding (is_dirichlet, fn(x, y, t)); missing entries are the
    reference's default, homogeneous Dirichlet (include/time_disc.hpp:23-50).
    Functions are evaluated on numpy arrays (or broadcast scalars).
    """

    def __init__(self, corner1, corner2, cx, cy, init, params, bcs, m, tdp=False, extended=None):
        L = lib()
        self.cx, self.cy, self.m, self.tdp = cx, cy, m, tdp
        self.bcs = {}
        flags = (C.c_int * 12)()
        for f in range(3):
            for s in range(4):
                is_d, fn = bcs.get((f, s), (True, None))
                self.bcs[(f, s)] = (bool(is_d), fn)
                flags[f * 4 + s] = int(bool(is_d))
        min_x = min(corner1[0], corner2[0])
        max_x = max(corner1[0], corner2[0])
        min_y = min(corner1[1], corner2[1])
        max_y = max(corner1[1], corner2[1])
        dx = (max_x - min_x) / float(cx)
        dy = (max_y - min_y) / float(cy)
        self.min_x, self.min_y, self.dx, self.dy = min_x, min_y, dx, dy
        xs = self.median_x(np.arange(cx))
        ys = self.median_y(np.arange(cy))
        T0, u0, v0 = init(xs[:, None], ys[None, :])
        T0, u0, v0 = (np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (cx, cy)))
                      for a in (T0, u0, v0))
        self.h = L.orc_ie_create(corner1[0], corner1[1], corner2[0], corner2[1], cx, cy,
                                 T0.ctypes.data_as(_dp), u0.ctypes.data_as(_dp),
                                 v0.ctypes.data_as(_dp), params[0], params[1], params[2],
                                 params[3], flags, m, int(tdp))
        assert L.orc_ie_dx(self.h) == dx and L.orc_ie_dy(self.h) == dy
        self.n = L.orc_ie_num_eqns(self.h)
        self._push_bc_values(0.0)
        L.orc_ie_init_offdiag(self.h)
        if extended is not None:
            self.solver().set_extended(*extended)

    # include/mesh.hpp:20-42 (same operation order)
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
        for (f, s), (_, fn) in self.bcs.items():
            x, y = self.bc_coords(s)
            vals = np.zeros(x.shape[0]) if fn is None else np.broadcast_to(
                np.asarray(fn(x, y, t), dtype=np.float64), x.shape)
            vals = np.ascontiguousarray(vals)
            lib().orc_ie_set_bc_values(self.h, f, s, vals.ctypes.data_as(_dp))

    def cur_time(self):
        return lib().orc_ie_time(self.h)

    def apply_bcs(self):
        self._push_bc_values(self.cur_time())
        lib().orc_ie_apply_bcs(self.h)

    def assemble_system(self, dt):
        self._push_bc_values(self.cur_time())
        lib().orc_ie_assemble_system(self.h, dt)

    def timestep(self, dt):
        self._push_bc_values(self.cur_time())
        return lib().orc_ie_timestep(self.h, dt)

    def cell_idx(self, i, j):
        return lib().orc_cell_idx(self.h, i, j)

    def field(self, f):
        return _view(lib().orc_ie_field(self.h, f), (self.cx + 2) * (self.cy + 2)).reshape(
            self.cx + 2, self.cy + 2).copy()

    def set_field(self, f, arr):
        dst = _view(lib().orc_ie_field_mut(self.h, f), (self.cx + 2) * (self.cy + 2))
        dst[:] = np.asarray(arr, dtype=np.float64).reshape(-1)

    def csr(self):
        L = lib()
        nnz = L.orc_ie_nnz(self.h)
        rowptr = np.ctypeslib.as_array(L.orc_ie_rowptr(self.h), shape=(self.n + 1,)).copy()
        col = np.ctypeslib.as_array(L.orc_ie_col(self.h), shape=(nnz,)).copy()
        val = _view(L.orc_ie_val(self.h), nnz).copy()
        return rowptr, col, val

    def dense_system(self):
        rowptr, col, val = self.csr()
        a = np.zeros((self.n, self.n))
        for r in range(self.n):
            a[r, col[rowptr[r]:rowptr[r + 1]]] = val[rowptr[r]:rowptr[r + 1]]
        return a

    def sol_vec(self):
        return _view(lib().orc_ie_sol_vec(self.h), self.n).copy()

    def delta(self):
        return self.solver().prev_result()

    def solver(self):
        return Gmres(self.n, self.m, self.tdp, handle=lib().orc_ie_solver(self.h))

    def op(self):
        return Op(lib().orc_ie_op(self.h), self.n, None)

    # member functions of SecondOrderCentered / EnergyAssembly (include/space_disc.hpp)
    QUANTITIES = ("Dx_m1", "Dx_0", "Dx_p1", "Dy_m1", "Dy_0", "Dy_p1", "dx_flux", "dy_flux",
                  "lapl_T_flux_integral", "flux_integral", "du_dx_fd", "dv_dy_fd", "du_dy_fd", "dv_dx_fd",
                  "source_fd")

    def eval_cells(self, name):
        """One of QUANTITIES on every interior cell: array (cx, cy)."""
        out = np.empty((self.cx, self.cy))
        rc = lib().orc_eval_cells(self.h, self.QUANTITIES.index(name), out.ctypes.data_as(_dp))
        assert rc == 0
        return out

    def flux_integral(self, i, j):
        return lib().orc_flux_integral(self.h, i, j)

    def source_fd(self, i, j):
        return lib().orc_source_fd(self.h, i, j)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_ie_free(self.h)
            self.h = None


class ApproxFact(ImplicitEuler):
    """ImplicitEulerApproxFact<SecondOrderCentered> (include/time_disc.hpp:388-540)."""

    def __init__(self, corner1, corner2, cx, cy, init, params, bcs):
        super().__init__(corner1, corner2, cx, cy, init, params, bcs, m=1)
        self.apply_bcs()  # include/time_disc.hpp:440

    def timestep(self, dt):
        self._push_bc_values(self.cur_time())
        return lib().orc_af_timestep(self.h, dt)

    def assemble_system_dy(self, dt, i):
        out = np.empty((self.cy + 2, 3))
        lib().orc_af_system_dy(self.h, dt, i, out.ctypes.data_as(_dp))
        return out

    def assemble_system_dx(self, dt, j):
        out = np.empty((self.cx + 2, 3))
        lib().orc_af_system_dx(self.h, dt, j, out.ctypes.data_as(_dp))
        return out
