/*
 * krylov_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A single-threaded CPU restatement, in plain C, of the implicit-Euler / GMRES
 * path of mfdeakin/krylov.  It exists so that the HIP implementation can be
 * checked against the reference's arithmetic; it is never part of the product
 * path and must only be used by tests/, the smoke check and the CPU-baseline
 * leg of the benchmark.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).  Arithmetic is written in the reference's operation
 * order; build with -ffp-contract=off and without -ffast-math so that no
 * multiply-add is fused (the reference is built as ISO C++17 for baseline
 * x86-64, which has neither FMA contraction nor FMA instructions).
 *
 * Index spaces
 *   padded : fields T,u,v are (cx+2) x (cy+2) row-major, j fastest; interior
 *            cell (i,j) lives at ((i+1),(j+1))          (include/mesh.hpp:77-120)
 *   cell   : the unknown numbering of ImplicitEuler::cell_idx
 *            (include/time_disc.hpp:341-376), N = (cx+2)(cy+2)-4 unknowns.
 *
 * The reference operator is a dense N x N matrix whose rows hold at most five
 * non-zeros.  The dense row sum of include/matrix.hpp:23-33 runs over columns
 * in ascending order starting from +0.0; adding the products of exact zeros
 * does not change a finite partial sum, so a CSR row sum with ascending column
 * indices is bit-identical for finite inputs.  Both forms are provided here
 * and the test-suite checks the claim.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

static double orc_snan(void) {
  /* std::numeric_limits<double>::signaling_NaN(), include/constants.hpp:18 */
  union { uint64_t u; double d; } x;
  x.u = 0x7ff4000000000000ULL;
  return x.d;
}

/* ------------------------------------------------------------------------ */
/* include/matrix.hpp                                                        */
/* ------------------------------------------------------------------------ */

/* dot(matrix&, const matrix&, const matrix&), include/matrix.hpp:8-21 */
ORC_API void orc_dense_matmul(double *res, const double *lhs, const double *rhs,
                              long n, long k, long m) {
  for (long i = 0; i < n; i++) {
    for (long j = 0; j < m; j++) {
      double acc = 0.0;
      for (long p = 0; p < k; p++) {
        acc += lhs[i * k + p] * rhs[p * m + j];
      }
      res[i * m + j] = acc;
    }
  }
}

/* Timing-only variant: with -DORC_OPENMP -fopenmp the vector-length loops of
 * the GMRES path run on all threads (libkrylov_oracle_omp.so, used for the
 * all-cores CPU baseline of the benchmark).  Reductions then sum in a different
 * order, so that build is never used as the parity checker. */
#ifdef ORC_OPENMP
#define ORC_PAR_FOR _Pragma("omp parallel for schedule(static)")
#define ORC_PAR_SUM(var) _Pragma("omp parallel for schedule(static) reduction(+ : acc)")
#else
#define ORC_PAR_FOR
#define ORC_PAR_SUM(var)
#endif

/* dot(vector&, const matrix&, const vector&), include/matrix.hpp:23-33 */
ORC_API void orc_dense_matvec(double *res, const double *a, const double *x,
                              long rows, long cols) {
  for (long i = 0; i < rows; i++) {
    double acc = 0.0;
    for (long k = 0; k < cols; k++) {
      acc += a[i * cols + k] * x[k];
    }
    res[i] = acc;
  }
}

/* Sparse form of include/matrix.hpp:23-33: same ascending-column row sum,
 * restricted to the stored entries. */
ORC_API void orc_csr_matvec(double *res, const int64_t *rowptr,
                            const int32_t *col, const double *val,
                            const double *x, long n) {
  ORC_PAR_FOR
  for (long i = 0; i < n; i++) {
    double acc = 0.0;
    for (int64_t p = rowptr[i]; p < rowptr[i + 1]; p++) {
      acc += val[p] * x[col[p]];
    }
    res[i] = acc;
  }
}

/* Yardstick build only (-DORC_EXACT_SUMS, libkrylov_oracle_exact.so): the two
 * reductions below are summed pairwise in extended precision (64-bit
 * significand, blocks of 128 products at the leaves), which bounds their
 * rounding error by about 150 * 2^-64 of the sum of the magnitudes -- three
 * orders of magnitude below any double-precision summation order.  It answers
 * one question, "which of two double-precision runs is closer to exactly
 * summed dot products", and is never the parity checker: the reference sums
 * sequentially in double (include/matrix.hpp:35-52), as the default build does. */
#ifdef ORC_EXACT_SUMS
static long double orc_pairwise(const double *a, const double *b, long n) {
  if (n <= 128) {
    long double s = 0.0L;
    for (long i = 0; i < n; i++) s += (long double)a[i] * (long double)b[i];
    return s;
  }
  const long h = n / 2;
  return orc_pairwise(a, b, h) + orc_pairwise(a + h, b + h, n - h);
}
#endif

/* dot(real&, const vector&, const vector&), include/matrix.hpp:35-44 */
ORC_API double orc_dot(const double *lhs, const double *rhs, long n) {
#ifdef ORC_EXACT_SUMS
  return (double)orc_pairwise(lhs, rhs, n);
#else
  double acc = 0.0;
  ORC_PAR_SUM(acc)
  for (long i = 0; i < n; i++) {
    acc += lhs[i] * rhs[i];
  }
  return acc;
#endif
}

/* l2_norm, include/matrix.hpp:46-52 */
ORC_API double orc_l2_norm(const double *v, long n) {
#ifdef ORC_EXACT_SUMS
  return (double)sqrtl(orc_pairwise(v, v, n));
#else
  double acc = 0.0;
  ORC_PAR_SUM(acc)
  for (long i = 0; i < n; i++) {
    acc += v[i] * v[i];
  }
  return sqrt(acc);
#endif
}

/* normalize, include/matrix.hpp:54-60 (true division, not a reciprocal) */
ORC_API double orc_normalize(double *v, long n) {
  const double l2 = orc_l2_norm(v, n);
  ORC_PAR_FOR
  for (long i = 0; i < n; i++) {
    v[i] /= l2;
  }
  return l2;
}

/* ------------------------------------------------------------------------ */
/* include/solver.hpp : LUSolver                                             */
/* ------------------------------------------------------------------------ */

/* LUSolver::lu_decomp, include/solver.hpp:58-96.  l and u are n x n. */
ORC_API void orc_lu_decomp(const double *sys, double *l, double *u, long n) {
  for (long p = 0; p < n * n; p++) {
    l[p] = 0.0;
    u[p] = 0.0;
  }
  for (long k = 0; k < n; k++) {
    l[k * n + k] = 1.0;
    u[k * n + k] = sys[k * n + k];
    for (long j = 0; j < k; j++) {
      u[k * n + k] -= l[k * n + j] * u[j * n + k];
    }
    u[k * n + k] /= l[k * n + k];
    for (long j = k + 1; j < n; j++) {
      u[k * n + j] = sys[k * n + j];
      for (long i = 0; i < k; i++) {
        u[k * n + j] -= l[k * n + i] * u[i * n + j];
      }
      u[k * n + j] /= l[k * n + k];
    }
    for (long j = k + 1; j < n; j++) {
      l[j * n + k] = sys[j * n + k];
      for (long i = 0; i < k; i++) {
        l[j * n + k] -= l[j * n + i] * u[i * n + k];
      }
      l[j * n + k] /= u[k * n + k];
    }
  }
}

/* LUSolver::solve, include/solver.hpp:34-56.  work holds 2*n*n + n doubles. */
ORC_API void orc_lu_solve(const double *sys, const double *rhs, double *x,
                          long n, double *work) {
  double *l = work, *u = work + n * n, *partial = work + 2 * n * n;
  orc_lu_decomp(sys, l, u, n);
  for (long i = 0; i < n; i++) {
    partial[i] = rhs[i];
    for (long j = 0; j < i; j++) {
      partial[i] -= l[i * n + j] * partial[j];
    }
    partial[i] /= l[i * n + i];
  }
  for (long i = n - 1; i >= 0; i--) {
    x[i] = partial[i];
    for (long j = i + 1; j < n; j++) {
      x[i] -= u[i * n + j] * x[j];
    }
    x[i] /= u[i * n + i];
  }
}

/* ------------------------------------------------------------------------ */
/* include/solver.hpp : SparseTriDiagSolver / TriDiagSolver                  */
/* ------------------------------------------------------------------------ */

/* SparseTriDiagSolver::solve, include/solver.hpp:246-273.
 * diags is n x 3 (lower, diagonal, upper); u_diag holds n-1 doubles. */
ORC_API void orc_sparse_tridiag_solve(const double *diags, const double *sol,
                                      double *vars, double *u_diag, long n) {
  vars[0] = sol[0] / diags[1];
  if (n > 1) u_diag[0] = diags[2] / diags[1];
  for (long i = 1; i < n; i++) {
    const double diag = diags[i * 3 + 1] - u_diag[i - 1] * diags[i * 3 + 0];
    vars[i] = (sol[i] - vars[i - 1] * diags[i * 3 + 0]) / diag;
    if (i + 1 < n) {
      u_diag[i] = diags[i * 3 + 2] / diag;
    }
  }
  for (long i = n - 2; i >= 0; i--) {
    vars[i] -= vars[i + 1] * u_diag[i];
  }
}

/* TriDiagSolver::solve, include/solver.hpp:340-364 (dense n x n input). */
ORC_API void orc_tridiag_solve(const double *sys, const double *sol,
                               double *vars, double *u_diag, long n) {
  vars[0] = sol[0] / sys[0];
  if (n > 1) u_diag[0] = sys[1] / sys[0];
  for (long i = 1; i < n; i++) {
    const double diag = sys[i * n + i] - u_diag[i - 1] * sys[i * n + i - 1];
    vars[i] = (sol[i] - vars[i - 1] * sys[i * n + i - 1]) / diag;
    if (i + 1 < n) {
      u_diag[i] = sys[i * n + i + 1] / diag;
    }
  }
  for (long i = n - 2; i >= 0; i--) {
    vars[i] -= vars[i + 1] * u_diag[i];
  }
}

/* ------------------------------------------------------------------------ */
/* Linear operator handle used by the GMRES restatement                      */
/* ------------------------------------------------------------------------ */

typedef struct orc_op {
  long n;
  int kind; /* 0 dense, 1 csr */
  const double *dense;
  const int64_t *rowptr;
  const int32_t *col;
  const double *val;
} orc_op;

ORC_API orc_op *orc_op_dense(const double *a, long n) {
  orc_op *op = (orc_op *)calloc(1, sizeof(orc_op));
  op->n = n;
  op->kind = 0;
  op->dense = a;
  return op;
}

ORC_API orc_op *orc_op_csr(const int64_t *rowptr, const int32_t *col,
                           const double *val, long n) {
  orc_op *op = (orc_op *)calloc(1, sizeof(orc_op));
  op->n = n;
  op->kind = 1;
  op->rowptr = rowptr;
  op->col = col;
  op->val = val;
  return op;
}

ORC_API void orc_op_free(orc_op *op) { free(op); }

ORC_API void orc_op_apply(const orc_op *op, double *y, const double *x) {
  if (op->kind == 0) {
    orc_dense_matvec(y, op->dense, x, op->n, op->n);
  } else {
    orc_csr_matvec(y, op->rowptr, op->col, op->val, x, op->n);
  }
}

/* A(r,c) of the operator (0.0 where nothing is stored). */
static double orc_op_entry(const orc_op *op, long r, long c) {
  if (op->kind == 0) return op->dense[r * op->n + c];
  for (int64_t p = op->rowptr[r]; p < op->rowptr[r + 1]; p++) {
    if (op->col[p] == c) return op->val[p];
  }
  return 0.0;
}

/* ------------------------------------------------------------------------ */
/* include/solver.hpp : GMRESSolver<m> / GMRESTDPSolver<m>                   */
/* ------------------------------------------------------------------------ */

typedef struct orc_gmres {
  long n;
  int m;
  int tdp;        /* 0: GMRESSolver, 1: GMRESTDPSolver */
  int fixed_tdp;  /* opt-in: apply M^-1 to the combination (not in the reference) */
  int nvec;       /* subspace.size() */
  double *V;      /* (m+1) x n basis, row k = subspace[k] */
  double *H;      /* (m+1) x m row-major */
  double *vars;   /* Solver::vars, include/solver.hpp:14-19 */
  double *next;   /* work vector of add_arnoldi */
  double *lu_work;
  double *sys, *rhs, *y;
  double *tdiags, *td_vars, *td_udiag;
  /* restarted solver with residual monitor (not in the reference) */
  int ext_on, ext_max_cycles;
  double ext_rtol;
  int last_iters, last_cycles, last_converged;
  double last_relres;
} orc_gmres;

ORC_API orc_gmres *orc_gmres_create(long n, int m, int tdp) {
  orc_gmres *g = (orc_gmres *)calloc(1, sizeof(orc_gmres));
  g->n = n;
  g->m = m;
  g->tdp = tdp;
  g->V = (double *)malloc(sizeof(double) * (size_t)(m + 1) * (size_t)n);
  g->H = (double *)calloc((size_t)(m + 1) * m, sizeof(double));
  g->vars = (double *)malloc(sizeof(double) * (size_t)n);
  g->next = (double *)malloc(sizeof(double) * (size_t)n);
  g->lu_work = (double *)malloc(sizeof(double) * (size_t)(2 * m * m + m));
  g->sys = (double *)malloc(sizeof(double) * (size_t)m * m);
  g->rhs = (double *)malloc(sizeof(double) * (size_t)m);
  g->y = (double *)malloc(sizeof(double) * (size_t)m);
  for (long i = 0; i < n; i++) g->vars[i] = orc_snan(); /* solver.hpp:16-18 */
  if (tdp) {
    g->tdiags = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    g->td_vars = (double *)malloc(sizeof(double) * (size_t)n);
    g->td_udiag = (double *)malloc(sizeof(double) * (size_t)n);
  }
  return g;
}

ORC_API void orc_gmres_free(orc_gmres *g) {
  if (!g) return;
  free(g->V); free(g->H); free(g->vars); free(g->next); free(g->lu_work);
  free(g->sys); free(g->rhs); free(g->y);
  free(g->tdiags); free(g->td_vars); free(g->td_udiag);
  free(g);
}

ORC_API void orc_gmres_set_fixed_preconditioning(orc_gmres *g, int on) {
  g->fixed_tdp = on;
}

/* GMRESSolver::initial_subspace, include/solver.hpp:168-174 */
ORC_API double orc_gmres_initial_subspace(orc_gmres *g, const double *sol) {
  memcpy(g->V, sol, sizeof(double) * (size_t)g->n);
  const double beta = orc_normalize(g->V, g->n);
  g->nvec = 1;
  return beta;
}

/* GMRESTDPSolver::assemble_preconditioner, include/solver.hpp:295-314 */
ORC_API void orc_gmres_assemble_preconditioner(orc_gmres *g, const orc_op *op) {
  const long n = g->n;
  double *t = g->tdiags;
  t[0] = orc_snan();
  t[1] = orc_op_entry(op, 0, 0);
  t[2] = orc_op_entry(op, 0, 1);
  for (long i = 1; i < n - 1; i++) {
    t[i * 3 + 0] = orc_op_entry(op, i, i - 1);
    t[i * 3 + 1] = orc_op_entry(op, i, i);
    t[i * 3 + 2] = orc_op_entry(op, i, i + 1);
  }
  t[(n - 1) * 3 + 0] = orc_op_entry(op, n - 1, n - 2);
  t[(n - 1) * 3 + 1] = orc_op_entry(op, n - 1, n - 1);
  t[(n - 1) * 3 + 2] = orc_snan();
}

/* GMRESSolver::gram_schmidt, include/solver.hpp:196-211 (modified G-S, the
 * dot is taken with the already-updated vector; `next -= h * v` is a
 * multiply followed by a subtract). */
static void orc_gram_schmidt(orc_gmres *g, double *next) {
  const long n = g->n;
  const int m = g->m;
  const int j = g->nvec - 1;
  for (int i = 0; i <= j; i++) {
    const double *v = g->V + (size_t)i * n;
    const double h = orc_dot(next, v, n);
    g->H[i * m + j] = h;
    ORC_PAR_FOR
    for (long k = 0; k < n; k++) {
      next[k] -= h * v[k];
    }
  }
  g->H[(j + 1) * m + j] = orc_normalize(next, n);
}

/* GMRESSolver::add_arnoldi (include/solver.hpp:185-194) and the
 * preconditioned override GMRESTDPSolver::add_arnoldi (:316-328). */
ORC_API double orc_gmres_add_arnoldi(orc_gmres *g, const orc_op *op) {
  const long n = g->n;
  const int j = g->nvec - 1;
  const double *prev = g->V + (size_t)j * n;
  if (g->tdp) {
    orc_sparse_tridiag_solve(g->tdiags, prev, g->td_vars, g->td_udiag, n);
    orc_op_apply(op, g->next, g->td_vars);
  } else {
    orc_op_apply(op, g->next, prev);
  }
  orc_gram_schmidt(g, g->next);
  memcpy(g->V + (size_t)(j + 1) * n, g->next, sizeof(double) * (size_t)n);
  g->nvec++;
  return g->H[(j + 1) * g->m + j];
}

/* GMRESSolver::solve (include/solver.hpp:124-161); GMRESTDPSolver::solve
 * (:290-293) first rebuilds the tridiagonal preconditioner.  The returned
 * vector is the plain combination of the basis vectors: for the
 * preconditioned solver the reference does not apply M^-1 to it. */
ORC_API const double *orc_gmres_solve_extended(orc_gmres *g, const orc_op *op,
                                               const double *sol);

ORC_API const double *orc_gmres_solve(orc_gmres *g, const orc_op *op,
                                      const double *sol) {
  const long n = g->n;
  const int m = g->m;
  if (g->ext_on) return orc_gmres_solve_extended(g, op, sol);
  if (g->tdp) orc_gmres_assemble_preconditioner(g, op);
  const double beta = orc_gmres_initial_subspace(g, sol);
  for (int p = 0; p < (m + 1) * m; p++) g->H[p] = 0.0;
  for (int i = 0; i < m; i++) {
    orc_gmres_add_arnoldi(g, op);
  }
  /* sys = transpose(H) * H via the generic triple loop (solver.hpp:144-145) */
  for (int i = 0; i < m; i++) {
    for (int jj = 0; jj < m; jj++) {
      double acc = 0.0;
      for (int k = 0; k < m + 1; k++) {
        acc += g->H[k * m + i] * g->H[k * m + jj];
      }
      g->sys[i * m + jj] = acc;
    }
  }
  for (int i = 0; i < m; i++) {
    g->rhs[i] = beta * g->H[0 * m + i];
  }
  orc_lu_solve(g->sys, g->rhs, g->y, m, g->lu_work);
  for (long k = 0; k < n; k++) g->vars[k] = 0.0;
  for (int i = 0; i < m; i++) {
    const double *dim = g->V + (size_t)i * n;
    const double s = g->y[i];
    ORC_PAR_FOR
    for (long k = 0; k < n; k++) {
      g->vars[k] += s * dim[k];
    }
  }
  if (g->tdp && g->fixed_tdp) {
    /* Opt-in repair that is NOT in the reference: x = M^-1 u. */
    orc_sparse_tridiag_solve(g->tdiags, g->vars, g->td_vars, g->td_udiag, n);
    memcpy(g->vars, g->td_vars, sizeof(double) * (size_t)n);
  }
  return g->vars;
}

/* ------------------------------------------------------------------------ */
/* Restarted GMRES(m) with Givens-rotation least squares and a residual      */
/* monitor.  NOT part of the reference (its solve is the fixed cycle above); */
/* this is the CPU statement of the device library's extended mode, used to  */
/* check that mode's iteration counts and solutions.  The Arnoldi step is    */
/* the reference's (add_arnoldi / gram_schmidt above); with the tridiagonal  */
/* preconditioner the result is x = M^-1 (V y), proper right                 */
/* preconditioning.                                                          */
/* ------------------------------------------------------------------------ */
ORC_API void orc_gmres_set_extended(orc_gmres *g, int on, double rtol, int max_cycles) {
  g->ext_on = on;
  g->ext_rtol = rtol;
  g->ext_max_cycles = max_cycles;
}

ORC_API void orc_gmres_last_stats(const orc_gmres *g, int *iters, int *cycles,
                                  int *converged, double *relres) {
  *iters = g->last_iters;
  *cycles = g->last_cycles;
  *converged = g->last_converged;
  *relres = g->last_relres;
}

ORC_API const double *orc_gmres_solve_extended(orc_gmres *g, const orc_op *op,
                                               const double *sol) {
  const long n = g->n;
  const int m = g->m;
  double *R = (double *)calloc((size_t)(m + 1) * m, sizeof(double));
  double *cs = (double *)calloc((size_t)m, sizeof(double));
  double *sn = (double *)calloc((size_t)m, sizeof(double));
  double *gv = (double *)calloc((size_t)m + 1, sizeof(double));
  double *col = (double *)calloc((size_t)m + 1, sizeof(double));
  double *r = (double *)malloc(sizeof(double) * (size_t)n);
  double *u = (double *)malloc(sizeof(double) * (size_t)n);
  double beta0 = 0.0, tol = 0.0, res = 0.0;
  if (g->tdp) orc_gmres_assemble_preconditioner(g, op);
  for (long k = 0; k < n; k++) g->vars[k] = 0.0;
  g->last_iters = 0;
  g->last_cycles = 0;
  g->last_converged = 0;
  for (int cycle = 0; cycle < g->ext_max_cycles; cycle++) {
    int stop = 0, iters = 0;
    if (cycle == 0) {
      memcpy(r, sol, sizeof(double) * (size_t)n);
    } else {
      orc_op_apply(op, g->next, g->vars);
      for (long k = 0; k < n; k++) r[k] = sol[k] - g->next[k];
    }
    const double beta = orc_gmres_initial_subspace(g, r); /* v_0 = r / ||r|| */
    if (cycle == 0) {
      beta0 = beta;
      tol = g->ext_rtol * beta;
    }
    res = beta;
    for (int i = 0; i <= m; i++) gv[i] = 0.0;
    gv[0] = beta;
    if (beta <= tol) {
      g->last_converged = 1;
      stop = 1;
    } else if (!(beta == beta)) {
      stop = 1;
    }
    for (int p = 0; p < (m + 1) * m; p++) g->H[p] = 0.0;
    for (int j = 0; j < m && !stop; j++) {
      orc_gmres_add_arnoldi(g, op);
      for (int i = 0; i <= j + 1; i++) col[i] = g->H[i * m + j];
      for (int i = 0; i < j; i++) {
        const double t = cs[i] * col[i] + sn[i] * col[i + 1];
        col[i + 1] = cs[i] * col[i + 1] - sn[i] * col[i];
        col[i] = t;
      }
      const double a = col[j], b = col[j + 1];
      const double d = sqrt(a * a + b * b);
      double cc = 1.0, ss = 0.0;
      if (d != 0.0) {
        cc = a / d;
        ss = b / d;
      }
      col[j] = cc * a + ss * b;
      col[j + 1] = 0.0;
      const double gj = gv[j];
      const double gn = -(ss * gj);
      gv[j] = cc * gj;
      gv[j + 1] = gn;
      cs[j] = cc;
      sn[j] = ss;
      for (int i = 0; i <= j + 1; i++) R[i * m + j] = col[i];
      res = fabs(gn);
      iters = j + 1;
      if (res <= tol) {
        g->last_converged = 1;
        stop = 1;
      } else if (!(res == res)) {
        stop = 1;
      }
    }
    /* back substitution over the completed columns */
    for (int i = 0; i < m; i++) g->y[i] = 0.0;
    for (int i = iters - 1; i >= 0; i--) {
      double acc = 0.0;
      for (int jj = i + 1; jj < iters; jj++) acc += R[i * m + jj] * g->y[jj];
      g->y[i] = (gv[i] - acc) / R[i * m + i];
    }
    for (long k = 0; k < n; k++) u[k] = 0.0;
    for (int i = 0; i < iters; i++) {
      const double *dim = g->V + (size_t)i * n;
      const double sc = g->y[i];
      for (long k = 0; k < n; k++) u[k] += sc * dim[k];
    }
    if (g->tdp) {
      orc_sparse_tridiag_solve(g->tdiags, u, g->td_vars, g->td_udiag, n);
      for (long k = 0; k < n; k++) g->vars[k] += g->td_vars[k];
    } else {
      for (long k = 0; k < n; k++) g->vars[k] += u[k];
    }
    g->last_cycles = cycle + 1;
    g->last_iters += iters;
    if (stop) break;
  }
  g->last_relres = beta0 > 0.0 ? res / beta0 : res;
  free(R); free(cs); free(sn); free(gv); free(col); free(r); free(u);
  return g->vars;
}

ORC_API const double *orc_gmres_subspace_vector(const orc_gmres *g, int i) {
  return g->V + (size_t)i * g->n;
}
ORC_API double orc_gmres_hessenberg(const orc_gmres *g, int i, int j) {
  return g->H[i * g->m + j];
}
ORC_API const double *orc_gmres_hessenberg_data(const orc_gmres *g) { return g->H; }
ORC_API const double *orc_gmres_prev_result(const orc_gmres *g) { return g->vars; }
ORC_API const double *orc_gmres_tdiags(const orc_gmres *g) { return g->tdiags; }
ORC_API const double *orc_gmres_reduced_solution(const orc_gmres *g) { return g->y; }

/* ------------------------------------------------------------------------ */
/* include/mesh.hpp, include/space_disc.hpp, include/boundary_conds.hpp,     */
/* include/time_disc.hpp                                                     */
/* ------------------------------------------------------------------------ */

enum { ORC_FIELD_T = 0, ORC_FIELD_U = 1, ORC_FIELD_V = 2 };
enum { ORC_SIDE_BOTTOM = 0, ORC_SIDE_TOP = 1, ORC_SIDE_LEFT = 2, ORC_SIDE_RIGHT = 3 };

typedef struct orc_ie {
  int cx, cy;
  double min_x, max_x, min_y, max_y, dx, dy;
  double diffusion, reynolds, prandtl, eckert, inv_rp;
  double time;
  double *fld[3];          /* padded (cx+2) x (cy+2) */
  double weight[3][4];     /* BoundaryCondition::weight_ per field, side */
  double *bcval[3][4];     /* bc(x,y,t) sampled at the boundary-cell coords */
  long n;
  int64_t *rowptr;
  int32_t *col;
  double *val;
  double *sol_vec;
  orc_op *op;
  orc_gmres *solver;
} orc_ie;

#define P_(s) ((s)->cy + 2)
#define AT(s, f, i, j) ((s)->fld[f][((i) + 1) * P_(s) + ((j) + 1)])

/* Mesh coordinate helpers, include/mesh.hpp:20-42 */
ORC_API double orc_left_x(const orc_ie *s, int i) { return s->min_x + i * s->dx; }
ORC_API double orc_right_x(const orc_ie *s, int i) { return s->min_x + (i + 1) * s->dx; }
ORC_API double orc_bottom_y(const orc_ie *s, int j) { return s->min_y + j * s->dy; }
ORC_API double orc_top_y(const orc_ie *s, int j) { return s->min_y + (j + 1) * s->dy; }
ORC_API double orc_median_x(const orc_ie *s, int i) { return s->min_x + s->dx / 2 + i * s->dx; }
ORC_API double orc_median_y(const orc_ie *s, int j) { return s->min_y + s->dy / 2 + j * s->dy; }

/* ImplicitEuler::cell_idx, include/time_disc.hpp:341-376 */
ORC_API long orc_cell_idx(const orc_ie *s, int i, int j) {
  const int gc = 1;
  if (i == -gc) {
    return j;
  } else if (i == s->cx) {
    return s->cy + i * (s->cy + 2 * gc) + j;
  } else {
    return s->cy + i * (s->cy + 2 * gc) + gc + j;
  }
}

/* SecondOrderCentered stencil coefficients, include/space_disc.hpp:83-113 */
static double Dx_p1(const orc_ie *s, int i, int j) {
  return (AT(s, 1, i + 1, j) / 2.0 - s->diffusion * s->inv_rp / s->dx) / s->dx;
}
static double Dx_0(const orc_ie *s) {
  return 2.0 * s->diffusion * s->inv_rp / (s->dx * s->dx);
}
static double Dx_m1(const orc_ie *s, int i, int j) {
  return (-AT(s, 1, i - 1, j) / 2.0 - s->diffusion * s->inv_rp / s->dx) / s->dx;
}
static double Dy_p1(const orc_ie *s, int i, int j) {
  return (AT(s, 2, i, j + 1) / 2.0 - s->diffusion * s->inv_rp / s->dy) / s->dy;
}
static double Dy_0(const orc_ie *s) {
  return 2.0 * s->diffusion * s->inv_rp / (s->dy * s->dy);
}
static double Dy_m1(const orc_ie *s, int i, int j) {
  return (-AT(s, 2, i, j - 1) / 2.0 - s->diffusion * s->inv_rp / s->dy) / s->dy;
}

/* Face fluxes, include/space_disc.hpp:116-141 */
static double uT_x_flux(const orc_ie *s, int i, int j) {
  return (AT(s, 0, i, j) * AT(s, 1, i, j) + AT(s, 0, i + 1, j) * AT(s, 1, i + 1, j)) / 2.0;
}
static double vT_y_flux(const orc_ie *s, int i, int j) {
  return (AT(s, 0, i, j) * AT(s, 2, i, j) + AT(s, 0, i, j + 1) * AT(s, 2, i, j + 1)) / 2.0;
}
static double dx_flux(const orc_ie *s, int i, int j) {
  return (AT(s, 0, i + 1, j) - AT(s, 0, i, j)) / s->dx;
}
static double dy_flux(const orc_ie *s, int i, int j) {
  return (AT(s, 0, i, j + 1) - AT(s, 0, i, j)) / s->dy;
}

/* EnergyAssembly::lapl_T_flux_integral, include/space_disc.hpp:33-44 */
static double lapl_T_flux_integral(const orc_ie *s, int i, int j) {
  const double x2_deriv = (dx_flux(s, i, j) - dx_flux(s, i - 1, j)) / s->dx;
  const double y2_deriv = (dy_flux(s, i, j) - dy_flux(s, i, j - 1)) / s->dy;
  return (x2_deriv + y2_deriv) * s->inv_rp;
}

/* EnergyAssembly::flux_integral, include/space_disc.hpp:19-31 */
ORC_API double orc_flux_integral(const orc_ie *s, int i, int j) {
  const double x_deriv = (uT_x_flux(s, i, j) - uT_x_flux(s, i - 1, j)) / s->dx;
  const double y_deriv = (vT_y_flux(s, i, j) - vT_y_flux(s, i, j - 1)) / s->dy;
  return (-x_deriv - y_deriv) + s->diffusion * lapl_T_flux_integral(s, i, j);
}

/* The four centred differences, include/space_disc.hpp:157-179 */
static double du_dx_fd(const orc_ie *s, int i, int j) {
  return (AT(s, 1, i + 1, j) - AT(s, 1, i - 1, j)) / (2.0 * s->dx);
}
static double du_dy_fd(const orc_ie *s, int i, int j) {
  return (AT(s, 1, i, j + 1) - AT(s, 1, i, j - 1)) / (2.0 * s->dy);
}
static double dv_dx_fd(const orc_ie *s, int i, int j) {
  return (AT(s, 2, i + 1, j) - AT(s, 2, i - 1, j)) / (2.0 * s->dx);
}
static double dv_dy_fd(const orc_ie *s, int i, int j) {
  return (AT(s, 2, i, j + 1) - AT(s, 2, i, j - 1)) / (2.0 * s->dy);
}

/* SecondOrderCentered::source_fd, include/space_disc.hpp:145-155 */
ORC_API double orc_source_fd(const orc_ie *s, int i, int j) {
  const double u_dx = du_dx_fd(s, i, j);
  const double v_dy = dv_dy_fd(s, i, j);
  const double u_dy = du_dy_fd(s, i, j);
  const double v_dx = dv_dx_fd(s, i, j);
  const double cross_term = u_dy + v_dx;
  return s->eckert / s->reynolds *
         (2.0 * (u_dx * u_dx + v_dy * v_dy) + cross_term * cross_term);
}

/* One member function of SecondOrderCentered / EnergyAssembly evaluated on every
 * interior cell (i, j), i slow: the loops of the reference's own tests
 * (src/test.cpp:753-763, :798-820, :845-888, ...).  out holds cx * cy values. */
enum {
  ORC_Q_DX_M1 = 0, ORC_Q_DX_0, ORC_Q_DX_P1, ORC_Q_DY_M1, ORC_Q_DY_0, ORC_Q_DY_P1,
  ORC_Q_DX_FLUX, ORC_Q_DY_FLUX, ORC_Q_LAPL_T_FLUX_INTEGRAL, ORC_Q_FLUX_INTEGRAL,
  ORC_Q_DU_DX_FD, ORC_Q_DV_DY_FD, ORC_Q_DU_DY_FD, ORC_Q_DV_DX_FD, ORC_Q_SOURCE_FD
};
ORC_API int orc_eval_cells(const orc_ie *s, int what, double *out) {
  for (int i = 0; i < s->cx; i++) {
    for (int j = 0; j < s->cy; j++) {
      double v;
      switch (what) {
        case ORC_Q_DX_M1: v = Dx_m1(s, i, j); break;
        case ORC_Q_DX_0: v = Dx_0(s); break;
        case ORC_Q_DX_P1: v = Dx_p1(s, i, j); break;
        case ORC_Q_DY_M1: v = Dy_m1(s, i, j); break;
        case ORC_Q_DY_0: v = Dy_0(s); break;
        case ORC_Q_DY_P1: v = Dy_p1(s, i, j); break;
        case ORC_Q_DX_FLUX: v = dx_flux(s, i, j); break;
        case ORC_Q_DY_FLUX: v = dy_flux(s, i, j); break;
        case ORC_Q_LAPL_T_FLUX_INTEGRAL: v = lapl_T_flux_integral(s, i, j); break;
        case ORC_Q_FLUX_INTEGRAL: v = orc_flux_integral(s, i, j); break;
        case ORC_Q_DU_DX_FD: v = du_dx_fd(s, i, j); break;
        case ORC_Q_DV_DY_FD: v = dv_dy_fd(s, i, j); break;
        case ORC_Q_DU_DY_FD: v = du_dy_fd(s, i, j); break;
        case ORC_Q_DV_DX_FD: v = dv_dx_fd(s, i, j); break;
        case ORC_Q_SOURCE_FD: v = orc_source_fd(s, i, j); break;
        default: return 1;
      }
      out[(size_t)i * s->cy + j] = v;
    }
  }
  return 0;
}

/* BoundaryCondition::ghost_cell, include/boundary_conds.hpp:62-66 */
ORC_API double orc_ghost_cell(double bndry_val, double bc, double size, double weight) {
  return (2.0 * bc - bndry_val) * weight + (size * bc + bndry_val) * (1.0 - weight);
}

/* TimeDisc::apply_bcs (include/time_disc.hpp:146-163) with the view wiring of
 * the TimeDisc constructor (:51-134):
 *  - bottom/top conditions use size = dy, left/right use size = dx;
 *  - every LEFT condition reads the RIGHT boundary column (:66-67, :93-94,
 *    :121-122), which is reproduced here on purpose.
 * All conditions read interior cells and write ghost cells only, so the call
 * order of the reference has no effect on the result. */
ORC_API void orc_ie_apply_bcs(orc_ie *s) {
  const int cx = s->cx, cy = s->cy;
  for (int f = 0; f < 3; f++) {
    for (int j = 0; j < cy; j++) {
      AT(s, f, -1, j) = orc_ghost_cell(AT(s, f, cx - 1, j), s->bcval[f][ORC_SIDE_LEFT][j],
                                       s->dx, s->weight[f][ORC_SIDE_LEFT]);
      AT(s, f, cx, j) = orc_ghost_cell(AT(s, f, cx - 1, j), s->bcval[f][ORC_SIDE_RIGHT][j],
                                       s->dx, s->weight[f][ORC_SIDE_RIGHT]);
    }
    for (int i = 0; i < cx; i++) {
      AT(s, f, i, cy) = orc_ghost_cell(AT(s, f, i, cy - 1), s->bcval[f][ORC_SIDE_TOP][i],
                                       s->dy, s->weight[f][ORC_SIDE_TOP]);
      AT(s, f, i, -1) = orc_ghost_cell(AT(s, f, i, 0), s->bcval[f][ORC_SIDE_BOTTOM][i],
                                       s->dy, s->weight[f][ORC_SIDE_BOTTOM]);
    }
  }
}

/* position of A(r,c) in the CSR arrays (it must exist) */
static int64_t orc_pos(const orc_ie *s, long r, long c) {
  for (int64_t p = s->rowptr[r]; p < s->rowptr[r + 1]; p++) {
    if (s->col[p] == c) return p;
  }
  return -1;
}

/* Sparsity pattern of the ImplicitEuler system in cell_idx numbering: the
 * entries written by the constructor (include/time_disc.hpp:249-274) and by
 * assemble_system (:294-337), columns ascending inside each row. */
static void orc_ie_build_pattern(orc_ie *s) {
  const int cx = s->cx, cy = s->cy;
  const long n = s->n;
  s->rowptr = (int64_t *)calloc((size_t)n + 1, sizeof(int64_t));
  int64_t nnz = 5LL * cx * cy + 2LL * (2 * cx + 2 * cy);
  s->col = (int32_t *)malloc(sizeof(int32_t) * (size_t)nnz);
  s->val = (double *)calloc((size_t)nnz, sizeof(double));
  int64_t p = 0;
  long r = 0;
  for (int j = 0; j < cy; j++) { /* left ghost column */
    s->rowptr[r++] = p;
    s->col[p++] = (int32_t)orc_cell_idx(s, -1, j);
    s->col[p++] = (int32_t)orc_cell_idx(s, 0, j);
  }
  for (int i = 0; i < cx; i++) {
    s->rowptr[r++] = p; /* bottom ghost */
    s->col[p++] = (int32_t)orc_cell_idx(s, i, -1);
    s->col[p++] = (int32_t)orc_cell_idx(s, i, 0);
    for (int j = 0; j < cy; j++) {
      s->rowptr[r++] = p;
      s->col[p++] = (int32_t)orc_cell_idx(s, i - 1, j);
      s->col[p++] = (int32_t)orc_cell_idx(s, i, j - 1);
      s->col[p++] = (int32_t)orc_cell_idx(s, i, j);
      s->col[p++] = (int32_t)orc_cell_idx(s, i, j + 1);
      s->col[p++] = (int32_t)orc_cell_idx(s, i + 1, j);
    }
    s->rowptr[r++] = p; /* top ghost */
    s->col[p++] = (int32_t)orc_cell_idx(s, i, cy - 1);
    s->col[p++] = (int32_t)orc_cell_idx(s, i, cy);
  }
  for (int j = 0; j < cy; j++) { /* right ghost column */
    s->rowptr[r++] = p;
    s->col[p++] = (int32_t)orc_cell_idx(s, cx - 1, j);
    s->col[p++] = (int32_t)orc_cell_idx(s, cx, j);
  }
  s->rowptr[r] = p;
}

/* Mesh constructor geometry, src/mesh.cpp:8-21.  Fields are given already
 * sampled at the cell medians (src/mesh.cpp:60-67); ghosts and corners start
 * as signalling NaN (src/mesh.cpp:34-51). */
ORC_API orc_ie *orc_ie_create(double c1x, double c1y, double c2x, double c2y,
                              int cx, int cy, const double *T0, const double *u0,
                              const double *v0, double diffusion, double reynolds,
                              double prandtl, double eckert, const int *bc_is_dirichlet,
                              int m, int tdp) {
  orc_ie *s = (orc_ie *)calloc(1, sizeof(orc_ie));
  s->cx = cx;
  s->cy = cy;
  s->min_x = c1x < c2x ? c1x : c2x;
  s->max_x = c1x < c2x ? c2x : c1x;
  s->min_y = c1y < c2y ? c1y : c2y;
  s->max_y = c1y < c2y ? c2y : c1y;
  s->dx = (s->max_x - s->min_x) / (double)cx;
  s->dy = (s->max_y - s->min_y) / (double)cy;
  s->diffusion = diffusion;
  s->reynolds = reynolds;
  s->prandtl = prandtl;
  s->eckert = eckert;
  s->inv_rp = 1.0 / (reynolds * prandtl); /* space_disc.hpp:61 */
  s->time = 0.0;
  const size_t np = (size_t)(cx + 2) * (cy + 2);
  const double *src[3] = {T0, u0, v0};
  for (int f = 0; f < 3; f++) {
    s->fld[f] = (double *)malloc(sizeof(double) * np);
    for (size_t k = 0; k < np; k++) s->fld[f][k] = orc_snan();
    for (int i = 0; i < cx; i++)
      for (int j = 0; j < cy; j++) AT(s, f, i, j) = src[f][(size_t)i * cy + j];
    for (int side = 0; side < 4; side++) {
      /* weight_ = 1 for Dirichlet, 0 for Neumann, boundary_conds.hpp:50 */
      s->weight[f][side] = bc_is_dirichlet[f * 4 + side] ? 1.0 : 0.0;
      const int len = (side == ORC_SIDE_BOTTOM || side == ORC_SIDE_TOP) ? cx : cy;
      s->bcval[f][side] = (double *)calloc((size_t)len, sizeof(double));
    }
  }
  s->n = (long)(cx + 2) * (cy + 2) - 4; /* num_eqns, time_disc.hpp:179-188 */
  orc_ie_build_pattern(s);
  s->sol_vec = (double *)calloc((size_t)s->n, sizeof(double));
  s->op = orc_op_csr(s->rowptr, s->col, s->val, s->n);
  s->solver = m > 0 ? orc_gmres_create(s->n, m, tdp) : 0;
  return s;
}

ORC_API void orc_ie_free(orc_ie *s) {
  if (!s) return;
  for (int f = 0; f < 3; f++) {
    free(s->fld[f]);
    for (int side = 0; side < 4; side++) free(s->bcval[f][side]);
  }
  free(s->rowptr); free(s->col); free(s->val); free(s->sol_vec);
  orc_op_free(s->op);
  orc_gmres_free(s->solver);
  free(s);
}

/* Values of bc(x, y, t) at the boundary-cell coordinates of one side, as the
 * host evaluates them before every apply_bcs (the coordinates are those of
 * include/time_disc.hpp:55-76). */
ORC_API void orc_ie_set_bc_values(orc_ie *s, int field, int side, const double *vals) {
  const int len = (side == ORC_SIDE_BOTTOM || side == ORC_SIDE_TOP) ? s->cx : s->cy;
  memcpy(s->bcval[field][side], vals, sizeof(double) * (size_t)len);
}

/* Second half of the ImplicitEuler constructor, include/time_disc.hpp:246-274:
 * ghost fill at t = 0, then the four off-diagonal bands, which are never
 * refreshed afterwards. */
ORC_API void orc_ie_init_offdiag(orc_ie *s) {
  orc_ie_apply_bcs(s);
  for (int i = 0; i < s->cx; i++) {
    for (int j = 0; j < s->cy; j++) {
      const long cur = orc_cell_idx(s, i, j);
      s->val[orc_pos(s, cur, orc_cell_idx(s, i, j - 1))] = Dy_m1(s, i, j);
      s->val[orc_pos(s, cur, orc_cell_idx(s, i, j + 1))] = Dy_p1(s, i, j);
      s->val[orc_pos(s, cur, orc_cell_idx(s, i - 1, j))] = Dx_m1(s, i, j);
      s->val[orc_pos(s, cur, orc_cell_idx(s, i + 1, j))] = Dx_p1(s, i, j);
    }
  }
}

/* ImplicitEuler::assemble_system, include/time_disc.hpp:291-339.  The right
 * ghost rows take the LEFT condition's implicit_euler_term (:335-336). */
ORC_API void orc_ie_assemble_system(orc_ie *s, double dt) {
  const int cx = s->cx, cy = s->cy;
  orc_ie_apply_bcs(s);
  for (int i = 0; i < cx; i++) {
    for (int j = 0; j < cy; j++) {
      const long cur = orc_cell_idx(s, i, j);
      s->val[orc_pos(s, cur, cur)] = Dx_0(s) + Dy_0(s) + 1.0 / dt;
      s->sol_vec[cur] = orc_flux_integral(s, i, j) + orc_source_fd(s, i, j);
    }
  }
  /* implicit_euler_term() = 2 w - 1, boundary_conds.hpp:77-81 */
  const double term_top = 2.0 * s->weight[0][ORC_SIDE_TOP] - 1.0;
  const double term_bottom = 2.0 * s->weight[0][ORC_SIDE_BOTTOM] - 1.0;
  const double term_left = 2.0 * s->weight[0][ORC_SIDE_LEFT] - 1.0;
  for (int i = 0; i < cx; i++) {
    const long top = orc_cell_idx(s, i, cy);
    s->sol_vec[top] = 0.0;
    s->val[orc_pos(s, top, orc_cell_idx(s, i, cy - 1))] = 1.0;
    s->val[orc_pos(s, top, top)] = term_top;
    const long bot = orc_cell_idx(s, i, -1);
    s->sol_vec[bot] = 0.0;
    s->val[orc_pos(s, bot, orc_cell_idx(s, i, 0))] = 1.0;
    s->val[orc_pos(s, bot, bot)] = term_bottom;
  }
  for (int j = 0; j < cy; j++) {
    const long left = orc_cell_idx(s, -1, j);
    s->sol_vec[left] = 0.0;
    s->val[orc_pos(s, left, orc_cell_idx(s, 0, j))] = 1.0;
    s->val[orc_pos(s, left, left)] = term_left;
    const long right = orc_cell_idx(s, cx, j);
    s->sol_vec[right] = 0.0;
    s->val[orc_pos(s, right, orc_cell_idx(s, cx - 1, j))] = 1.0;
    s->val[orc_pos(s, right, right)] = term_left;
  }
}

/* ImplicitEuler::timestep, include/time_disc.hpp:277-289.  The caller sets the
 * bc values for the current time first.  max_delta follows std::max's
 * comparison, `(a < b) ? b : a` with a = |delta|. */
ORC_API double orc_ie_timestep(orc_ie *s, double dt) {
  orc_ie_assemble_system(s, dt);
  double max_delta = 0.0;
  const double *delta = orc_gmres_solve(s->solver, s->op, s->sol_vec);
  for (int i = 0; i < s->cx; i++) {
    for (int j = 0; j < s->cy; j++) {
      const double d = delta[orc_cell_idx(s, i, j)];
      AT(s, 0, i, j) += d;
      const double a = fabs(d);
      max_delta = (a < max_delta) ? max_delta : a;
    }
  }
  s->time += dt;
  return max_delta;
}

/* ------------------------------------------------------------------------ */
/* ImplicitEulerApproxFact, include/time_disc.hpp:388-540                    */
/* (I + dt Dy)(I + dt Dx) dT = dt (FI + S): one tridiagonal solve per        */
/* column along y, then one per row along x.                                 */
/* ------------------------------------------------------------------------ */

/* assemble_system_dy, include/time_disc.hpp:487-507; out is (cy+2) x 3 */
ORC_API void orc_af_system_dy(const orc_ie *s, double dt, int i, double *out) {
  const int rows = s->cy + 2;
  out[0] = orc_snan();
  out[1] = 1.0;
  out[2] = 2.0 * s->weight[0][ORC_SIDE_BOTTOM] - 1.0;
  out[(rows - 1) * 3 + 0] = 2.0 * s->weight[0][ORC_SIDE_TOP] - 1.0;
  out[(rows - 1) * 3 + 1] = 1.0;
  out[(rows - 1) * 3 + 2] = orc_snan();
  for (int j = 1; j < rows - 1; j++) {
    out[j * 3 + 0] = Dy_m1(s, i, j - 1) * dt;
    out[j * 3 + 1] = Dy_0(s) * dt + 1.0;
    out[j * 3 + 2] = Dy_p1(s, i, j - 1) * dt;
  }
}

/* assemble_system_dx, include/time_disc.hpp:509-529; out is (cx+2) x 3 */
ORC_API void orc_af_system_dx(const orc_ie *s, double dt, int j, double *out) {
  const int rows = s->cx + 2;
  out[0] = orc_snan();
  out[1] = 1.0;
  out[2] = 2.0 * s->weight[0][ORC_SIDE_LEFT] - 1.0;
  out[(rows - 1) * 3 + 0] = 2.0 * s->weight[0][ORC_SIDE_RIGHT] - 1.0;
  out[(rows - 1) * 3 + 1] = 1.0;
  out[(rows - 1) * 3 + 2] = orc_snan();
  for (int i = 1; i < rows - 1; i++) {
    out[i * 3 + 0] = Dx_m1(s, i - 1, j) * dt;
    out[i * 3 + 1] = Dx_0(s) * dt + 1.0;
    out[i * 3 + 2] = Dx_p1(s, i - 1, j) * dt;
  }
}

/* ImplicitEulerApproxFact::timestep, include/time_disc.hpp:443-485.  The
 * returned maximum is taken over row_sol(i), i = 0..cx-1, i.e. WITHOUT the
 * ghost offset that the temperature update uses (:479-480): it covers the
 * left ghost entry and all interior cells but the last of every row. */
ORC_API double orc_af_timestep(orc_ie *s, double dt) {
  const int cx = s->cx, cy = s->cy;
  const int ny = cy + 2, nx = cx + 2;
  const int nmax = nx > ny ? nx : ny;
  double *sys = (double *)malloc(sizeof(double) * 3 * (size_t)nmax);
  double *rhs = (double *)malloc(sizeof(double) * (size_t)nmax);
  double *sol = (double *)malloc(sizeof(double) * (size_t)nmax);
  double *ud = (double *)malloc(sizeof(double) * (size_t)nmax);
  double *partial = (double *)malloc(sizeof(double) * (size_t)cx * cy);
  orc_ie_apply_bcs(s);
  for (int i = 0; i < cx; i++) {
    orc_af_system_dy(s, dt, i, sys);
    rhs[0] = 0.0;
    rhs[ny - 1] = 0.0;
    for (int j = 0; j < cy; j++) {
      rhs[j + 1] = (orc_flux_integral(s, i, j) + orc_source_fd(s, i, j)) * dt;
    }
    orc_sparse_tridiag_solve(sys, rhs, sol, ud, ny);
    for (int j = 0; j < cy; j++) partial[(size_t)i * cy + j] = sol[j + 1];
  }
  double max_delta = 0.0;
  for (int j = 0; j < cy; j++) {
    orc_af_system_dx(s, dt, j, sys);
    rhs[0] = 0.0;
    rhs[nx - 1] = 0.0;
    for (int i = 0; i < cx; i++) rhs[i + 1] = partial[(size_t)i * cy + j];
    orc_sparse_tridiag_solve(sys, rhs, sol, ud, nx);
    for (int i = 0; i < cx; i++) {
      const double a = fabs(sol[i]);
      max_delta = (a < max_delta) ? max_delta : a; /* std::max(a, max_delta) */
      AT(s, 0, i, j) += sol[i + 1];
    }
  }
  s->time += dt;
  free(sys); free(rhs); free(sol); free(ud); free(partial);
  return max_delta;
}

ORC_API double orc_ie_time(const orc_ie *s) { return s->time; }
ORC_API long orc_ie_num_eqns(const orc_ie *s) { return s->n; }
ORC_API int64_t orc_ie_nnz(const orc_ie *s) { return s->rowptr[s->n]; }
ORC_API const double *orc_ie_field(const orc_ie *s, int f) { return s->fld[f]; }
ORC_API double *orc_ie_field_mut(orc_ie *s, int f) { return s->fld[f]; }
ORC_API const int64_t *orc_ie_rowptr(const orc_ie *s) { return s->rowptr; }
ORC_API const int32_t *orc_ie_col(const orc_ie *s) { return s->col; }
ORC_API const double *orc_ie_val(const orc_ie *s) { return s->val; }
ORC_API const double *orc_ie_sol_vec(const orc_ie *s) { return s->sol_vec; }
ORC_API orc_gmres *orc_ie_solver(orc_ie *s) { return s->solver; }
ORC_API const orc_op *orc_ie_op(const orc_ie *s) { return s->op; }
ORC_API double orc_ie_dx(const orc_ie *s) { return s->dx; }
ORC_API double orc_ie_dy(const orc_ie *s) { return s->dy; }
