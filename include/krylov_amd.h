/*
 * krylov_amd.h -- C ABI of the MI355X (gfx950) implementation of the
 * implicit-Euler / GMRES path of mfdeakin/krylov.
 *
 * All entry points are plain C: opaque handles, plain pointers and sizes, an
 * int status as return value (0 = KRY_OK), never an exception.  Host pointers
 * are caller-owned; every device buffer is owned by the handle that created
 * it.  Work is queued on the context's HIP stream; entry points that hand
 * results back to host memory synchronise that stream before returning, the
 * *_device entry points do not.
 *
 * Each declaration names the reference interface it stands in for
 * (paths relative to the reference checkout).
 *
 * Vector index spaces
 *   "cell"   : the unknown numbering of ImplicitEuler::cell_idx
 *              (include/time_disc.hpp:341-376), N = (cx+2)(cy+2)-4 entries.
 *   "padded" : the (cx+2) x (cy+2) row-major layout of the mesh fields
 *              (include/mesh.hpp:77-120), j fastest, including the four unused
 *              corner entries, which every kernel keeps at exactly 0.
 *   The stencil operator and the solver attached to a kry_ie work in padded
 *   space on the device; the kry_ie getters convert to cell numbering.
 */
#ifndef KRYLOV_AMD_H_
#define KRYLOV_AMD_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct kry_ctx kry_ctx;     /* device + stream + reduction scratch          */
typedef struct kry_op kry_op;       /* linear operator (5-band stencil or CSR)      */
typedef struct kry_gmres kry_gmres; /* GMRESSolver<m> / GMRESTDPSolver<m> state     */
typedef struct kry_ie kry_ie;       /* ImplicitEuler<SecondOrderCentered, Solver>   */

enum kry_status {
  KRY_OK = 0,
  KRY_ERR_INVALID = 1, /* bad argument / shape mismatch                  */
  KRY_ERR_HIP = 2,     /* a HIP runtime call failed                      */
  KRY_ERR_STATE = 3,   /* call order violated (e.g. arnoldi before init) */
  KRY_ERR_NO_DEVICE = 4
};

/* Field / side selectors: the order of the BC arguments of the TimeDisc
 * constructor, include/time_disc.hpp:23-50. */
enum kry_field { KRY_FIELD_T = 0, KRY_FIELD_U = 1, KRY_FIELD_V = 2 };
enum kry_side { KRY_SIDE_BOTTOM = 0, KRY_SIDE_TOP = 1, KRY_SIDE_LEFT = 2, KRY_SIDE_RIGHT = 3 };

/* Solver flags (kry_gmres_create, kry_ie_create). */
enum kry_gmres_flags {
  KRY_GMRES_PLAIN = 0,
  KRY_GMRES_TDP = 1,           /* GMRESTDPSolver: tridiagonal right preconditioner   */
  KRY_GMRES_FIXED_PRECOND = 2, /* opt-in, not in the reference: return M^-1 u        */
  KRY_GMRES_EXTENDED = 8       /* opt-in, not in the reference: restarted GMRES(m)
                                  with Givens-rotation least squares and a residual
                                  monitor on the device (see kry_gmres_set_tolerance);
                                  with KRY_GMRES_TDP the preconditioner is applied to
                                  the result, i.e. proper right preconditioning      */
};

/* Message of the last failure on the calling thread ("" if none). */
const char *kry_last_error(void);
/* Library version string. */
const char *kry_version(void);
int kry_device_count(int *count);

/* ---- context -------------------------------------------------------------- */
/* stream == NULL: the context creates (and owns) a stream on `device`.
 * Otherwise `stream` is a hipStream_t the caller keeps alive. */
int kry_ctx_create(int device, void *stream, kry_ctx **out);
void kry_ctx_destroy(kry_ctx *ctx);
int kry_ctx_synchronize(kry_ctx *ctx);
/* Free and total memory of the context's device in bytes (either may be NULL). */
int kry_ctx_memory_info(kry_ctx *ctx, int64_t *free_bytes, int64_t *total_bytes);
/* hipStream_t of the context (for event timing by the caller). */
void *kry_ctx_stream(kry_ctx *ctx);
/* Row-block decomposition over several processes (one per GPU).  The library
 * owns the algorithm and calls back for the two collectives it needs; both act
 * on DEVICE pointers and must be ordered after prior work on `stream` and
 * before later work on it (e.g. RCCL calls enqueued on that stream).  They
 * return 0 on success.  Has no counterpart in the reference (single process).
 *   allreduce: in-place reduction of count doubles over all ranks
 *   sendrecv : send count doubles to send_peer and receive count doubles from
 *              recv_peer (a peer < 0 means: no such transfer) */
enum kry_reduce_op { KRY_REDUCE_SUM = 0, KRY_REDUCE_MAX = 1 };
typedef int (*kry_allreduce_fn)(void *user, double *dev_buf, int64_t count, int op, void *stream);
typedef int (*kry_sendrecv_fn)(void *user, const double *send_dev, int send_peer, double *recv_dev,
                               int recv_peer, int64_t count, void *stream);
int kry_ctx_set_comm(kry_ctx *ctx, int rank, int nranks, kry_allreduce_fn allreduce,
                     kry_sendrecv_fn sendrecv, void *user);
/* Alternative to the callbacks: the library issues the collectives itself
 * through RCCL on the context's stream (ncclAllReduce; the four halo transfers
 * of an exchange as one ncclGroup), which also lets a captured step (hipGraph)
 * contain them.  One rank obtains a 128-byte id with kry_rccl_unique_id and
 * hands it to all ranks by whatever launcher is in use (MPI, torch.distributed
 * store, a file); every rank then calls kry_ctx_set_comm_rccl (collective:
 * returns when all nranks have joined).  librccl is loaded on first use; a copy
 * already present in the process is reused.  No counterpart in the reference. */
int kry_rccl_unique_id(unsigned char id[128]);
int kry_ctx_set_comm_rccl(kry_ctx *ctx, int rank, int nranks, const unsigned char id[128]);
int kry_ctx_comm_is_native(const kry_ctx *ctx);
/* Collective check of whatever communication is installed: sum and max
 * all-reduce of the rank numbers and a ring exchange, verified on every rank. */
int kry_ctx_comm_selftest(kry_ctx *ctx);
int kry_ctx_rank(const kry_ctx *ctx);
int kry_ctx_nranks(const kry_ctx *ctx);
/* Optional instrumentation (off by default): bracket every kernel launch with
 * HIP events on the context's stream and accumulate the elapsed time per
 * kernel class (enum kry_kernel_class below).  get_kernel_times synchronises
 * the stream, returns the totals since the previous call and resets them. */
int kry_ctx_set_kernel_timing(kry_ctx *ctx, int on);
int kry_ctx_get_kernel_times(kry_ctx *ctx, double ms[], int launches[]);
/* Self-test of an arithmetic shortcut, without a counterpart in the reference.  Kernels that evaluate the
 * off-diagonal entries of the implicit-Euler operator on the fly (SecondOrderCentered::Dx_m1 / Dx_p1 / Dy_m1 /
 * Dy_p1, include/space_disc.hpp:83-113: four numerators over the mesh widths dx, dy per cell) replace the
 * division instruction sequence by five multiply-adds with a correctly rounded reciprocal, inside a guarded
 * range in which that yields the correctly rounded quotient, and fall back to the division outside it.  This
 * entry point runs that code for `count` groups of four pseudo-random numerators and compares every quotient
 * with the device's division, bit for bit.  mode 0: numerators within 2^+-40 of 1; 1: arbitrary bit patterns
 * (zeros, subnormals, infinities, NaNs); 2: quotients around the limits of the guarded range; 3: quotients next
 * to powers of two and to midpoints between neighbouring numbers.  *mismatches: quotients that differ (must be
 * 0); *fast_groups (may be NULL): groups served by the shortcut. */
int kry_selftest_shared_division(kry_ctx *ctx, double dx, double dy, uint64_t seed, int64_t count, int mode,
                                 int64_t *mismatches, int64_t *fast_groups);

/* ---- operators: stand-ins for the dense `matrix` handed to Solver::solve ----
 * reference: dot(vector&, const matrix&, const vector&), include/matrix.hpp:23-33 */

/* Five coefficient bands on a rows x cols padded grid (host arrays of
 * rows*cols doubles each): y[k] = w[k] x[k-cols] + s[k] x[k-1] + c[k] x[k]
 * + n[k] x[k+1] + e[k] x[k+cols], summed in that (ascending column) order. */
int kry_op_create_stencil(kry_ctx *ctx, int rows, int cols, const double *w,
                          const double *s, const double *c, const double *n,
                          const double *e, kry_op **out);
/* CSR with ascending column indices per row. */
int kry_op_create_csr(kry_ctx *ctx, int64_t n, const int64_t *rowptr,
                      const int32_t *col, const double *val, kry_op **out);
/* Row-major dense n x n matrix (the reference's `matrix`); stored entries that
 * are exactly 0.0 are dropped, the rest becomes CSR.
 * reference: Matrix.from_array, src/bindpy.cpp:618-629 */
int kry_op_create_dense(kry_ctx *ctx, int64_t n, const double *a, kry_op **out);
void kry_op_destroy(kry_op *op);
int64_t kry_op_size(const kry_op *op);
/* y = A x for host vectors of kry_op_size entries.
 * reference: include/matrix.hpp:23-33 */
int kry_op_apply(kry_op *op, const double *x_host, double *y_host);
/* Same on device pointers, asynchronous on the context's stream. */
int kry_op_apply_device(kry_op *op, const double *x_dev, double *y_dev);

/* ---- GMRES: include/solver.hpp:115-239 (GMRESSolver<m>), :279-333 (TDP) ----- */
int kry_gmres_create(kry_ctx *ctx, int64_t n, int m, int flags, kry_gmres **out);
void kry_gmres_destroy(kry_gmres *g);
/* GMRESSolver::initial_subspace, include/solver.hpp:168-174; *beta = ||b||. */
int kry_gmres_initial_subspace(kry_gmres *g, const double *b_host, double *beta);
/* GMRESTDPSolver::assemble_preconditioner, include/solver.hpp:295-314.
 * tdiags_host (may be NULL) receives the n x 3 band copy (entries (0,0) and
 * (n-1,2) are NaN as in the reference). */
int kry_gmres_assemble_preconditioner(kry_gmres *g, kry_op *op, double *tdiags_host);
/* SparseTriDiagSolver::solve as the preconditioner uses it
 * (include/solver.hpp:246-273 called from :316-321): out = M^-1 in for the
 * tridiagonal assembled by kry_gmres_assemble_preconditioner.  Vectors are in
 * the operator's own layout (padded for a stencil operator), length n. */
int kry_gmres_apply_preconditioner(kry_gmres *g, const double *in_host, double *out_host);
/* GMRESSolver::add_arnoldi (include/solver.hpp:185-194) or the TDP override
 * (:316-328), incl. gram_schmidt (:196-211); *h_next = H(j+1, j). */
int kry_gmres_add_arnoldi(kry_gmres *g, kry_op *op, double *h_next);
/* GMRESSolver::solve / GMRESTDPSolver::solve, include/solver.hpp:124-161,
 * :290-293; x_host may be NULL (fetch later with kry_gmres_prev_result). */
int kry_gmres_solve(kry_gmres *g, kry_op *op, const double *b_host, double *x_host);
/* Same with device pointers (n entries each), asynchronous, no host sync:
 * one whole fixed-m cycle queued on the stream. x_dev may be NULL. */
int kry_gmres_solve_device(kry_gmres *g, kry_op *op, const double *b_dev, double *x_dev);
/* GMRESSolver::subspace_vector, include/solver.hpp:163-166 */
int kry_gmres_subspace_vector(kry_gmres *g, int i, double *out_host);
/* GMRESSolver::hessenberg, include/solver.hpp:176-181 */
int kry_gmres_hessenberg(kry_gmres *g, int i, int j, double *out);
/* Whole (m+1) x m row-major Hessenberg array. */
int kry_gmres_hessenberg_matrix(kry_gmres *g, double *out_host);
/* The reduced least-squares solve of GMRESSolver::solve on its own
 * (include/solver.hpp:142-151: sys = H^T H, rhs_i = beta H(0,i), then
 * LUSolver::solve, :34-96, un-pivoted Doolittle LU) for a caller-supplied
 * (num_iters+1) x num_iters row-major Hessenberg array: y_host receives the
 * num_iters coefficients.  Every entry is produced by the reference's sequence
 * of operations, so the result equals the reference's bit for bit.  Replaces the
 * solver's Hessenberg matrix and beta. */
int kry_gmres_reduced_solve(kry_gmres *g, const double *H_host, double beta, double *y_host);
/* Solver::prev_result, include/solver.hpp:21 (signalling NaN before a solve). */
int kry_gmres_prev_result(kry_gmres *g, double *out_host);
/* Device pointer to the n-entry result vector. */
const double *kry_gmres_result_device(kry_gmres *g);
/* Number of modified-Gram-Schmidt links fused into one sweep over the work
 * vector: 1, 2, 3, 4, 8, 16 or 32 (default 32).  1 is the reference's plain chain, one link per
 * sweep (include/solver.hpp:198-207); larger groups take the group's dot
 * products in one pass and recover the same coefficients through the group's
 * Gram entries, h_k = <w, v_k> - sum_{a<k} h_a <v_a, v_k>, which are computed
 * once, in the step that creates v_k, and carried in a small matrix. */
int kry_gmres_set_mgs_block(kry_gmres *g, int k);
int kry_gmres_get_mgs_block(const kry_gmres *g);
/* One-sweep Arnoldi steps.  The chain of include/solver.hpp:196-211 reads the
 * basis twice per step (dot products, then updates).  From step `first_step` of
 * every cycle on, kry_gmres_solve* instead uses sweeps that read it once: each
 * normalises the newest vector, applies all coefficients, applies the operator
 * to the un-normalised result and takes every dot product of the following
 * step, whose coefficients are then recovered from those sums and the carried
 * Gram matrix as above.  Same Krylov space and Hessenberg matrix up to rounding.
 * Available for a 5-band operator without couplings across the ends of a grid
 * row (on several ranks: the row-block decomposition of kry_ie_create_slab),
 * no preconditioner and 3 <= m <= 50; otherwise, and for kry_gmres_add_arnoldi,
 * the chain is used.  With KRY_GMRES_EXTENDED the residual monitor sees column i
 * one sweep later than with the chain (H(i+1,i) is the norm the next sweep's sums
 * deliver): the iterate and the reported iteration count are the same, one sweep
 * more has run when the solver stops.
 * first_step < 0 switches the sweeps off; values above min(m - 2, 31) are clamped.
 * Default 2, or 0 for the solver of a step object on one rank (environment
 * override: KRY_FUSED_FROM).  The getter returns -1 when the sweeps are off or
 * unavailable for this solver. */
int kry_gmres_set_one_sweep(kry_gmres *g, int first_step);
int kry_gmres_get_one_sweep(const kry_gmres *g);
/* Storage of the Krylov basis, without a counterpart in the reference (its basis is a
 * std::vector of vectors, include/solver.hpp:236).  When every step of a cycle is a
 * one-sweep step (first_step 0) of a step object's operator with generated bands on one
 * rank, the solver stores the basis of that solve strip by strip: the columns of the grid
 * are cut into the strips the sweeps' workgroups walk, and a strip's rows follow each other
 * in memory as aligned 2 KB lines (the two columns on either side of a strip are stored by
 * both neighbours), so that a workgroup streams consecutive lines instead of one piece per
 * 64 KB grid row.  Results are the same bits; kry_gmres_subspace_vector and the solution
 * are returned in grid order.  Returns 1 if the basis of the last solve was stored that way,
 * else 0.  Environment KRY_STRIP_MAJOR=0 switches it off. */
int kry_gmres_basis_strip_major(const kry_gmres *g);
/* The other storage without a counterpart in the reference: a solve on one rank whose every step runs the
 * Gram-Schmidt chain (line preconditioner, num_iters above the one-sweep range, CSR operators) may keep its
 * basis chunk-interleaved -- 512 values of vector 0, the same 512 of vector 1, ... -- so that the chain's
 * kernels, which read up to 64 vectors per pass, stream one contiguous piece per chunk instead of 64 pieces
 * in 64 different memory pages.  It does where that pays: from num_iters 32 on, and for the line
 * preconditioner on meshes where the preconditioner kernel also normalises its operand (the copy of the
 * newest vector that the layout otherwise costs then falls away).  Same bits; kry_gmres_subspace_vector
 * returns contiguous vectors, and the step-wise interface (initial_subspace / add_arnoldi) works on
 * contiguous vectors.  Returns 1 if the basis of the last solve was stored that way.  Environment
 * KRY_CHAIN_INTERLEAVE (read when a solver is created): 0 never, 2 wherever possible. */
int kry_gmres_basis_chunk_interleaved(const kry_gmres *g);
/* How the last Arnoldi step of a cycle obtains H(m, m-1) = ||w - sum_k h_k v_k||, the normalisation
 * at the end of gram_schmidt (include/solver.hpp:209); no counterpart in the reference, which
 * always forms the vector.  The vector of that step is not part of the solution (x = sum of m
 * basis vectors, :153-159), only its norm enters the reduced problem, and modified Gram-Schmidt
 * takes ||w||^2 down by h_k^2 per link: with one-sweep steps the solver of a step object
 * therefore takes that norm as sqrt(||A v_{m-1}||^2 - sum_k h_k^2) from sums the last sweep has
 * already formed and skips the closing pass over the basis, unless less than 1/64 of ||A v||^2
 * is left (cancellation, e.g. near a breakdown), in which case the device itself falls back to
 * forming the vector.  kry_gmres_subspace_vector(g, m) forms the vector on demand, with the same
 * bits as the closing pass.  Stand-alone solvers (kry_gmres_create) always form the vector.
 * on = 0 switches the shortcut off (environment: KRY_NORM_FROM_SUMS=0).  The getter returns 1
 * when the shortcut is available and on; kry_gmres_last_norm_from_sums reads back (it waits for
 * the device) whether the last solve's final cycle used it (1) or fell back (0). */
int kry_gmres_set_norm_from_sums(kry_gmres *g, int on);
int kry_gmres_get_norm_from_sums(const kry_gmres *g);
int kry_gmres_last_norm_from_sums(kry_gmres *g, int *used);
/* Tuning knob without a counterpart in the reference: from this step of a cycle on
 * the one-sweep steps run as workgroups of 512 threads whose halves split the basis
 * vectors (two wavefronts per SIMD where the 256-thread kernel fits one); every
 * result bit is the same.  Process-wide; first_step < 0 restores the default (18;
 * env KRY_SPLIT_FROM), values below 12 mean 12, values beyond the last step: never.
 * Returns the value in effect. */
int kry_set_one_sweep_split(int first_step);
/* The range of steps in effect for the 512-thread form: first and last step of a cycle it serves. */
int kry_get_one_sweep_split(int *first_step, int *last_step);
int64_t kry_gmres_size(const kry_gmres *g);
int kry_gmres_iterations(const kry_gmres *g);
/* KRY_GMRES_EXTENDED only; no counterpart in the reference, whose solve always
 * runs num_iters Arnoldi steps from x0 = 0 and solves the normal equations
 * (include/solver.hpp:124-161).  A solve stops as soon as the residual norm
 * ||b - A x|| <= rtol * ||b|| (measured through the rotated Hessenberg system,
 * checked after every Arnoldi step on the device without a host round trip)
 * and restarts from the current iterate after num_iters steps, at most
 * max_cycles cycles.  Defaults: rtol = 1e-10, max_cycles = 20.  The host reads
 * the solver state once per cycle. */
int kry_gmres_set_tolerance(kry_gmres *g, double rtol, int max_cycles);
/* Outcome of the last extended solve: Arnoldi steps over all cycles, cycles
 * started, whether the tolerance was met, final residual norm / ||b||.
 * Any pointer may be NULL. */
int kry_gmres_last_stats(kry_gmres *g, int *iterations, int *cycles, int *converged,
                         double *rel_residual);
/* Kernel-level timing hooks for the benchmark: algorithmic bytes moved by the
 * kernels of one full cycle, per class (see kry_kernel_class). */
enum kry_kernel_class {
  KRY_K_SPMV = 0,      /* operator apply fused with <w, v_0>                    */
  KRY_K_MGS = 1,       /* full Gram-Schmidt sweep: 4 projections applied, 4 dots taken */
  KRY_K_MGS_PART = 2,  /* Gram-Schmidt sweep over a shorter group               */
  KRY_K_MGS_LAST = 3,  /* last sweep of a step: projections applied, <w, w>     */
  KRY_K_NORMALIZE = 4, /* v = w / ||w|| (and v_0 = b / ||b||)                   */
  KRY_K_COMBINE = 5,   /* x = sum y_i v_i                                       */
  KRY_K_PRECOND = 6,   /* tridiagonal line solves                               */
  KRY_K_ASSEMBLY = 7,  /* ghost fill, right-hand side, state update             */
  KRY_K_OTHER = 8,     /* reduced m x m solve                                   */
  KRY_K_FUSED = 9,     /* one-sweep Arnoldi step: normalise, orthogonalise, apply A, dots */
  KRY_K_COUNT = 10
};
int kry_gmres_cycle_bytes(const kry_gmres *g, const kry_op *op, double bytes[KRY_K_COUNT],
                          int launches[KRY_K_COUNT]);

/* ---- the implicit-Euler step: include/time_disc.hpp ------------------------- */
/* Mesh + TimeDisc + ImplicitEuler constructor, first half
 * (src/mesh.cpp:8-68, include/time_disc.hpp:17-134, :196-245).
 * T0/u0/v0: interior values sampled at the cell medians, cx*cy row-major
 * (i slow, j fast).  bc_is_dirichlet[field*4+side].  m, flags: the solver. */
int kry_ie_create(kry_ctx *ctx, double c1x, double c1y, double c2x, double c2y,
                  int cx, int cy, const double *T0, const double *u0,
                  const double *v0, double diffusion, double reynolds,
                  double prandtl, double eckert, const int *bc_is_dirichlet,
                  int m, int flags, kry_ie **out);
/* The same for ONE RANK of a row-block decomposition: this rank owns the padded
 * rows [row_begin, row_end) of the (cx+2) x (cy+2) mesh (row 0 and row cx+1 are
 * the left / right ghost columns of the reference's layout).  T0/u0/v0 hold
 * the interior cells of the owned rows only: cell rows
 * i = max(row_begin,1)-1 .. min(row_end,cx+1)-2, cy values each.  The context
 * must have collectives installed (kry_ctx_set_comm) unless the range is the
 * whole mesh.  Every rank calls the step functions collectively. */
int kry_ie_create_slab(kry_ctx *ctx, double c1x, double c1y, double c2x, double c2y,
                       int cx, int cy, int row_begin, int row_end, const double *T0,
                       const double *u0, const double *v0, double diffusion,
                       double reynolds, double prandtl, double eckert,
                       const int *bc_is_dirichlet, int m, int flags, kry_ie **out);
void kry_ie_destroy(kry_ie *ie);
/* bc(x, y, t) sampled by the host at the boundary-cell coordinates of
 * include/time_disc.hpp:55-76 for one field and side (cx values for
 * bottom/top, cy for left/right); consumed by the next ghost fill. */
int kry_ie_set_bc_values(kry_ie *ie, int field, int side, const double *vals, int len);
/* TimeDisc::apply_bcs, include/time_disc.hpp:146-163 (+ the constructor's
 * view wiring, :51-134) */
int kry_ie_apply_bcs(kry_ie *ie);
/* ImplicitEuler constructor, second half (include/time_disc.hpp:246-274):
 * ghost fill at t = 0 and the four off-diagonal bands (never refreshed). */
int kry_ie_init_offdiag(kry_ie *ie);
/* The off-diagonals of an interior cell are functions of the velocities next to it
 * at construction time (SecondOrderCentered::Dx_m1 .. Dy_p1, include/space_disc.hpp:
 * 83-113, frozen by include/time_disc.hpp:246-274) and all interior cells share
 * one diagonal value (:294-302).  With on != 0 (the default) the one-sweep Arnoldi
 * steps evaluate those entries from a copy of u and v taken by kry_ie_init_offdiag
 * (same operations, same bits) instead of reading five stored bands: two streamed
 * arrays instead of five; the entries of ghost rows and columns (a 1 towards the
 * interior, the diagonal 2w - 1 of their side, :304-337) follow from their position.
 * Results are bit-identical either way.  get: 1 when the evaluation is in use. */
int kry_ie_set_generated_bands(kry_ie *ie, int on);
int kry_ie_get_generated_bands(const kry_ie *ie);
/* ImplicitEuler::assemble_system, include/time_disc.hpp:291-339 */
int kry_ie_assemble_system(kry_ie *ie, double dt);
/* ImplicitEuler::timestep, include/time_disc.hpp:277-289.  *max_delta is the
 * reference's value: the fold max_delta = std::max(|delta|, max_delta) over the
 * interior cells in (i, j) order (:284), whose comparisons are all false for a
 * NaN, i.e. the largest |delta| over the cells BEHIND the last NaN cell, and NaN
 * only if the last interior cell is NaN. */
int kry_ie_timestep(kry_ie *ie, double dt, double *max_delta);
/* The second half of that step on its own (include/time_disc.hpp:281-286) for a
 * caller-supplied delta: T += delta on interior cells (ghost entries are
 * ignored, as in the reference) and the same maximum.  delta_local: the owned
 * rows in padded layout, kry_ie_local_size doubles; it replaces the solver's
 * previous result.  Time does not advance.  Collective on a row-block
 * decomposition. */
int kry_ie_apply_delta(kry_ie *ie, const double *delta_local, double *max_delta);
/* No counterpart in the reference.  The same state update for callers that
 * cannot allocate device memory (code under hipGraph capture, tight serving
 * loops, harnesses that forbid hipMalloc): no step object, nothing is allocated
 * or freed, nothing waits for the device.  T_dev and delta_dev are
 * (cx+2) x (cy+2) device arrays in padded layout (whole mesh), *max_delta_dev
 * receives the maximum on the device; the work is queued on the context's
 * stream.  The kernels' partial results live in a scratch array of 32.8 KB
 * that is part of the library's device image (every GPU the library runs on
 * holds one copy for the life of the process): calls that overlap on one GPU
 * race on it, so keep them on one stream or order them yourself.  Meshes
 * with more than KRY_NOALLOC_MAX_PADDED_CELLS padded cells are rejected
 * (KRY_ERR_INVALID). */
#define KRY_NOALLOC_MAX_PADDED_CELLS 2147483000LL
int kry_apply_delta_noalloc(kry_ctx *ctx, int cx, int cy, double *T_dev, const double *delta_dev,
                            double *max_delta_dev);
/* The same step queued without reading max|delta| back (for back-to-back
 * steps with time-independent boundary values); fetch with kry_ie_last_max_delta. */
int kry_ie_timestep_async(kry_ie *ie, double dt);
int kry_ie_last_max_delta(kry_ie *ie, double *max_delta);
/* ImplicitEulerApproxFact::timestep (include/time_disc.hpp:443-485): ghost
 * fill, one tridiagonal solve per mesh column along y with right-hand side
 * dt (flux_integral + source_fd), one per mesh row along x, T += dT; returns
 * the reference's maximum (taken over solution entries 0..cx-1 of every x
 * line, :479).  The step object must have been created with m = 0 (it uses
 * the operator storage for the line systems) and own the whole mesh. */
int kry_ie_timestep_approx_fact(kry_ie *ie, double dt, double *max_delta);
/* assemble_system_dy(dt, i) (along_y != 0, include/time_disc.hpp:487-507) or
 * assemble_system_dx(dt, j) (:509-529) for the current ghost values: rows x 3
 * row-major (lower, diagonal, upper), rows = cy + 2 or cx + 2; the two
 * entries the reference never uses are signalling NaN as there. */
int kry_ie_get_af_system(kry_ie *ie, int along_y, int index, double dt, double *out);
/* Replay the step through a captured hipGraph (1) or launch kernels eagerly (0). */
int kry_ie_use_graph(kry_ie *ie, int on);
/* TimeDisc::cur_time, include/time_disc.hpp:138 */
double kry_ie_time(const kry_ie *ie);
int64_t kry_ie_num_eqns(const kry_ie *ie);
/* Mesh::array_Temp / array_vel_u / array_vel_v (include/mesh.hpp:251-257):
 * (cx+2)*(cy+2) doubles, corners are NaN. */
int kry_ie_get_field(kry_ie *ie, int field, double *out_padded);
/* ImplicitEuler::sys (include/time_disc.hpp:378) as a dense N x N row-major
 * matrix in cell numbering (small meshes only: N*N doubles). */
int kry_ie_get_dense_system(kry_ie *ie, double *out);
/* The five bands (w, s, c, n, e), each (cx+2)*(cy+2) doubles, padded layout. */
int kry_ie_get_bands(kry_ie *ie, double *out5);
/* ImplicitEuler::sol (include/time_disc.hpp:379), N entries, cell numbering. */
int kry_ie_get_rhs(kry_ie *ie, double *out);
/* ImplicitEuler::delta (include/time_disc.hpp:380), N entries, cell numbering. */
int kry_ie_get_delta(kry_ie *ie, double *out);
/* Owned entries of this rank, (row_end - row_begin) * (cy+2). */
int64_t kry_ie_local_size(const kry_ie *ie);
/* Owned rows of a quantity in padded layout (kry_ie_local_size doubles). */
enum kry_local_quantity {
  KRY_LOCAL_FIELD_T = 0, KRY_LOCAL_FIELD_U = 1, KRY_LOCAL_FIELD_V = 2, KRY_LOCAL_RHS = 3,
  KRY_LOCAL_DELTA = 4, KRY_LOCAL_BAND_W = 5, KRY_LOCAL_BAND_S = 6, KRY_LOCAL_BAND_C = 7,
  KRY_LOCAL_BAND_N = 8, KRY_LOCAL_BAND_E = 9
};
int kry_ie_get_local(kry_ie *ie, int what, double *out);
/* The solver / operator owned by the step object (borrowed handles). */
kry_gmres *kry_ie_solver(kry_ie *ie);
kry_op *kry_ie_operator(kry_ie *ie);
/* Device pointer of the padded right-hand side (for benchmarks). */
const double *kry_ie_rhs_device(kry_ie *ie);

#ifdef __cplusplus
}
#endif
#endif /* KRYLOV_AMD_H_ */
