// Internal declarations shared by the kernel and API translation units.
#ifndef KRY_INTERNAL_H_
#define KRY_INTERNAL_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "krylov_amd.h"

namespace kry {

constexpr int kBlock = 256;        // threads per workgroup (4 wave64)
constexpr int kMaxGrid = 2048;     // 256 CUs x 8 resident workgroups
constexpr int kMaxParts = 2048;    // partial sums per reduction buffer
constexpr int kMaxM = 256;         // largest Krylov dimension supported
constexpr int kChunk = 512;        // values per chunk of a chunk-interleaved basis (see il1 in kry_device.h)
constexpr int kMgsBlock = 32;      // most Gram-Schmidt links fused into one sweep over w
constexpr int kMaxRed = 2 * kMgsBlock - 1;  // partial-sum arrays per pass: dots + Gram column of the last vector
constexpr int kFusedMaxM = 50;     // largest Krylov dimension served by the one-sweep Arnoldi step
constexpr int kFusedNormSlot = kFusedMaxM;  // coefficient array: h_0..h_i, then ||u_i|| in this slot
constexpr int kFusedCoef = kFusedMaxM + 2;
constexpr int kFusedRed = 2 * kFusedMaxM + 5;  // partial-sum arrays of one fused sweep (upper bound)

// Five coefficient bands of a stencil operator on a rows x cols padded grid.
struct Bands {
  const double *w, *s, *c, *n, *e;
};

// Recipe for the interior entries of the bands of an assembled implicit-Euler operator
// (include/time_disc.hpp:249-274, :294-302): the four off-diagonals of an interior cell
// are functions of the velocities next to it at the time the off-diagonals were formed,
// and all interior cells share one diagonal value.  A kernel that holds a recipe
// evaluates those entries (same operations as k_offdiag, hence the same bits) instead
// of reading them; the entries of ghost rows and columns follow from their position.
// 2^-100 <= |d| < 2^101: the divisors for which the kernels replace divisions by corrections with a correctly
// rounded reciprocal (div_shared_fast in kry_device.h)
__host__ __device__ inline bool shared_divisor_in_range(double d) {
  uint64_t bits;
  __builtin_memcpy(&bits, &d, sizeof bits);
  return (uint32_t)((bits >> 52) & 0x7ffu) - 923u <= 200u;
}

struct BandGen {
  const double *u = nullptr, *v = nullptr;  // velocity snapshots, laid out like the fields (halo row in front and behind)
  const double *cdiag = nullptr;            // the diagonal entry of one interior cell (any entry when the rank owns none)
  double kx = 0.0, ky = 0.0, dx = 1.0, dy = 1.0;  // kx = diffusion / (Re Pr) / dx, ky likewise, as k_offdiag forms them
  // 1 / dx and 1 / dy, correctly rounded (one IEEE division each on the host), and whether both widths lie in the
  // range in which the division by correction is used (see div_shared_fast; else the kernels divide)
  double rdx = 1.0, rdy = 1.0;
  int div_ok = 0;
  int row0 = 0, cx = 0, cy = 0;  // first padded mesh row of the local grid; interior cells of the mesh
  // diagonal entries 2w - 1 of the ghost rows (include/time_disc.hpp:304-337): bottom, top, and
  // left, which the right ghost column shares (:335-336); their only other entry is a 1 towards
  // the interior, and the corner entries (no unknowns) carry a unit diagonal
  double dg_bottom = 0.0, dg_top = 0.0, dg_left = 0.0;
};

// Strip-major storage of a vector on a rows x cols grid: the columns are cut into nstrips strips of W
// owned columns (the strips the one-sweep kernels' workgroups walk), and strip s is stored as a block of
// (rows + 2) rows of kBlock values -- a grid row in front, the rows, one behind; position t of a row holds
// column s W - 2 + t, so the two columns next to a strip on either side are stored twice.
// A single vector (the velocity snapshots): rows follow each other at a pitch of kBlock values, SB = (rows + 2)
// x kBlock.  The Krylov basis: the rows of ALL its vectors are interleaved -- row r of vector 0, row r of vector
// 1, ... (slots of them), then row r + 1 -- so pitch = slots x kBlock, SB = (rows + 2) x pitch, and vector k
// starts k x kBlock values into the array.  A workgroup that walks down its strip then reads, per row, ONE
// contiguous piece holding that row of every vector (instead of one 2 KB piece per vector, each in a different
// memory page: with more than about thirty vectors that overflowed the translation caches of the compute units,
// 2 % of the translations missed from 29 earlier vectors on, next to none below).
struct StripMap {
  int on = 0, W = 0, nstrips = 0, pitch = 0;
  int64_t SB = 0;
};

// Device-resident state of the restarted solver with residual monitor.
struct ExtState {
  double tol_abs;  // rtol * ||b||, fixed by the first cycle
  double res;      // residual norm of the best iterate of the current cycle
  double beta0;    // ||b||
  int stop;        // non-zero: the remaining iteration kernels of the cycle are no-ops
  int iters;       // Arnoldi steps completed in the current cycle
  int converged;   // res <= tol_abs was observed
  int pad;
};

// While an object of this type is alive, the iteration kernels launched by the
// calling thread (operator apply, Gram-Schmidt sweeps, normalisation, line
// solves, partial-sum collapse) receive `stop` and return at once when *stop
// is non-zero on the device.
const int *launch_predicate_current();  // the flag of the innermost live LaunchPredicate of this thread, or nullptr

class LaunchPredicate {
 public:
  explicit LaunchPredicate(const int *stop);
  ~LaunchPredicate();
  LaunchPredicate(const LaunchPredicate &) = delete;
  LaunchPredicate &operator=(const LaunchPredicate &) = delete;

 private:
  const int *saved_;
};

// Launchers return nothing; one that cannot prepare its launch (e.g. the device refuses the
// LDS size a kernel asks for) records the first such failure of the calling thread here,
// and the entry point that queued the work reports it (take: returns it and clears it).
void note_launch_failure(hipError_t e, const char *what);
hipError_t take_launch_failure(const char **what);

// A kernel that needs more than 64 KB of LDS has to be allowed to, once per kernel and
// DEVICE (a process may hold contexts on several).  `done` has one bit per device id.
template <typename K>
inline void allow_lds(K kernel, size_t bytes, std::atomic<uint64_t> &done, const char *what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) dev = 63;  // beyond 63: set it every time
  const uint64_t bit = 1ull << dev;
  if (dev < 63 && (done.load(std::memory_order_acquire) & bit)) return;
  const hipError_t e =
      hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e != hipSuccess) {
    note_launch_failure(e, what);
    return;
  }
  done.fetch_or(bit, std::memory_order_release);
}

// ---- launchers (all asynchronous on `st`) ---------------------------------
int grid_for(int64_t items);
// self-test of the division by a shared divisor (see k_selftest_division); out: two device counters
void launch_selftest_division(hipStream_t st, double dx, double dy, uint64_t seed, int64_t count, int mode,
                              unsigned long long *out);

// part[b] = sum over block b of x[k]^2
void launch_nrm2_part(hipStream_t st, const double *x, int64_t n, double *part, int *nparts);
// nrm = sqrt(sum(part)); dst = src / nrm; *out = nrm
// pitch != 0: dst is a vector of a chunk-interleaved basis (il1); dst2 != nullptr: a second, contiguous copy
void launch_scale_from(hipStream_t st, const double *src, double *dst, const double *part,
                       int nparts, int64_t n, double *out, int64_t pitch = 0, double *dst2 = nullptr);

// y = A x; with kc > 0 also the partial sums of <y, c_k> and <c_a, c_{kc-1}> for the
// kc vectors cbase + k*ld (layout: part[q * nparts + workgroup], dots first)
// gen != nullptr (rows of even length only; otherwise ignored): the band entries are evaluated from the recipe
// instead of being read (grid-order velocity snapshots with a grid row in front and behind)
void launch_spmv_stencil(hipStream_t st, const Bands &b, int64_t n, int cols, const double *x,
                         double *y, int kc, const double *cbase, int64_t ld, double *part,
                         int *nparts, bool x_has_halo, const BandGen *gen = nullptr, int64_t pitch = 0);
// tot[q] = sum_i part[q * np + i], q < nq (one workgroup)
void launch_reduce_parts(hipStream_t st, const double *part, int nq, int np, double *tot);
// (pitch, here and below: the partner vectors cbase + k ld belong to a chunk-interleaved basis)
void launch_spmv_csr(hipStream_t st, const int64_t *rowptr, const int32_t *col, const double *val,
                     int64_t n, const double *x, double *y, int kc, const double *cbase, int64_t ld,
                     double *part, int *nparts, int64_t pitch = 0);

// One blocked Gram-Schmidt sweep: finish the kp coefficients of the previous
// group (pbase, pbase+ld, ...; basis vectors gp .. gp+kp-1) from part_in and the
// carried Gram matrix G (row-major, leading dimension gld; the group's new
// column is stored there), store them at hout[k*hstride], apply them to w, and
// accumulate the dots / last-vector Gram entries of the next kc <= kp vectors
// (cbase, cbase+ld, ...) into part_out; kc == 0: <w, w> instead.
// block sizes (links per sweep) for which sweep kernels exist
bool mgs_block_supported(int k);
void launch_mgs_block(hipStream_t st, int kp, int kc, double *w, const double *pbase,
                      const double *cbase, int64_t ld, const double *part_in, int np_in,
                      double *part_out, int *np_out, double *hout, int hstride, int64_t n, double *G,
                      int gld, int gp, int64_t pitch = 0);

// y = argmin ||beta e1 - H y|| through the normal equations and an un-pivoted
// Doolittle LU, in the reference's loop order.  work: 3*m*m + 2*m doubles.
void launch_small_solve(hipStream_t st, const double *H, const double *beta, int m, double *y,
                        double *work);
// x[k] = sum_i y[i] * V[i*ld + k], accumulated in order i = 0..m-1 from 0.0 (or
// from x[k] with accumulate); count != nullptr: only the first min(m, *count) terms
// sigma != nullptr: V[i] holds sigma[i] times the i-th vector, y[i] is divided by sigma[i] first
void launch_combine(hipStream_t st, const double *V, int64_t ld, const double *y, int m,
                    double *x, int64_t n, const int *count = nullptr, bool accumulate = false,
                    const double *sigma = nullptr, int64_t pitch = 0);
// x += w
void launch_add(hipStream_t st, double *x, const double *w, int64_t n);
// out = a - b; part[b] = sum over block b of out[k]^2 (out may be b)
void launch_sub_nrm2(hipStream_t st, const double *a, const double *b, double *out, int64_t n,
                     double *part, int *nparts);

// Restarted solver with residual monitor: cycle start, Givens update of
// column j (R: (m+1) x m rotated Hessenberg, cs/sn: m, gv: m+1), cycle end.
void launch_ext_begin(hipStream_t st, const double *beta, ExtState *state, double *gv, int m, bool first,
                      double rtol);
void launch_givens(hipStream_t st, const double *H, int m, int j, double *R, double *cs, double *sn,
                   double *gv, ExtState *state);
void launch_ext_solve(hipStream_t st, const double *R, const double *gv, int m, const ExtState *state,
                      double *y);

// Tridiagonal preconditioner made of independent lines of length L.
void launch_tdp_factor(hipStream_t st, const double *lo, const double *di, const double *up,
                       double *dd, double *ud, int64_t nlines, int64_t L);
void launch_tdp_apply(hipStream_t st, const double *lo, const double *dd, const double *ud,
                      const double *sol, double *out, int64_t nlines, int64_t L);
// The same solve of sol / nrm, nrm = sqrt(sum of part[0 .. np)), which also stores that normalised vector to vout
// (pitch != 0: as a vector of a chunk-interleaved basis) and nrm to *nout: k_scale_from and the solve in one
// pass.  tdp_scaled_supported: whether the 16-byte-lane kernel that offers it serves these arrays.
struct TdpScale {
  const double *part = nullptr;
  int np = 0;
  double *vout = nullptr;
  int64_t pitch = 0;
  double *nout = nullptr;
};
bool tdp_scaled_supported(const double *lo, const double *dd, const double *ud, const double *sol, const double *out,
                          const double *vout, int64_t nlines, int64_t L);
void launch_tdp_apply_scaled(hipStream_t st, const double *lo, const double *dd, const double *ud, const double *sol,
                             double *out, int64_t nlines, int64_t L, const TdpScale &sc);
// lo/di/up[k] = A(k,k-1), A(k,k), A(k,k+1) of a CSR matrix (0 where absent)
void launch_tdp_extract_csr(hipStream_t st, const int64_t *rowptr, const int32_t *col,
                            const double *val, int64_t n, double *lo, double *di, double *up);

void launch_fill(hipStream_t st, double *x, double v, int64_t n);

// ---- one-sweep Arnoldi step (5-band grid operator) ----------------------------
// The vectors these kernels produce stay un-normalised in the basis array V (slot k
// holds sigma_k v_k); sigma lives in a small device array next to the coefficients.
// Workgroups (= partial sums per quantity) of a fused sweep over a rows x cols grid.
int fused_parts(int rows, int cols);
// First step of a cycle that the 512-thread variant of the sweep serves (process-wide; < 0: the default).
void set_split_from(int first_step);
int get_split_from();
int get_split_upto();  // last step served by it
// Step nv of a cycle: with the coefficients coef[0..nv] (already divided by sigma_k)
// and coef[kFusedNormSlot] = sigma_nv, store the new vector
// (A V_nv) / sigma_nv - sum_k coef[k] V_k in slot nv + 1 and the 2 nv + 5 partial sums
// (see k_arnoldi_fused) in part.
// has_halo: the grid rows in front of the first and behind the last row exist in the
// halos of slots nv and nv + 1 (a neighbouring rank's rows, zeros beyond the mesh)
// instead of being treated as zero.
// gen != nullptr: interior band entries are evaluated from the recipe instead of being read.
// which: all bands of rows in one launch, or only the bands that do not touch the rows in front
// of / behind the local grid (they need no neighbour's row of the new vector and can run while
// those rows are still travelling), or only the first and the last band (everything if there
// are fewer than three bands).  The partial sums of an interior and an edge launch together
// fill the array of an all-bands launch.
enum { kSweepAll = 0, kSweepInterior = 1, kSweepEdges = 2 };
// A sweep may evaluate its own coefficients (what launch_fused_coeffs does as a launch of its own) in its prologue,
// from the totals of the previous sweep: every workgroup evaluates the same recurrence on the same totals (same
// bits everywhere), the first one stores G, H, coef and sigma.  One dependent launch per Arnoldi step less.
// tot == nullptr: the coefficients are read from `coef` as a separate launch left them.
struct CoefIn {
  const double *tot = nullptr;
  double *G = nullptr;
  int gld = 0;
  double *H = nullptr;
  int m = 0;
  double *sigma = nullptr;
};
// sm_slots > 0: the basis is strip-major with the rows of sm_slots vectors interleaved (V: vector 0, ld = kBlock)
// and gen's velocity snapshots are strip-major single vectors (StripMap).
void launch_arnoldi_fused(hipStream_t st, int nv, int rows, int cols, const Bands &b, const double *V, int64_t ld,
                          const double *coef, double *part, int *nparts, bool has_halo,
                          const BandGen *gen = nullptr, int which = kSweepAll, int sm_slots = 0,
                          const CoefIn *cin = nullptr);
// The strip-major layout of a vector on a rows x cols grid (slots = 1), or of a basis of `slots` vectors with
// interleaved rows, and the size of one vector's share of it in doubles.
StripMap strip_map(int rows, int cols, int slots = 1);
inline int64_t strip_major_size(int rows, int cols) {
  const StripMap m = strip_map(rows, cols);
  return (int64_t)m.nstrips * m.SB;
}
// dst (strip-major single vector) = src (grid order; src_has_halo_rows: with a grid row in front and behind, which
// are copied too)
void launch_to_strips(hipStream_t st, const double *src, double *dst, int rows, int cols, bool src_has_halo_rows);
// src / sqrt(sum(part)) in one pass, stored in grid order (grid_dst) and as vector 0 of a strip-major basis of
// `slots` interleaved vectors (strip_dst); *out = the norm
void launch_scale_to_strips(hipStream_t st, const double *src, double *grid_dst, double *strip_dst, const double *part,
                            int nparts, int rows, int cols, int slots, double *out);
// launch_combine / launch_unscale for the vectors of a strip-major basis of `slots` interleaved vectors (V: vector
// 0; src: the vector itself); x and dst are in grid order
void launch_combine_strips(hipStream_t st, const double *V, const double *y, int m, double *x, int rows, int cols,
                           int slots, const int *count, bool accumulate, const double *sigma);
void launch_unscale_strips(hipStream_t st, const double *src, const double *sigma, int k, double *dst, int rows, int cols,
                           int slots);
bool fused_bands_splittable(int nv, int rows, int cols);
// Whether the sweep of step nv delivers <t, t> as its last partial sum (2 nv + 5 sums instead of 2 nv + 4).
bool fused_sweep_has_tt(int nv);
// The update alone on the cnt unknowns from k0 on (n: unknowns of the local vector):
// the new vector in slot nv + 1 and its squared norm in part.
// map != nullptr: the basis vectors are strip-major.
// skip != nullptr (here and in the two launchers below): a device flag; the kernel returns at once
// when it is non-zero (the norm the closing step exists for was obtained from the last sweep's
// sums, see launch_fused_coeffs).
void launch_arnoldi_close(hipStream_t st, int nv, const Bands &b, const double *V, int64_t ld, const double *coef,
                          int64_t n, int cols, int64_t k0, int64_t cnt, bool has_halo, double *part, int *nparts,
                          const StripMap *map = nullptr, const int *skip = nullptr);
// tot[q] = sum_i part[q * np + i], one workgroup per quantity
void launch_reduce_parts_wide(hipStream_t st, const double *part, int nq, int np, double *tot,
                              const int *skip = nullptr);
// Coefficients of step i from the totals of the previous sweep (boot: of the operator
// kernel applied to the stored, normalised v_i); see k_fused_coeffs.
// norm_flag != nullptr (step m - 1, the one the closing kernel serves): also the norm of the vector
// that step would produce, ||A v_i - sum_k h_k v_k||^2 = ||A v_i||^2 - sum_k h_k^2 (the identity
// modified Gram-Schmidt satisfies link by link), from the previous sweep's <t, t>.  Where the
// difference keeps at least 1/64 of ||A v_i||^2 the kernel stores its root in sigma[i + 1] and
// H[i + 1][i] and sets *norm_flag = 1 (the closing kernel and what follows it then return at once);
// otherwise (cancellation, or anything that is not finite) it sets *norm_flag = 0 and the closing
// kernel forms the vector and takes its norm explicitly.
void launch_fused_coeffs(hipStream_t st, const double *tot, int i, bool boot, double *G, int gld, double *H,
                         int m, double *coef, double *sigma, int *norm_flag = nullptr);
// sigma[m] = sqrt(tot[0]), H[m][m-1] = sigma[m]
void launch_fused_finish(hipStream_t st, const double *tot, int m, double *H, double *sigma,
                         const int *skip = nullptr);
// dst = src / sigma[k] (sigma == nullptr: dst = src); pitch != 0: src is a vector of a chunk-interleaved basis
void launch_unscale(hipStream_t st, const double *src, const double *sigma, int k, double *dst, int64_t n,
                    int64_t pitch = 0);

// ---- implicit-Euler assembly ------------------------------------------------
struct IeGeom {
  int cx, cy;      // interior cells of the whole mesh
  int row0, rows;  // padded rows [row0, row0 + rows) owned by this rank
  double dx, dy;
  double diffusion, reynolds, prandtl, eckert, inv_rp;
};
struct IeBc {
  double weight[12];  // [field*4 + side]
};

void launch_pack_wrap(hipStream_t st, const IeGeom &g, const double *T, const double *u,
                      const double *v, double *wrap);
void launch_bc_fill(hipStream_t st, const IeGeom &g, const IeBc &bc, double *T, double *u,
                    double *v, const double *bcvals, const double *wrap);
void launch_offdiag(hipStream_t st, const IeGeom &g, const double *u, const double *v, double *bw,
                    double *bs, double *bc, double *bn, double *be);
void launch_assemble(hipStream_t st, const IeGeom &g, const IeBc &bc, double dt, const double *T,
                     const double *u, const double *v, double *bw, double *bs, double *bcn,
                     double *bn, double *be, double *rhs);
// T += delta on interior cells; part (2 * nparts doubles) = per workgroup the largest
// |delta| that is not NaN and the last position of a NaN
void launch_update_max(hipStream_t st, const IeGeom &g, double *T, const double *delta,
                       double *part, int *nparts);
// The reference's max_delta (std::max fold in cell order: the maximum behind the last NaN
// cell, NaN if the last cell is NaN) from those partial results; scal = {result, last NaN
// position, largest |delta| that is not NaN}; phases: bit 0 collapse the partial results,
// bit 1 the maximum behind the last NaN position, bit 2 the NaN case (see k_max_fold)
void launch_max_fold(hipStream_t st, const IeGeom &g, const double *delta, const double *part, int nparts,
                     double *scal, int phases);
// The same fold for the approximate-factorisation step from k_af_sweep_x2's summaries.
void launch_af_max_final(hipStream_t st, const IeGeom &g, const double *part, int nparts, double *out);
// Approximate factorisation: y-direction line systems (coefficients and right-hand
// side, padded layout), and the x-direction sweep with the state update.
void launch_af_rows_y(hipStream_t st, const IeGeom &g, const IeBc &bc, double dt, const double *T,
                      const double *u, const double *v, double *lo, double *di, double *up, double *rhs);
void launch_af_sweep_x(hipStream_t st, const IeGeom &g, const IeBc &bc, double dt, double *T, const double *u,
                       const double *rhs, double *y, double *ud, double *part, int *nparts);

}  // namespace kry

#endif  // KRY_INTERNAL_H_
