// HIP kernels (gfx950) for the implicit-Euler / GMRES path.
//
// The whole path is HBM-bound fp64 streaming: no MFMA.  Design rules used here
//  * 256-thread workgroups (4 wave64), at most 2048 of them, grid-stride loops;
//  * 16-byte (double2) loads and stores wherever alignment allows;
//  * reductions: per-lane accumulators -> wave64 shuffle -> LDS -> one partial
//    per workgroup in a scratch array.  The consumer kernel of a scalar sums
//    the partials itself, in a fixed order that is the same in every
//    workgroup, so results are deterministic, no atomics or fences are needed
//    and the host never reads a scalar inside a GMRES cycle;
//  * this file is compiled with -ffp-contract=off: element-wise arithmetic
//    (stencil row sums, w -= h v, x += y v, assembly) is rounded exactly like
//    the reference's scalar C++; fma() is used explicitly only inside
//    reductions, whose summation order differs from the reference anyway.
#include <stdlib.h>

#include <cstdlib>

#include "kry_device.h"

namespace kry {

// Device flag that turns the iteration kernels launched by this thread into
// no-ops once it is non-zero (see LaunchPredicate in kry_internal.h).
static thread_local const int *t_stop = nullptr;

LaunchPredicate::LaunchPredicate(const int *stop) : saved_(t_stop) { t_stop = stop; }
LaunchPredicate::~LaunchPredicate() { t_stop = saved_; }
const int *launch_predicate_current() { return t_stop; }

// A failure met while preparing a launch (launchers return nothing): kept per thread
// until the entry point that queued the work asks for it.
static thread_local hipError_t t_failure = hipSuccess;
static thread_local const char *t_failure_what = "";

void note_launch_failure(hipError_t e, const char *what) {
  if (t_failure == hipSuccess) {
    t_failure = e;
    t_failure_what = what;
  }
}

hipError_t take_launch_failure(const char **what) {
  const hipError_t e = t_failure;
  if (what) *what = t_failure_what;
  t_failure = hipSuccess;
  t_failure_what = "";
  return e;
}

int grid_for(int64_t items) {
  int64_t b = (items + kBlock - 1) / kBlock;
  if (b < 1) b = 1;
  if (b > kMaxGrid) b = kMaxGrid;
  return (int)b;
}

// ---------------------------------------------------------------------------
// BLAS-1 shaped kernels
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_fill(double *__restrict__ x, double v, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    x[i] = v;
}

// tot[q] = sum_i part[q * np + i]: collapses the per-workgroup partial sums to
// one value per quantity so that ranks can all-reduce a handful of doubles.
__global__ __launch_bounds__(kBlock) void k_reduce_parts(const double *__restrict__ part, int nq, int np,
                                                          double *__restrict__ tot, const int *__restrict__ stop) {
  if (stopped(stop)) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int q = wv; q < nq; q += kBlock / 64) {
    double a = 0.0;
    for (int i = lane; i < np; i += 64) a += part[(size_t)q * np + i];
    a = wave_sum(a);
    if (lane == 0) tot[q] = a;
  }
}

__global__ __launch_bounds__(kBlock) void k_nrm2_part(const double *__restrict__ x, int64_t n,
                                                       double *__restrict__ part) {
  __shared__ double s[kBlock / 64];
  const int64_t n2 = n >> 1;
  const d2 *x2 = reinterpret_cast<const d2 *>(x);
  double a0 = 0.0, a1 = 0.0;
  constexpr int U = 4;
  const int64_t tile = (int64_t)kBlock * U;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n2; base += (int64_t)gridDim.x * tile) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = base + u * kBlock + threadIdx.x;
      if (i < n2) {
        const d2 v = ld_nt(&x2[i]);
        a0 = fma(v.x, v.x, a0);
        a1 = fma(v.y, v.y, a1);
      }
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) a0 = fma(x[n - 1], x[n - 1], a0);
  const double t = block_sum(a0 + a1, s);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// dst = src / sqrt(sum(part)); true division, as normalize() in include/matrix.hpp:54-60
__global__ __launch_bounds__(kBlock) void k_scale_from(const double *__restrict__ src,
                                                        double *__restrict__ dst,
                                                        const double *__restrict__ part, int np,
                                                        int64_t n, double *__restrict__ out,
                                                        const int *__restrict__ stop, int64_t pitch,
                                                        double *__restrict__ dst2) {
  if (stopped(stop)) return;
  __shared__ double s[kBlock / 64];
  const double nrm = sqrt(sum_parts(part, np, s));
  if (blockIdx.x == 0 && threadIdx.x == 0 && out) *out = nrm;
  const int64_t n2 = n >> 1;
  const d2 *s2 = reinterpret_cast<const d2 *>(src);
  d2 *o2 = reinterpret_cast<d2 *>(dst);
  constexpr int U = 4;
  const int64_t tile = (int64_t)kBlock * U;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n2; base += (int64_t)gridDim.x * tile) {
    d2 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = base + u * kBlock + threadIdx.x;
      if (i < n2) v[u] = ld_nt(&s2[i]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = base + u * kBlock + threadIdx.x;
      if (i < n2) {
        d2 r;
        r.x = v[u].x / nrm;
        r.y = v[u].y / nrm;
        st_nt(r, &o2[il2(i, pitch)]);
        if (dst2 != nullptr) st_nt(r, reinterpret_cast<d2 *>(dst2) + i);
      }
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    dst[il1(n - 1, pitch)] = src[n - 1] / nrm;
    if (dst2 != nullptr) dst2[n - 1] = src[n - 1] / nrm;
  }
}

// ---------------------------------------------------------------------------
// Blocked modified Gram-Schmidt pass (include/solver.hpp:196-211).
//
// The reference orthogonalises w against v_0, v_1, ... one vector at a time:
//   h_i = <w, v_i>;  w -= h_i v_i        (dot taken with the UPDATED w).
// Every link is one sweep over w.  This kernel does up to kMgsBlock links per
// sweep.  Basis vectors are taken in fixed groups of K (group g = vectors
// g K .. g K + K - 1).  For a group c_0..c_{KC-1} the kernel accumulates, in one
// pass over the already-updated w,
//     d_k  = <w, c_k>            and      g_a = <c_a, c_{KC-1}>   (a < KC-1),
// i.e. the Gram entries of the group's LAST vector only: the Gram entries of
// the earlier vectors were produced the same way in the Arnoldi step in which
// each of them was the newest vector, and are kept in the small matrix G.  The
// NEXT kernel (its prologue, identically in every workgroup) recovers the
// modified-Gram-Schmidt coefficients exactly as the recurrence defines them,
//     h_0 = d_0,  h_1 = d_1 - h_0 g_01,  h_2 = d_2 - h_0 g_02 - h_1 g_12, ...
// (because <w - h_0 c_0, c_1> = <w, c_1> - h_0 <c_0, c_1>), stores the new Gram
// column in G, then applies
//     w -= h_0 c_0;  w -= h_1 c_1; ...   element-wise in the reference's order
// (multiply, then subtract) while taking the dots of the following group.
// KC = 0 with NORM: the last pass of an Arnoldi step, accumulates <w, w>.
//
// Partial sums: part[q * gridDim.x + workgroup], q < 2 KC - 1 (dots first, then
// the Gram entries of the last vector), or q = 0 for NORM.
// G: row-major, leading dimension gld; the previous group starts at row/column gp.
// ---------------------------------------------------------------------------
template <int KP, int KC, bool NORM, int U>
__global__ __launch_bounds__(kBlock) void k_mgs_block(double *__restrict__ w,
                                                       const double *__restrict__ pbase,
                                                       const double *__restrict__ cbase, int64_t ld,
                                                       const double *__restrict__ part_in, int np_in,
                                                       double *__restrict__ part_out,
                                                       double *__restrict__ hout, int hstride,
                                                       int64_t n, double *__restrict__ G, int gld, int gp,
                                                       const int *__restrict__ stop, int64_t pitch) {
  if (stopped(stop)) return;
  constexpr int NQI = nred(KP);
  constexpr int NQO = NORM ? 1 : nred(KC);
  __shared__ double tot[NQI];
  __shared__ double gs[KP * KP];
  __shared__ double red[NQO * (kBlock / 64)];
  // Gram entries of the previous group's earlier vectors (upper triangle, a < k < KP-1)
  for (int t = threadIdx.x; t < KP * KP; t += kBlock) {
    const int a = t / KP, k = t % KP;
    gs[t] = (a < k && k < KP - 1) ? G[(size_t)(gp + a) * gld + (gp + k)] : 0.0;
  }
  sum_parts_multi<NQI>(part_in, np_in, tot);  // ends with a barrier
  // modified Gram-Schmidt coefficients of the previous group
  double h[KP];
#pragma unroll
  for (int k = 0; k < KP; k++) {
    double v = tot[k];
#pragma unroll
    for (int a = 0; a < k; a++) v = v - h[a] * (k == KP - 1 ? tot[KP + a] : gs[a * KP + k]);
    h[k] = v;
  }
  if (blockIdx.x == 0 && threadIdx.x < KP) {
#pragma unroll
    for (int k = 0; k < KP; k++)
      if ((int)threadIdx.x == k) {
        hout[(size_t)k * hstride] = h[k];
        if (k < KP - 1) G[(size_t)(gp + k) * gld + (gp + KP - 1)] = tot[KP + k];  // new Gram column
      }
  }
  const int64_t n2 = n >> 1;
  d2 *w2 = reinterpret_cast<d2 *>(w);
  double acc[NQO];
#pragma unroll
  for (int q = 0; q < NQO; q++) acc[q] = 0.0;
  const int64_t tile = (int64_t)kBlock * U;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n2; base += (int64_t)gridDim.x * tile) {
    d2 wv[U], pv[KP][U], cv[KC > 0 ? KC : 1][U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = base + u * kBlock + threadIdx.x;
      if (i < n2) {
        wv[u] = ld_nt(&w2[i]);
        const int64_t iv = il2(i, pitch);
#pragma unroll
        for (int k = 0; k < KP; k++) pv[k][u] = ld_nt(reinterpret_cast<const d2 *>(pbase + (int64_t)k * ld) + iv);
#pragma unroll
        for (int k = 0; k < KC; k++) cv[k][u] = ld_nt(reinterpret_cast<const d2 *>(cbase + (int64_t)k * ld) + iv);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = base + u * kBlock + threadIdx.x;
      if (i < n2) {
        d2 r = wv[u];
#pragma unroll
        for (int k = 0; k < KP; k++) r = r - h[k] * pv[k][u];
        st_nt(r, &w2[i]);
        if (NORM) {
          acc[0] = fma(r.x, r.x, acc[0]);
          acc[0] = fma(r.y, r.y, acc[0]);
        } else {
#pragma unroll
          for (int a = 0; a < KC; a++) {
            acc[a] = fma(r.x, cv[a][u].x, acc[a]);
            acc[a] = fma(r.y, cv[a][u].y, acc[a]);
            if (a < KC - 1) {
              acc[KC + a] = fma(cv[a][u].x, cv[KC - 1][u].x, acc[KC + a]);
              acc[KC + a] = fma(cv[a][u].y, cv[KC - 1][u].y, acc[KC + a]);
            }
          }
        }
      }
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {  // odd tail element
    const int64_t e = n - 1, ev = il1(e, pitch);
    double r = w[e];
#pragma unroll
    for (int k = 0; k < KP; k++) r = r - h[k] * pbase[(int64_t)k * ld + ev];
    w[e] = r;
    if (NORM) {
      acc[0] = fma(r, r, acc[0]);
    } else {
#pragma unroll
      for (int a = 0; a < KC; a++) {
        acc[a] = fma(r, cbase[(int64_t)a * ld + ev], acc[a]);
        if (a < KC - 1)
          acc[KC + a] = fma(cbase[(int64_t)a * ld + ev], cbase[(int64_t)(KC - 1) * ld + ev], acc[KC + a]);
      }
    }
  }
  reduce_store_multi<NQO>(acc, red, part_out);
}

// x = sum_i y_i v_i, accumulated per element in the order i = 0..m-1 starting
// from 0.0 with a separate multiply and add (include/solver.hpp:153-159).
// count (optional, device): only the first min(m, *count) vectors take part;
// accumulate: start from the current x instead of 0.0.
__global__ __launch_bounds__(kBlock) void k_combine(const double *__restrict__ V, int64_t ld,
                                                     const double *__restrict__ y, int m,
                                                     double *__restrict__ x, int64_t n,
                                                     const int *__restrict__ count, int accumulate,
                                                     const double *__restrict__ sigma, int64_t pitch) {
  __shared__ double ys[kMaxM];
  if (count != nullptr) {
    const int c = *count;
    if (c < m) m = c;
  }
  // sigma: slot i of V holds sigma[i] times the i-th basis vector
  for (int i = threadIdx.x; i < m; i += kBlock) ys[i] = sigma != nullptr ? y[i] / sigma[i] : y[i];
  __syncthreads();
  const int64_t n2 = n >> 1;
  d2 *x2 = reinterpret_cast<d2 *>(x);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
    d2 acc = {0.0, 0.0};
    if (accumulate) acc = x2[i];
    const int64_t iv = il2(i, pitch);
#pragma unroll 8
    for (int k = 0; k < m; k++) {
      const d2 v = ld_nt(reinterpret_cast<const d2 *>(V + (int64_t)k * ld) + iv);
      acc = acc + ys[k] * v;
    }
    st_nt(acc, &x2[i]);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    double acc = accumulate ? x[n - 1] : 0.0;
    for (int k = 0; k < m; k++) acc = acc + ys[k] * V[(int64_t)k * ld + il1(n - 1, pitch)];
    x[n - 1] = acc;
  }
}

// x += w
__global__ __launch_bounds__(kBlock) void k_add(double *__restrict__ x, const double *__restrict__ w, int64_t n) {
  const int64_t n2 = n >> 1;
  d2 *x2 = reinterpret_cast<d2 *>(x);
  const d2 *w2 = reinterpret_cast<const d2 *>(w);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock)
    x2[i] = x2[i] + ld_nt(&w2[i]);
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) x[n - 1] = x[n - 1] + w[n - 1];
}

// out = a - b and the per-workgroup partial sums of out^2 (restart residual)
__global__ __launch_bounds__(kBlock) void k_sub_nrm2(const double *__restrict__ a, const double *b, double *out,
                                                      int64_t n, double *__restrict__ part) {
  __shared__ double s[kBlock / 64];
  const int64_t n2 = n >> 1;
  const d2 *a2 = reinterpret_cast<const d2 *>(a);
  const d2 *b2 = reinterpret_cast<const d2 *>(b);
  d2 *o2 = reinterpret_cast<d2 *>(out);
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n2; i += (int64_t)gridDim.x * kBlock) {
    const d2 r = a2[i] - b2[i];
    o2[i] = r;
    acc = fma(r.x, r.x, acc);
    acc = fma(r.y, r.y, acc);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double r = a[n - 1] - b[n - 1];
    out[n - 1] = r;
    acc = fma(r, r, acc);
  }
  const double t = block_sum(acc, s);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------
// Restarted GMRES with a residual monitor (no counterpart in the reference,
// whose cycle always runs num_iters steps and solves the normal equations):
// the Hessenberg column of every Arnoldi step is reduced by Givens rotations
// on the device, the rotated right-hand side gives the residual norm, and once
// it meets the tolerance a device flag turns the remaining iteration kernels
// of the cycle into no-ops - the host looks at the state once per cycle.
// ---------------------------------------------------------------------------
// Start of a cycle: g = (beta, 0, ..., 0); the first cycle fixes the absolute
// tolerance rtol * ||b||.
__global__ __launch_bounds__(64) void k_ext_begin(const double *__restrict__ beta, ExtState *__restrict__ st,
                                                   double *__restrict__ gv, int m, int first, double rtol) {
  for (int i = 1 + threadIdx.x; i <= m; i += 64) gv[i] = 0.0;
  if (threadIdx.x == 0) {
    const double b = *beta;
    if (first) {
      st->beta0 = b;
      st->tol_abs = rtol * b;
    }
    gv[0] = b;
    st->res = b;
    st->iters = 0;
    const bool done = b <= st->tol_abs;
    st->converged = done ? 1 : 0;
    st->stop = (done || !(b == b)) ? 1 : 0;
  }
}

// Step j: apply the j earlier rotations to column j of H, form the new
// rotation, update g; |g_{j+1}| is the residual norm of the best iterate.
__global__ __launch_bounds__(64) void k_givens(const double *__restrict__ H, int m, int j,
                                                double *__restrict__ R, double *__restrict__ cs,
                                                double *__restrict__ sn, double *__restrict__ gv,
                                                ExtState *__restrict__ st) {
  if (st->stop != 0) return;
  __shared__ double r[kMaxM + 1], c[kMaxM], s[kMaxM];
  for (int i = threadIdx.x; i <= j + 1; i += 64) r[i] = H[(size_t)i * m + j];
  for (int i = threadIdx.x; i < j; i += 64) {
    c[i] = cs[i];
    s[i] = sn[i];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 0; i < j; i++) {
      const double t = c[i] * r[i] + s[i] * r[i + 1];
      r[i + 1] = c[i] * r[i + 1] - s[i] * r[i];
      r[i] = t;
    }
    const double a = r[j], b = r[j + 1];
    const double d = sqrt(a * a + b * b);
    double cc = 1.0, ss = 0.0;
    if (d != 0.0) {
      cc = a / d;
      ss = b / d;
    }
    r[j] = cc * a + ss * b;
    r[j + 1] = 0.0;
    const double gj = gv[j];
    const double gn = -(ss * gj);
    gv[j] = cc * gj;
    gv[j + 1] = gn;
    cs[j] = cc;
    sn[j] = ss;
    const double res = fabs(gn);
    st->res = res;
    st->iters = j + 1;
    if (res <= st->tol_abs) {
      st->converged = 1;
      st->stop = 1;
    } else if (!(res == res)) {
      st->stop = 1;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i <= j + 1; i += 64) R[(size_t)i * m + j] = r[i];
}

// End of a cycle: back substitution R y = g over the k = st->iters columns
// that were completed; y[i >= k] = 0.
__global__ __launch_bounds__(64) void k_ext_solve(const double *__restrict__ R, const double *__restrict__ gv,
                                                   int m, const ExtState *__restrict__ st,
                                                   double *__restrict__ y) {
  __shared__ double ys[kMaxM];
  const int k = st->iters;
  for (int i = threadIdx.x; i < m; i += 64) ys[i] = 0.0;
  __syncthreads();
  for (int i = k - 1; i >= 0; i--) {
    double a = 0.0;
    for (int jj = i + 1 + threadIdx.x; jj < k; jj += 64) a = fma(R[(size_t)i * m + jj], ys[jj], a);
    a = wave_sum(a);
    if (threadIdx.x == 0) ys[i] = (gv[i] - a) / R[(size_t)i * m + i];
    __syncthreads();
  }
  for (int i = threadIdx.x; i < m; i += 64) y[i] = ys[i];
}

// ---------------------------------------------------------------------------
// Operator application (include/matrix.hpp:23-33 for a 5-band / CSR matrix),
// fused with the first Gram-Schmidt group: besides y = A x the kernels
// accumulate <y, c_k> for KC basis vectors c_k = cbase + k*ld and the Gram
// entries <c_a, c_{KC-1}> of the last one (partial sums laid out as in k_mgs_block).
// ---------------------------------------------------------------------------
template <int KC>
__device__ __forceinline__ void accum_scalar(double r, const double *__restrict__ cbase, int64_t ld,
                                             int64_t k, double (&acc)[KC > 0 ? nred(KC) : 1], int64_t pitch) {
  double c[KC > 0 ? KC : 1];
  const int64_t kv = il1(k, pitch);
#pragma unroll
  for (int a = 0; a < KC; a++) c[a] = cbase[(int64_t)a * ld + kv];
  if constexpr (KC > 0) {
#pragma unroll
    for (int a = 0; a < KC; a++) {
      acc[a] = fma(r, c[a], acc[a]);
      if (a < KC - 1) acc[KC + a] = fma(c[a], c[KC - 1], acc[KC + a]);
    }
  }
}

// Row sum in ascending column order starting from +0.0:
//   ((((0 + w x[k-P]) + s x[k-1]) + c x[k]) + n x[k+1]) + e x[k+P]
template <int KC, bool HALO>
__global__ __launch_bounds__(kBlock) void k_spmv_stencil(Bands b, int64_t n, int P,
                                                          const double *__restrict__ x,
                                                          double *__restrict__ y,
                                                          const double *__restrict__ cbase, int64_t ld,
                                                          double *__restrict__ part, const int *__restrict__ stop,
                                                          int64_t pitch) {
  if (stopped(stop)) return;
  constexpr int NQ = KC > 0 ? nred(KC) : 1;
  __shared__ double red[NQ * (kBlock / 64)];
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = 0.0;
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < n; k += (int64_t)gridDim.x * kBlock) {
    const double xw = (HALO || k >= P) ? x[k - P] : 0.0;
    const double xs = (HALO || k >= 1) ? x[k - 1] : 0.0;
    const double xc = x[k];
    const double xn = (HALO || k + 1 < n) ? x[k + 1] : 0.0;
    const double xe = (HALO || k + P < n) ? x[k + P] : 0.0;
    double r = 0.0;
    r = r + b.w[k] * xw;
    r = r + b.s[k] * xs;
    r = r + b.c[k] * xc;
    r = r + b.n[k] * xn;
    r = r + b.e[k] * xe;
    y[k] = r;
    if (KC > 0) accum_scalar<KC>(r, cbase, ld, k, acc, pitch);
  }
  if (KC > 0) reduce_store_multi<NQ>(acc, red, part);
}

// HALO: x is the local slab of a row-block decomposition with one valid row of
// P entries in front of x[0] and behind x[n-1] (the neighbours' boundary rows,
// or zeros at the ends of the mesh), so no bounds checks are needed.
//
// Vectorised form for even P and even n: every lane owns two consecutive
// unknowns; band, centre and +-P accesses are aligned 16-byte loads.  Bands,
// partners and the result are streamed non-temporally; x keeps the default
// policy because its rows are reused by the neighbouring rows' sums.
// The five band entries of the unknown in mesh row ip, column c of an assembled implicit-Euler operator,
// from the velocities next to it: SecondOrderCentered::Dx_m1 / Dy_m1 / Dy_p1 / Dx_p1
// (include/space_disc.hpp:83-113) with the operations of k_offdiag for interior cells, the rows of ghost
// cells as k_assemble leaves them (include/time_disc.hpp:304-337): the same bits as the stored bands.
__device__ __forceinline__ void gen_band_entries(const BandGen &gen, const SharedDivisor &sdx, const SharedDivisor &sdy,
                                                 double gdiag, int ip, int c, double um, double up, double vm, double vp,
                                                 double (&out)[5]) {
  const bool xin = ip >= 1 && ip <= gen.cx, yin = c >= 1 && c <= gen.cy;
  double iw, is, in, ie;
  band_quotients(-um / 2.0 - gen.kx, -vm / 2.0 - gen.ky, vp / 2.0 - gen.ky, up / 2.0 - gen.kx, sdx, sdy, iw, is, in, ie);
  const bool inner = xin && yin;
  out[0] = inner ? iw : ((yin && ip == gen.cx + 1) ? 1.0 : 0.0);
  out[1] = inner ? is : ((xin && c == gen.cy + 1) ? 1.0 : 0.0);
  out[3] = inner ? in : ((xin && c == 0) ? 1.0 : 0.0);
  out[4] = inner ? ie : ((yin && ip == 0) ? 1.0 : 0.0);
  out[2] = inner ? gdiag : (xin ? (c == 0 ? gen.dg_bottom : gen.dg_top) : (yin ? gen.dg_left : 1.0));
}

// GEN: the band entries are not read but evaluated from the recipe `gen` (two velocity snapshots instead of
// five bands: three streamed arrays less per apply).
template <int KC, bool HALO, int U, bool GEN = false>
__global__ __launch_bounds__(kBlock) void k_spmv_stencil_v2(Bands b, int64_t n, int P,
                                                             const double *__restrict__ x,
                                                             double *__restrict__ y,
                                                             const double *__restrict__ cbase, int64_t ld,
                                                             double *__restrict__ part, const int *__restrict__ stop,
                                                             BandGen gen, int64_t pitch) {
  if (stopped(stop)) return;
  constexpr int NQ = KC > 0 ? nred(KC) : 1;
  __shared__ double red[NQ * (kBlock / 64)];
  const int64_t n2 = n >> 1;
  const int64_t P2 = P >> 1;
  const double gdiag = GEN ? *gen.cdiag : 0.0;
  const SharedDivisor sdx = shared_divisor(gen.dx, gen.rdx, gen.div_ok != 0), sdy = shared_divisor(gen.dy, gen.rdy, gen.div_ok != 0);
  const d2 *gu2 = reinterpret_cast<const d2 *>(gen.u);
  const d2 *gv2 = reinterpret_cast<const d2 *>(gen.v);
  const d2 *x2 = reinterpret_cast<const d2 *>(x);
  const d2 *bw = reinterpret_cast<const d2 *>(b.w);
  const d2 *bs = reinterpret_cast<const d2 *>(b.s);
  const d2 *bc = reinterpret_cast<const d2 *>(b.c);
  const d2 *bn = reinterpret_cast<const d2 *>(b.n);
  const d2 *be = reinterpret_cast<const d2 *>(b.e);
  d2 *y2 = reinterpret_cast<d2 *>(y);
  const d2 zero = {0.0, 0.0};
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = 0.0;
  const int64_t tile = (int64_t)kBlock * U;
  for (int64_t base = (int64_t)blockIdx.x * tile; base < n2; base += (int64_t)gridDim.x * tile) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const int64_t i = base + u * kBlock + threadIdx.x;
      if (i < n2) {
        const int64_t k = 2 * i;
        const d2 xc = x2[i];
        const d2 xw = (HALO || i >= P2) ? x2[i - P2] : zero;
        const d2 xe = (HALO || i + P2 < n2) ? x2[i + P2] : zero;
        const double xm1 = (HALO || k >= 1) ? x[k - 1] : 0.0;
        const double xp2 = (HALO || k + 2 < n) ? x[k + 2] : 0.0;
        d2 cw, cs, cc, cn, ce;
        if (GEN) {
          // the velocity snapshots carry a grid row in front and behind (like the fields), so no bounds checks
          const d2 um = ld_nt(&gu2[i - P2]), up = ld_nt(&gu2[i + P2]);
          const d2 vc = gv2[i];
          const double vm1 = gen.v[k - 1], vp2 = gen.v[k + 2];
          const int row = (int)((uint32_t)k / (uint32_t)P), col0 = (int)k - row * P;
          double e0[5], e1[5];
          gen_band_entries(gen, sdx, sdy, gdiag, gen.row0 + row, col0, um.x, up.x, vm1, vc.y, e0);
          gen_band_entries(gen, sdx, sdy, gdiag, gen.row0 + row, col0 + 1, um.y, up.y, vc.x, vp2, e1);
          cw = {e0[0], e1[0]};
          cs = {e0[1], e1[1]};
          cc = {e0[2], e1[2]};
          cn = {e0[3], e1[3]};
          ce = {e0[4], e1[4]};
        } else {
          cw = ld_nt(&bw[i]);
          cs = ld_nt(&bs[i]);
          cc = ld_nt(&bc[i]);
          cn = ld_nt(&bn[i]);
          ce = ld_nt(&be[i]);
        }
        d2 cv[KC > 0 ? KC : 1];
#pragma unroll
        for (int a = 0; a < KC; a++) cv[a] = ld_nt(reinterpret_cast<const d2 *>(cbase + (int64_t)a * ld) + il2(i, pitch));
        double r0 = 0.0, r1 = 0.0;
        r0 = r0 + cw.x * xw.x;
        r1 = r1 + cw.y * xw.y;
        r0 = r0 + cs.x * xm1;
        r1 = r1 + cs.y * xc.x;
        r0 = r0 + cc.x * xc.x;
        r1 = r1 + cc.y * xc.y;
        r0 = r0 + cn.x * xc.y;
        r1 = r1 + cn.y * xp2;
        r0 = r0 + ce.x * xe.x;
        r1 = r1 + ce.y * xe.y;
        d2 r = {r0, r1};
        st_nt(r, &y2[i]);
        if (KC > 0) {
#pragma unroll
          for (int a = 0; a < KC; a++) {
            acc[a] = fma(r0, cv[a].x, acc[a]);
            acc[a] = fma(r1, cv[a].y, acc[a]);
            if (a < KC - 1) {
              acc[KC + a] = fma(cv[a].x, cv[KC - 1].x, acc[KC + a]);
              acc[KC + a] = fma(cv[a].y, cv[KC - 1].y, acc[KC + a]);
            }
          }
        }
      }
    }
  }
  if (KC > 0) reduce_store_multi<NQ>(acc, red, part);
}

// CSR, stored entries in ascending column order.  The row sum of include/matrix.hpp:23-33 is sequential
// (result = 0, then += in ascending column order), so one lane adds up one row -- but a lane that also fetched
// its own row would read val / col at a stride of one row length, a different cache line per lane.  Here a
// workgroup owns kBlock consecutive rows, whose entries form ONE contiguous piece of val / col: all lanes fetch
// that piece side by side (coalesced, streamed once), multiply every entry by its x (a product needs no order)
// and park the products in LDS; then lane r adds up the products of row r one after the other, in the stored
// order: the same operations on the same operands as `sum += val[p] * x[col[p]]`, hence the same bits.
constexpr int kCsrChunk = 2048;  // products parked per round (16 KB of LDS)

template <int KC>
__global__ __launch_bounds__(kBlock) void k_spmv_csr(const int64_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ col,
                                                      const double *__restrict__ val, int64_t n,
                                                      const double *__restrict__ x,
                                                      double *__restrict__ y,
                                                      const double *__restrict__ cbase, int64_t ld,
                                                      double *__restrict__ part, const int *__restrict__ stop,
                                                      int64_t pitch) {
  if (stopped(stop)) return;
  constexpr int NQ = KC > 0 ? nred(KC) : 1;
  __shared__ double red[NQ * (kBlock / 64)];
  __shared__ double prod[kCsrChunk];
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = 0.0;
  for (int64_t r0 = (int64_t)blockIdx.x * kBlock; r0 < n; r0 += (int64_t)gridDim.x * kBlock) {
    const int64_t r = r0 + threadIdx.x;
    const int64_t rend = r0 + kBlock < n ? r0 + kBlock : n;
    const int64_t p0 = rowptr[r0], p1 = rowptr[rend];            // the workgroup's piece of val / col
    const int64_t mb = r < n ? rowptr[r] : p1, me = r < n ? rowptr[r + 1] : p1;  // this lane's row
    double sum = 0.0;
    for (int64_t c0 = p0; c0 < p1; c0 += kCsrChunk) {
      const int64_t c1 = c0 + kCsrChunk < p1 ? c0 + kCsrChunk : p1;
      for (int64_t p = c0 + threadIdx.x; p < c1; p += kBlock)
        prod[p - c0] = __builtin_nontemporal_load(&val[p]) * x[__builtin_nontemporal_load(&col[p])];
      __syncthreads();
      const int64_t a = mb > c0 ? mb : c0, b = me < c1 ? me : c1;
      for (int64_t p = a; p < b; p++) sum = sum + prod[p - c0];
      __syncthreads();
    }
    if (r < n) {
      y[r] = sum;
      if (KC > 0) accum_scalar<KC>(sum, cbase, ld, r, acc, pitch);
    }
  }
  if (KC > 0) reduce_store_multi<NQ>(acc, red, part);
}

// ---------------------------------------------------------------------------
// Reduced (m x m) least-squares solve, include/solver.hpp:142-151 with
// LUSolver::lu_decomp (:58-96) and LUSolver::solve (:34-56).  One workgroup;
// every matrix entry is produced by exactly the reference's sequence of
// multiplies, subtracts and divides; only independent entries are spread over
// the lanes:
//   * the entries of H^T H, and of row k of u / column k of l in elimination step k
//     (each a chain over the earlier steps in ascending order, as in the reference);
//   * the forward substitution, column by column: once partial(j) is final, every
//     later row subtracts l(i,j) partial(j) -- rows receive their terms in ascending
//     j, the order of :42-44;
//   * the back substitution stays one sequential walk: row i subtracts its terms in
//     ASCENDING j (:50-52) and the first of them, u(i,i+1) vars(i+1), is the last
//     value to become known.
// With LDS the three m x m arrays live in LDS (60 KB for m = 50; up to kSmallLdsMaxM),
// otherwise in `work` (3 m^2 + 3 m doubles of global memory).  l is kept transposed
// (lt[i m + j] = l(j,i)): the lanes that own different rows j then read neighbouring
// addresses.
// ---------------------------------------------------------------------------
constexpr int kSmallLdsMaxM = 64;

template <bool LDS>
__global__ __launch_bounds__(kBlock) void k_small_solve(const double *__restrict__ H,
                                                         const double *__restrict__ beta_p, int m,
                                                         double *__restrict__ y,
                                                         double *__restrict__ work) {
  extern __shared__ double small_lds[];
  double *base = LDS ? small_lds : work;
  double *sys = base;
  double *lt = base + (size_t)m * m;
  double *u = base + 2 * (size_t)m * m;
  double *rhs = base + 3 * (size_t)m * m;
  double *partial = rhs + m;
  double *ys = partial + m;
  const int tid = threadIdx.x;
  const double beta = *beta_p;
  for (int e = tid; e < m * m; e += kBlock) {
    const int i = e / m, j = e % m;
    double acc = 0.0;
    for (int k = 0; k < m + 1; k++) acc = acc + H[k * m + i] * H[k * m + j];
    sys[e] = acc;
    lt[e] = 0.0;
    u[e] = 0.0;
  }
  for (int i = tid; i < m; i += kBlock) rhs[i] = beta * H[i];
  __syncthreads();
  for (int k = 0; k < m; k++) {
    // row k of u (the diagonal entry first; l(k,k) = 1)
    for (int j = k + tid; j < m; j += kBlock) {
      double v = sys[k * m + j];
      for (int i = 0; i < k; i++) v = v - lt[i * m + k] * u[i * m + j];
      v = v / 1.0;
      u[k * m + j] = v;
    }
    if (tid == 0) lt[k * m + k] = 1.0;
    __syncthreads();
    const double ukk = u[k * m + k];
    for (int j = k + 1 + tid; j < m; j += kBlock) {
      double v = sys[j * m + k];
      for (int i = 0; i < k; i++) v = v - lt[i * m + j] * u[i * m + k];
      v = v / ukk;
      lt[k * m + j] = v;
    }
    __syncthreads();
  }
  if (m <= 64) {
    if (tid < 64) {
      // forward substitution in the first wavefront: lane i owns row i
      const int i = tid;
      double v = i < m ? rhs[i] : 0.0;
      for (int j = 0; j < m; j++) {
        const double pj = __shfl(v / 1.0, j, 64);  // partial(j) /= l(j,j), l(j,j) = 1
        if (i > j && i < m) v = v - lt[j * m + i] * pj;
      }
      if (i < m) partial[i] = v / 1.0;
    }
  } else if (tid == 0) {
    for (int i = 0; i < m; i++) {
      double v = rhs[i];
      for (int j = 0; j < i; j++) v = v - lt[j * m + i] * partial[j];
      v = v / lt[i * m + i];
      partial[i] = v;
    }
  }
  __syncthreads();
  if (tid == 0) {
    for (int i = m - 1; i >= 0; i--) {
      double v = partial[i];
      for (int j = i + 1; j < m; j++) v = v - u[i * m + j] * ys[j];
      v = v / u[i * m + i];
      ys[i] = v;
      y[i] = v;
    }
  }
}

// ---------------------------------------------------------------------------
// Tridiagonal preconditioner (include/solver.hpp:246-273, :295-314).
// The matrix is a set of independent lines of length L (for the stencil
// operator: one padded grid row each, couplings between rows are exact zeros
// in the reference's numbering), each solved by the Thomas recurrences of the
// reference in the same order.  The pivots do not depend on the right-hand
// side, so they are computed once per assembled matrix.
//
// Device form: one wavefront owns LPW lines.  A chunk of kTdpC = 64 columns of
// those lines is fetched with lane = column (each load instruction moves one
// line's 512 contiguous bytes), parked in LDS, and then lane l < LPW walks
// line l through the chunk out of LDS (row pitch 66 doubles: conflict-free in
// both directions, two elements per LDS instruction in the walk).  The next
// chunk's loads are issued before the walk and land in registers while the
// dependent chain (one multiply-subtract and one true division per element)
// runs; results go back through LDS so that the stores are coalesced again,
// one chunk late so that they are older than the loads the next round waits
// for.  LPW is chosen for enough wavefronts to cover the memory latency, not
// to fill lanes.
// ---------------------------------------------------------------------------
constexpr int kTdpC = 64;

// Row pitch 66 doubles: rows stay 16-byte aligned, so the walking lanes can move two
// elements per LDS instruction, and 16 lanes x 4 dwords at a lane stride of 132 dwords
// cover the 64 banks exactly once (no conflicts); the staging accesses (lane = column)
// are consecutive in any case.
template <int LPW>
struct alignas(16) TdpTile {
  double v[LPW][kTdpC + 2];
};

template <int LPW>
__device__ __forceinline__ void tdp_fetch(double (&r)[LPW], const double *__restrict__ a, int64_t line0,
                                          int nl, int64_t L, int64_t col, double fill) {
#pragma unroll
  for (int i = 0; i < LPW; i++) r[i] = (i < nl && col < L) ? a[(line0 + i) * L + col] : fill;
}

template <int LPW>
__device__ __forceinline__ void tdp_park(TdpTile<LPW> &t, const double (&r)[LPW], int lane) {
#pragma unroll
  for (int i = 0; i < LPW; i++) t.v[i][lane] = r[i];
}

template <int LPW>
__device__ __forceinline__ void tdp_flush(const TdpTile<LPW> &t, double *a, int64_t line0, int nl,
                                          int64_t L, int64_t col, int lane) {
#pragma unroll
  for (int i = 0; i < LPW; i++)
    if (i < nl && col < L) a[(line0 + i) * L + col] = t.v[i][lane];
}

// The walks below run in one lane per line.  A full chunk is processed in
// register blocks of kTdpB elements whose LDS reads are issued one block ahead
// of the dependent chain; a partial chunk takes the plain loop.
constexpr int kTdpB = 8;

template <int LPW>
__device__ __forceinline__ void tdp_ld_block(double (&r)[kTdpB], const TdpTile<LPW> &t, int lane, int j0) {
  const d2 *p = reinterpret_cast<const d2 *>(&t.v[lane][j0]);  // j0 is a multiple of kTdpB (even)
#pragma unroll
  for (int k = 0; k < kTdpB / 2; k++) {
    const d2 x = p[k];
    r[2 * k] = x.x;
    r[2 * k + 1] = x.y;
  }
}

template <int LPW>
__device__ __forceinline__ void tdp_st_block(TdpTile<LPW> &t, const double (&r)[kTdpB], int lane, int j0) {
  d2 *p = reinterpret_cast<d2 *>(&t.v[lane][j0]);
#pragma unroll
  for (int k = 0; k < kTdpB / 2; k++) {
    d2 x = {r[2 * k], r[2 * k + 1]};
    p[k] = x;
  }
}

// forward elimination over one chunk: y_j = (s_j - y_{j-1} lo_j) / dd_j
template <int LPW>
__device__ __forceinline__ double tdp_walk_forward(TdpTile<LPW> &ts, const TdpTile<LPW> &ta,
                                                   const TdpTile<LPW> &td, int lane, int cw, double prev) {
  if (cw == kTdpC) {
    double sv[2][kTdpB], av[2][kTdpB], dv[2][kTdpB];
    tdp_ld_block<LPW>(sv[0], ts, lane, 0);
    tdp_ld_block<LPW>(av[0], ta, lane, 0);
    tdp_ld_block<LPW>(dv[0], td, lane, 0);
#pragma unroll
    for (int b = 0; b < kTdpC / kTdpB; b++) {
      const int cur = b & 1;
      if (b + 1 < kTdpC / kTdpB) {
        tdp_ld_block<LPW>(sv[cur ^ 1], ts, lane, (b + 1) * kTdpB);
        tdp_ld_block<LPW>(av[cur ^ 1], ta, lane, (b + 1) * kTdpB);
        tdp_ld_block<LPW>(dv[cur ^ 1], td, lane, (b + 1) * kTdpB);
      }
#pragma unroll
      for (int k = 0; k < kTdpB; k++) {
        prev = (sv[cur][k] - prev * av[cur][k]) / dv[cur][k];
        sv[cur][k] = prev;
      }
      tdp_st_block<LPW>(ts, sv[cur], lane, b * kTdpB);
    }
  } else {
    for (int j = 0; j < cw; j++) {
      prev = (ts.v[lane][j] - prev * ta.v[lane][j]) / td.v[lane][j];
      ts.v[lane][j] = prev;
    }
  }
  return prev;
}

// back substitution over elements jtop..0 of one chunk: x_j = y_j - x_{j+1} ud_j
template <int LPW>
__device__ __forceinline__ double tdp_walk_backward(TdpTile<LPW> &ts, const TdpTile<LPW> &ta, int lane,
                                                    int jtop, double prev) {
  if (jtop == kTdpC - 1) {
    double sv[2][kTdpB], av[2][kTdpB];
    constexpr int NB = kTdpC / kTdpB;
    tdp_ld_block<LPW>(sv[0], ts, lane, (NB - 1) * kTdpB);
    tdp_ld_block<LPW>(av[0], ta, lane, (NB - 1) * kTdpB);
#pragma unroll
    for (int b = NB - 1; b >= 0; b--) {
      const int cur = (NB - 1 - b) & 1;
      if (b > 0) {
        tdp_ld_block<LPW>(sv[cur ^ 1], ts, lane, (b - 1) * kTdpB);
        tdp_ld_block<LPW>(av[cur ^ 1], ta, lane, (b - 1) * kTdpB);
      }
#pragma unroll
      for (int k = kTdpB - 1; k >= 0; k--) {
        prev = sv[cur][k] - prev * av[cur][k];
        sv[cur][k] = prev;
      }
      tdp_st_block<LPW>(ts, sv[cur], lane, b * kTdpB);
    }
  } else {
    for (int j = jtop; j >= 0; j--) {
      prev = ts.v[lane][j] - prev * ta.v[lane][j];
      ts.v[lane][j] = prev;
    }
  }
  return prev;
}

// Pivots dd and normalised upper diagonal ud of every line
// (include/solver.hpp:252-261 without the right-hand-side part).  The first
// element of a line has no predecessor: its lower coefficient is taken as +0
// and the running value starts at +0, which leaves d_0 = di_0 exactly.
template <int LPW>
__global__ __launch_bounds__(kTdpC) void k_tdp_factor(const double *__restrict__ lo,
                                                       const double *__restrict__ di,
                                                       const double *__restrict__ up,
                                                       double *__restrict__ dd,
                                                       double *__restrict__ ud, int64_t nlines,
                                                       int64_t L) {
  __shared__ TdpTile<LPW> ta, td, tu;
  const int lane = threadIdx.x;
  const int64_t line0 = (int64_t)blockIdx.x * LPW;
  const int nl = (int)((nlines - line0) < LPW ? (nlines - line0) : LPW);
  const int64_t nchunk = (L + kTdpC - 1) / kTdpC;
  double ra[LPW], rd[LPW], ru[LPW];
  tdp_fetch<LPW>(ra, lo, line0, nl, L, lane, 0.0);
  tdp_fetch<LPW>(rd, di, line0, nl, L, lane, 1.0);
  tdp_fetch<LPW>(ru, up, line0, nl, L, lane, 0.0);
  if (lane == 0) {
#pragma unroll
    for (int i = 0; i < LPW; i++) ra[i] = 0.0;
  }
  double uprev = 0.0;
  for (int64_t c = 0; c < nchunk; c++) {
    const int64_t col = c * kTdpC + lane;
    tdp_park<LPW>(ta, ra, lane);
    tdp_park<LPW>(td, rd, lane);
    tdp_park<LPW>(tu, ru, lane);
    __syncthreads();
    if (c + 1 < nchunk) {
      tdp_fetch<LPW>(ra, lo, line0, nl, L, col + kTdpC, 0.0);
      tdp_fetch<LPW>(rd, di, line0, nl, L, col + kTdpC, 1.0);
      tdp_fetch<LPW>(ru, up, line0, nl, L, col + kTdpC, 0.0);
    }
    if (lane < nl) {
      const int cw = (int)((L - c * kTdpC) < kTdpC ? (L - c * kTdpC) : kTdpC);
#pragma unroll 4
      for (int j = 0; j < cw; j++) {
        const double diag = td.v[lane][j] - uprev * ta.v[lane][j];
        uprev = tu.v[lane][j] / diag;
        td.v[lane][j] = diag;
        tu.v[lane][j] = uprev;
      }
    }
    __syncthreads();
    tdp_flush<LPW>(td, dd, line0, nl, L, col, lane);
    tdp_flush<LPW>(tu, ud, line0, nl, L, col, lane);
    __syncthreads();
  }
}

// out = M^-1 sol for every line (include/solver.hpp:255-271: forward
// elimination, then back substitution).  sol and out may be the same array.
template <int LPW>
__device__ __forceinline__ void tdp_apply_lines(TdpTile<LPW> (&ts)[2], TdpTile<LPW> &ta, TdpTile<LPW> &td,
                                                const double *__restrict__ lo, const double *__restrict__ dd,
                                                const double *__restrict__ ud, const double *sol, double *out,
                                                int64_t line0, int nl, int64_t L, int lane) {
  // The running right-hand side has two tiles: chunk c is walked in one while chunk c-1
  // leaves the other.  Its stores are issued before the loads of chunk c+1, so the wait
  // for those loads at the top of the next round (the memory counter retires in order)
  // never stalls on stores that were issued after them.
  const int64_t nchunk = (L + kTdpC - 1) / kTdpC;
  double rs[LPW], ra[LPW], rd[LPW];
  tdp_fetch<LPW>(rs, sol, line0, nl, L, lane, 0.0);
  tdp_fetch<LPW>(ra, lo, line0, nl, L, lane, 0.0);
  tdp_fetch<LPW>(rd, dd, line0, nl, L, lane, 1.0);
  if (lane == 0) {
    // no predecessor: y_0 = (s_0 - (+0)(+0)) / d_0 = s_0 / d_0 exactly
#pragma unroll
    for (int i = 0; i < LPW; i++) ra[i] = 0.0;
  }
  double prev = 0.0;
  for (int64_t c = 0; c < nchunk; c++) {
    const int64_t col = c * kTdpC + lane;
    TdpTile<LPW> &cur = ts[c & 1];
    tdp_park<LPW>(cur, rs, lane);
    tdp_park<LPW>(ta, ra, lane);
    tdp_park<LPW>(td, rd, lane);
    __syncthreads();
    if (c > 0) tdp_flush<LPW>(ts[(c - 1) & 1], out, line0, nl, L, col - kTdpC, lane);
    if (c + 1 < nchunk) {
      tdp_fetch<LPW>(rs, sol, line0, nl, L, col + kTdpC, 0.0);
      tdp_fetch<LPW>(ra, lo, line0, nl, L, col + kTdpC, 0.0);
      tdp_fetch<LPW>(rd, dd, line0, nl, L, col + kTdpC, 1.0);
    }
    if (lane < nl) {
      const int cw = (int)((L - c * kTdpC) < kTdpC ? (L - c * kTdpC) : kTdpC);
      prev = tdp_walk_forward<LPW>(cur, ta, td, lane, cw, prev);
    }
    __syncthreads();
  }
  tdp_flush<LPW>(ts[(nchunk - 1) & 1], out, line0, nl, L, (nchunk - 1) * kTdpC + lane, lane);
  // back substitution: element L-1 keeps its forward value (= prev); walk the
  // chunks in reverse.  Every element of out is re-read by the thread that stored it.
  tdp_fetch<LPW>(rs, out, line0, nl, L, (nchunk - 1) * kTdpC + lane, 0.0);
  tdp_fetch<LPW>(ra, ud, line0, nl, L, (nchunk - 1) * kTdpC + lane, 0.0);
  for (int64_t c = nchunk - 1; c >= 0; c--) {
    const int64_t col = c * kTdpC + lane;
    TdpTile<LPW> &cur = ts[c & 1];
    tdp_park<LPW>(cur, rs, lane);
    tdp_park<LPW>(ta, ra, lane);
    __syncthreads();
    if (c + 1 < nchunk) tdp_flush<LPW>(ts[(c + 1) & 1], out, line0, nl, L, col + kTdpC, lane);
    if (c > 0) {
      tdp_fetch<LPW>(rs, out, line0, nl, L, col - kTdpC, 0.0);
      tdp_fetch<LPW>(ra, ud, line0, nl, L, col - kTdpC, 0.0);
    }
    if (lane < nl) {
      const int cw = (int)((L - c * kTdpC) < kTdpC ? (L - c * kTdpC) : kTdpC);
      // the last chunk starts one element lower and therefore never takes the blocked walk
      const int jtop = (c == nchunk - 1) ? cw - 2 : cw - 1;
      prev = tdp_walk_backward<LPW>(cur, ta, lane, jtop, prev);
    }
    __syncthreads();
  }
  tdp_flush<LPW>(ts[0], out, line0, nl, L, lane, lane);
}

template <int LPW>
__global__ __launch_bounds__(kTdpC) void k_tdp_apply(const double *__restrict__ lo,
                                                      const double *__restrict__ dd,
                                                      const double *__restrict__ ud,
                                                      const double *sol, double *out,
                                                      int64_t nlines, int64_t L,
                                                      const int *__restrict__ stop) {
  if (stopped(stop)) return;
  __shared__ TdpTile<LPW> ts[2], ta, td;
  const int64_t line0 = (int64_t)blockIdx.x * LPW;
  const int nl = (int)((nlines - line0) < LPW ? (nlines - line0) : LPW);
  tdp_apply_lines<LPW>(ts, ta, td, lo, dd, ud, sol, out, line0, nl, L, threadIdx.x);
}

// ---------------------------------------------------------------------------
// The same solve with requests of 16 bytes per lane: one instruction moves the 64 columns of a
// chunk for TWO lines (lanes 0-31 one line, lanes 32-63 the next), i.e. half the requests per
// chunk.  The steady state is straight-line code without conditions on lines or columns, so the
// compiler waits for exactly the chunk it is about to park instead of for every outstanding
// request; it covers the chunks that are complete; the incomplete chunk at the end of a line
// goes through the 8-byte path.  kTdpAheadF / kTdpAheadB chunks are requested ahead in the two
// passes; measured, 1 to 3 make no difference (both passes then run at the rate at which the
// memory system serves this access pattern, about 5 TB/s over all workgroups), so it is 1.
// Needs an even line length and 16-byte aligned arrays.  Tiles, recurrences and their order are
// those of k_tdp_apply.
// ---------------------------------------------------------------------------
constexpr int kTdpAheadF = 1;
constexpr int kTdpAheadB = 1;

// A workgroup with fewer than LPW lines treats its last line as the missing ones too: their
// lanes load, walk and store that line again (same values to the same addresses), which
// keeps the instruction stream free of conditions.
template <int LPW>
__device__ __forceinline__ void tdp2_fetch(d2 (&r)[LPW / 2], const double *a, int64_t line0, int nl, int64_t L,
                                           int64_t col0, int lane) {
  const double *p0 = a + line0 * L + col0 + 2 * (lane & 31);
  const int h = lane >> 5;
#pragma unroll
  for (int p = 0; p < LPW / 2; p++) {
    const int line = 2 * p + h < nl ? 2 * p + h : nl - 1;
    r[p] = *reinterpret_cast<const d2 *>(p0 + line * L);
  }
}

template <int LPW>
__device__ __forceinline__ void tdp2_park(TdpTile<LPW> &t, const d2 (&r)[LPW / 2], int lane) {
  const int h = lane >> 5, j = 2 * (lane & 31);
#pragma unroll
  for (int p = 0; p < LPW / 2; p++) *reinterpret_cast<d2 *>(&t.v[2 * p + h][j]) = r[p];
}

template <int LPW>
__device__ __forceinline__ void tdp2_flush(const TdpTile<LPW> &t, double *a, int64_t line0, int nl, int64_t L,
                                           int64_t col0, int lane) {
  const int h = lane >> 5, j = 2 * (lane & 31);
  double *p0 = a + line0 * L + col0 + j;
#pragma unroll
  for (int p = 0; p < LPW / 2; p++) {
    const int line = 2 * p + h < nl ? 2 * p + h : nl - 1;
    *reinterpret_cast<d2 *>(p0 + line * L) = *reinterpret_cast<const d2 *>(&t.v[2 * p + h][j]);
  }
}

// Normalisation on the way in: a fetched chunk of the un-normalised vector is divided by its norm (true
// division, as normalize() in include/matrix.hpp:54-60 and k_scale_from), stored to the basis slot it belongs
// to and handed on to the solve.
template <int LPW>
__device__ __forceinline__ void tdp2_scale_store(d2 (&r)[LPW / 2], double nrm, double *v, int64_t pitch, int64_t line0,
                                                 int nl, int64_t L, int64_t col0, int lane) {
  const int64_t e0 = line0 * L + col0 + 2 * (lane & 31);
  const int h = lane >> 5;
#pragma unroll
  for (int p = 0; p < LPW / 2; p++) {
    const int line = 2 * p + h < nl ? 2 * p + h : nl - 1;
    r[p].x = r[p].x / nrm;
    r[p].y = r[p].y / nrm;
    st_nt(r[p], reinterpret_cast<d2 *>(v + il1(e0 + line * L, pitch)));
  }
}

// 8-byte counterparts for the incomplete chunk at the end of a line, with the same treatment
// of missing lines (their lanes must keep walking copies of the last line, or their stores
// in the back substitution would put other values at that line's addresses)
template <int LPW>
__device__ __forceinline__ void tdp_fetch_dup(double (&r)[LPW], const double *__restrict__ a, int64_t line0,
                                              int nl, int64_t L, int64_t col, double fill) {
#pragma unroll
  for (int i = 0; i < LPW; i++) r[i] = col < L ? a[(line0 + (i < nl ? i : nl - 1)) * L + col] : fill;
}

template <int LPW>
__device__ __forceinline__ void tdp_flush_dup(const TdpTile<LPW> &t, double *a, int64_t line0, int nl,
                                              int64_t L, int64_t col, int lane) {
#pragma unroll
  for (int i = 0; i < LPW; i++)
    if (col < L) a[(line0 + (i < nl ? i : nl - 1)) * L + col] = t.v[i][lane];
}

// compact walks for the few chunks outside the steady state
template <int LPW>
__device__ __forceinline__ double tdp_walk_forward_small(TdpTile<LPW> &ts, const TdpTile<LPW> &ta,
                                                      const TdpTile<LPW> &td, int lane, int cw, double prev) {
#pragma unroll 1
  for (int j = 0; j < cw; j++) {
    prev = (ts.v[lane][j] - prev * ta.v[lane][j]) / td.v[lane][j];
    ts.v[lane][j] = prev;
  }
  return prev;
}

template <int LPW>
__device__ __forceinline__ double tdp_walk_backward_small(TdpTile<LPW> &ts, const TdpTile<LPW> &ta, int lane,
                                                       int jtop, double prev) {
#pragma unroll 1
  for (int j = jtop; j >= 0; j--) {
    prev = ts.v[lane][j] - prev * ta.v[lane][j];
    ts.v[lane][j] = prev;
  }
  return prev;
}

// SC: sol is the un-normalised work vector of an Arnoldi step and sc.part the partial sums of its squares.  The
// solve takes sol / nrm, nrm = sqrt(sum of sc.part) -- the expression tree of sum_parts for 256 threads, evaluated
// by the workgroup's one wavefront, so the bits k_scale_from would obtain -- and stores that vector to sc.vout
// (sc.pitch: as a vector of a chunk-interleaved basis) and nrm to *sc.nout: the normalisation at the end of
// gram_schmidt (include/solver.hpp:209) and M^-1 v (:290-328) in one pass over the vector.  Needs L >= 8 * 64.
template <int LPW, bool SC = false, int DF = kTdpAheadF, int DB = kTdpAheadB>
__global__ __launch_bounds__(kTdpC) void k_tdp_apply2(const double *__restrict__ lo,
                                                       const double *__restrict__ dd,
                                                       const double *__restrict__ ud, const double *sol,
                                                       double *out, int64_t nlines, int64_t L,
                                                       const int *__restrict__ stop, TdpScale sc) {
  if (stopped(stop)) return;
  static_assert(LPW % 2 == 0, "two lines per request");
  constexpr int H = LPW / 2, C = kTdpC;
  __shared__ TdpTile<LPW> ts[2], ta, td;
  const int lane = threadIdx.x;
  const int64_t line0 = (int64_t)blockIdx.x * LPW;
  const int nl = (int)((nlines - line0) < LPW ? (nlines - line0) : LPW);
  const int64_t nfull = L / C;  // chunks whose 64 columns all exist
  const int tail = (int)(L - nfull * C);  // columns of the incomplete chunk behind them
  if (nfull < 8) {
    if (!SC) tdp_apply_lines<LPW>(ts, ta, td, lo, dd, ud, sol, out, line0, nl, L, lane);
    return;
  }
  double nrm = 1.0;
  if (SC) {
    double t[kBlock / 64];
#pragma unroll
    for (int wv = 0; wv < kBlock / 64; wv++) {
      double a = 0.0;
      for (int i = wv * 64 + lane; i < sc.np; i += kBlock) a += sc.part[i];
      t[wv] = wave_sum(a);
    }
    double tot = t[0];
#pragma unroll
    for (int wv = 1; wv < kBlock / 64; wv++) tot += t[wv];
    nrm = sqrt(__shfl(tot, 0, 64));
    if (blockIdx.x == 0 && lane == 0 && sc.nout != nullptr) *sc.nout = nrm;
  }
  const bool walker = lane < LPW;
  double prev = 0.0;
  int tp = 0;  // tile of the chunk in hand
  {
    constexpr int D = DF;
    d2 rs[D][H], ra[D][H], rd[D][H];
#pragma unroll
    for (int k = 0; k < D; k++) {
      tdp2_fetch<LPW>(rs[k], sol, line0, nl, L, (int64_t)k * C, lane);
      tdp2_fetch<LPW>(ra[k], lo, line0, nl, L, (int64_t)k * C, lane);
      tdp2_fetch<LPW>(rd[k], dd, line0, nl, L, (int64_t)k * C, lane);
    }
    if ((lane & 31) == 0) {
      // column 0 has no predecessor: y_0 = (s_0 - (+0)(+0)) / d_0 = s_0 / d_0 exactly
#pragma unroll
      for (int p = 0; p < H; p++) ra[0][p].x = 0.0;
    }
    // chunk 0: nothing to flush yet
    if (SC) tdp2_scale_store<LPW>(rs[0], nrm, sc.vout, sc.pitch, line0, nl, L, 0, lane);
    tdp2_park<LPW>(ts[0], rs[0], lane);
    tdp2_park<LPW>(ta, ra[0], lane);
    tdp2_park<LPW>(td, rd[0], lane);
    __syncthreads();
    tdp2_fetch<LPW>(rs[0], sol, line0, nl, L, (int64_t)D * C, lane);
    tdp2_fetch<LPW>(ra[0], lo, line0, nl, L, (int64_t)D * C, lane);
    tdp2_fetch<LPW>(rd[0], dd, line0, nl, L, (int64_t)D * C, lane);
    if (walker) prev = tdp_walk_forward<LPW>(ts[0], ta, td, lane, C, prev);
    __syncthreads();
    int64_t c = 1;
    for (; c + D <= nfull; c += D) {
#pragma unroll
      for (int k = 0; k < D; k++) {
        const int set = (1 + k) % D;
        const int64_t cc = c + k;
        const int64_t cf = (cc + D < nfull ? cc + D : nfull - 1) * C;  // past the end: a chunk nobody uses
        TdpTile<LPW> &cur = ts[cc & 1];
        if (SC) tdp2_scale_store<LPW>(rs[set], nrm, sc.vout, sc.pitch, line0, nl, L, cc * C, lane);
        tdp2_park<LPW>(cur, rs[set], lane);
        tdp2_park<LPW>(ta, ra[set], lane);
        tdp2_park<LPW>(td, rd[set], lane);
        __syncthreads();
        tdp2_flush<LPW>(ts[(cc - 1) & 1], out, line0, nl, L, (cc - 1) * C, lane);
        tdp2_fetch<LPW>(rs[set], sol, line0, nl, L, cf, lane);
        tdp2_fetch<LPW>(ra[set], lo, line0, nl, L, cf, lane);
        tdp2_fetch<LPW>(rd[set], dd, line0, nl, L, cf, lane);
        if (walker) prev = tdp_walk_forward<LPW>(cur, ta, td, lane, C, prev);
        __syncthreads();
      }
    }
    // fewer than D complete chunks are left
#pragma unroll
    for (int k = 0; k < D - 1; k++) {
      const int set = (1 + k) % D;
      const int64_t cc = c + k;
      if (cc < nfull) {
        TdpTile<LPW> &cur = ts[cc & 1];
        if (SC) tdp2_scale_store<LPW>(rs[set], nrm, sc.vout, sc.pitch, line0, nl, L, cc * C, lane);
        tdp2_park<LPW>(cur, rs[set], lane);
        tdp2_park<LPW>(ta, ra[set], lane);
        tdp2_park<LPW>(td, rd[set], lane);
        __syncthreads();
        tdp2_flush<LPW>(ts[(cc - 1) & 1], out, line0, nl, L, (cc - 1) * C, lane);
        if (walker) prev = tdp_walk_forward_small<LPW>(cur, ta, td, lane, C, prev);
        __syncthreads();
      }
    }
    tp = (int)((nfull - 1) & 1);
    tdp2_flush<LPW>(ts[tp], out, line0, nl, L, (nfull - 1) * C, lane);
  }
  // Back substitution.  Element L-1 keeps its forward value (= prev).  Every element of out is
  // re-read by the lane that stored it.  The complete chunks are requested first, kTdpAheadB of
  // them, then the incomplete chunk is dealt with through the 8-byte path.
  {
    constexpr int D = DB;
    d2 rs[D][H], ra[D][H];
#pragma unroll
    for (int k = 0; k < D; k++) {
      tdp2_fetch<LPW>(rs[k], out, line0, nl, L, (nfull - 1 - k) * C, lane);
      tdp2_fetch<LPW>(ra[k], ud, line0, nl, L, (nfull - 1 - k) * C, lane);
    }
    if (tail > 0) {
      double r8s[LPW], r8a[LPW], r8d[LPW];
      const int64_t col = nfull * C + lane;
      tdp_fetch_dup<LPW>(r8s, sol, line0, nl, L, col, 0.0);
      tdp_fetch_dup<LPW>(r8a, lo, line0, nl, L, col, 0.0);
      tdp_fetch_dup<LPW>(r8d, dd, line0, nl, L, col, 1.0);
      if (SC && col < L) {
#pragma unroll
        for (int i = 0; i < LPW; i++) {
          r8s[i] = r8s[i] / nrm;
          sc.vout[il1((line0 + (i < nl ? i : nl - 1)) * L + col, sc.pitch)] = r8s[i];
        }
      }
      tp ^= 1;
      tdp_park<LPW>(ts[tp], r8s, lane);
      tdp_park<LPW>(ta, r8a, lane);
      tdp_park<LPW>(td, r8d, lane);
      __syncthreads();
      tdp_fetch_dup<LPW>(r8a, ud, line0, nl, L, col, 0.0);
      if (walker) prev = tdp_walk_forward_small<LPW>(ts[tp], ta, td, lane, tail, prev);
      __syncthreads();
      tdp_park<LPW>(ta, r8a, lane);
      __syncthreads();
      if (walker) prev = tdp_walk_backward_small<LPW>(ts[tp], ta, lane, tail - 2, prev);
      __syncthreads();
    }
    // step 0 (chunk nfull-1): flushes the incomplete chunk, if there is one
    {
      const int prev_tile = tp;
      tp ^= 1;
      tdp2_park<LPW>(ts[tp], rs[0], lane);
      tdp2_park<LPW>(ta, ra[0], lane);
      __syncthreads();
      if (tail > 0) tdp_flush_dup<LPW>(ts[prev_tile], out, line0, nl, L, nfull * C + lane, lane);
      tdp2_fetch<LPW>(rs[0], out, line0, nl, L, (nfull - 1 - D) * C, lane);
      tdp2_fetch<LPW>(ra[0], ud, line0, nl, L, (nfull - 1 - D) * C, lane);
      if (walker) prev = tdp_walk_backward_small<LPW>(ts[tp], ta, lane, tail > 0 ? C - 1 : C - 2, prev);
      __syncthreads();
    }
    int64_t s1 = 1;
    for (; s1 + D <= nfull; s1 += D) {
#pragma unroll
      for (int k = 0; k < D; k++) {
        const int set = (1 + k) % D;
        const int64_t cc = nfull - 1 - (s1 + k);
        const int64_t cf = (cc - D > 0 ? cc - D : 0) * C;  // before the start: a chunk nobody uses
        const int prev_tile = tp;
        tp ^= 1;
        TdpTile<LPW> &cur = ts[tp];
        tdp2_park<LPW>(cur, rs[set], lane);
        tdp2_park<LPW>(ta, ra[set], lane);
        __syncthreads();
        tdp2_flush<LPW>(ts[prev_tile], out, line0, nl, L, (cc + 1) * C, lane);
        tdp2_fetch<LPW>(rs[set], out, line0, nl, L, cf, lane);
        tdp2_fetch<LPW>(ra[set], ud, line0, nl, L, cf, lane);
        if (walker) prev = tdp_walk_backward<LPW>(cur, ta, lane, C - 1, prev);
        __syncthreads();
      }
    }
#pragma unroll
    for (int k = 0; k < D - 1; k++) {
      const int set = (1 + k) % D;
      const int64_t cc = nfull - 1 - (s1 + k);
      if (cc >= 0) {
        const int prev_tile = tp;
        tp ^= 1;
        tdp2_park<LPW>(ts[tp], rs[set], lane);
        tdp2_park<LPW>(ta, ra[set], lane);
        __syncthreads();
        tdp2_flush<LPW>(ts[prev_tile], out, line0, nl, L, (cc + 1) * C, lane);
        if (walker) prev = tdp_walk_backward_small<LPW>(ts[tp], ta, lane, C - 1, prev);
        __syncthreads();
      }
    }
    tdp2_flush<LPW>(ts[tp], out, line0, nl, L, 0, lane);
  }
}

__global__ __launch_bounds__(kBlock) void k_tdp_extract_csr(const int64_t *__restrict__ rowptr,
                                                             const int32_t *__restrict__ col,
                                                             const double *__restrict__ val,
                                                             int64_t n, double *__restrict__ lo,
                                                             double *__restrict__ di,
                                                             double *__restrict__ up) {
  for (int64_t r = (int64_t)blockIdx.x * kBlock + threadIdx.x; r < n; r += (int64_t)gridDim.x * kBlock) {
    double a = 0.0, d = 0.0, c = 0.0;
    for (int64_t p = rowptr[r]; p < rowptr[r + 1]; p++) {
      const int64_t cc = col[p];
      if (cc == r - 1) a = val[p];
      if (cc == r) d = val[p];
      if (cc == r + 1) c = val[p];
    }
    lo[r] = a;
    di[r] = d;
    up[r] = c;
  }
}

// ---------------------------------------------------------------------------
// Implicit-Euler assembly
// ---------------------------------------------------------------------------
// BoundaryCondition::ghost_cell, include/boundary_conds.hpp:62-66
__device__ __forceinline__ double ghost_cell(double bndry, double bc, double size, double weight) {
  return (2.0 * bc - bndry) * weight + (size * bc + bndry) * (1.0 - weight);
}

// Slab geometry: a rank owns the padded rows [row0, row0 + rows) of the
// (cx+2) x (cy+2) mesh; field and vector pointers address its first owned row
// and have one valid halo row of P entries in front and behind.  A single GPU
// owns everything (row0 = 0, rows = cx + 2).

// Copy of the right boundary column (interior cells of padded row cx) of the
// three fields, wrap[f * cy + j]; zeros on ranks that do not own that row, so
// that a sum over ranks hands the column to everybody.
__global__ __launch_bounds__(kBlock) void k_pack_wrap(IeGeom g, const double *__restrict__ T,
                                                       const double *__restrict__ u,
                                                       const double *__restrict__ v,
                                                       double *__restrict__ wrap) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= 3 * g.cy) return;
  const int f = t / g.cy, j = t % g.cy;
  const int P = g.cy + 2;
  const int lr = g.cx - g.row0;  // local index of padded row cx
  const double *F = f == 0 ? T : (f == 1 ? u : v);
  wrap[t] = (lr >= 0 && lr < g.rows) ? F[(size_t)lr * P + (j + 1)] : 0.0;
}

// All twelve ghost fills of TimeDisc::apply_bcs (include/time_disc.hpp:146-163)
// in one launch: each reads interior cells only and writes its own ghost
// strip, so their order is immaterial.  View wiring as in the TimeDisc
// constructor (:51-134): left conditions read the RIGHT boundary column, which
// (like the right conditions' source) is taken from `wrap`.
// bcvals layout per field: [bottom cx][top cx][left cy][right cy], global.
__global__ __launch_bounds__(kBlock) void k_bc_fill(IeGeom g, IeBc bc, double *__restrict__ T,
                                                     double *__restrict__ u, double *__restrict__ v,
                                                     const double *__restrict__ bcvals,
                                                     const double *__restrict__ wrap) {
  const int per_field = 2 * (g.cx + g.cy);
  const int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= 3 * per_field) return;
  const int f = t / per_field;
  int r = t % per_field;
  double *F = f == 0 ? T : (f == 1 ? u : v);
  const int P = g.cy + 2;
  const double bcv = bcvals[(size_t)f * per_field + r];
  if (r < 2 * g.cx) {  // bottom (r < cx) or top ghost of interior column i
    const bool top = r >= g.cx;
    const int i = top ? r - g.cx : r;
    const int lr = i + 1 - g.row0;
    if (lr < 0 || lr >= g.rows) return;
    double *row = F + (size_t)lr * P;
    if (top)
      row[g.cy + 1] = ghost_cell(row[g.cy], bcv, g.dy, bc.weight[f * 4 + KRY_SIDE_TOP]);
    else
      row[0] = ghost_cell(row[1], bcv, g.dy, bc.weight[f * 4 + KRY_SIDE_BOTTOM]);
    return;
  }
  r -= 2 * g.cx;
  const bool right = r >= g.cy;
  const int j = right ? r - g.cy : r;
  const int lr = (right ? g.cx + 1 : 0) - g.row0;
  if (lr < 0 || lr >= g.rows) return;
  F[(size_t)lr * P + (j + 1)] = ghost_cell(wrap[f * g.cy + j], bcv, g.dx,
                                           bc.weight[f * 4 + (right ? KRY_SIDE_RIGHT : KRY_SIDE_LEFT)]);
}

// Off-diagonal bands, include/time_disc.hpp:249-274 with
// SecondOrderCentered::Dx_m1/Dx_p1/Dy_m1/Dy_p1 (include/space_disc.hpp:83-113).
// Everything else is cleared; corner entries get a unit diagonal so that the
// padded operator stays non-singular (their vector entries are always 0).
__global__ __launch_bounds__(kBlock) void k_offdiag(IeGeom g, const double *__restrict__ u,
                                                     const double *__restrict__ v,
                                                     double *__restrict__ bw, double *__restrict__ bs,
                                                     double *__restrict__ bc, double *__restrict__ bn,
                                                     double *__restrict__ be) {
  const int P = g.cy + 2;
  const int64_t nloc = (int64_t)g.rows * P;
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nloc; k += (int64_t)gridDim.x * kBlock) {
    const int ip = g.row0 + (int)(k / P), jp = (int)(k % P);
    const bool interior = ip >= 1 && ip <= g.cx && jp >= 1 && jp <= g.cy;
    const bool corner = (ip == 0 || ip == g.cx + 1) && (jp == 0 || jp == g.cy + 1);
    double w = 0.0, s = 0.0, n = 0.0, e = 0.0;
    if (interior) {
      const double kx = g.diffusion * g.inv_rp / g.dx;
      const double ky = g.diffusion * g.inv_rp / g.dy;
      s = (-v[k - 1] / 2.0 - ky) / g.dy;
      n = (v[k + 1] / 2.0 - ky) / g.dy;
      w = (-u[k - P] / 2.0 - kx) / g.dx;
      e = (u[k + P] / 2.0 - kx) / g.dx;
    }
    bw[k] = w;
    bs[k] = s;
    bn[k] = n;
    be[k] = e;
    bc[k] = corner ? 1.0 : 0.0;
  }
}

// ImplicitEuler::assemble_system after the ghost fill, include/time_disc.hpp:294-337:
// interior diagonal Dx_0 + Dy_0 + 1/dt, right-hand side flux_integral +
// source_fd (include/space_disc.hpp:19-44, :116-179), ghost rows
// (rhs 0, inner coefficient 1, diagonal 2w-1; the right ghost column takes the
// LEFT temperature condition's term, :335-336).
__global__ __launch_bounds__(kBlock) void k_assemble(IeGeom g, IeBc bc, double dt,
                                                      const double *__restrict__ T,
                                                      const double *__restrict__ u,
                                                      const double *__restrict__ v,
                                                      double *__restrict__ bw, double *__restrict__ bs,
                                                      double *__restrict__ bcn, double *__restrict__ bn,
                                                      double *__restrict__ be,
                                                      double *__restrict__ rhs) {
  const int P = g.cy + 2;
  const int64_t nloc = (int64_t)g.rows * P;
  const double dx = g.dx, dy = g.dy;
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nloc; k += (int64_t)gridDim.x * kBlock) {
    const int ip = g.row0 + (int)(k / P), jp = (int)(k % P);
    const bool xin = ip >= 1 && ip <= g.cx, yin = jp >= 1 && jp <= g.cy;
    if (xin && yin) {
      const double Dx0 = 2.0 * g.diffusion * g.inv_rp / (dx * dx);
      const double Dy0 = 2.0 * g.diffusion * g.inv_rp / (dy * dy);
      bcn[k] = Dx0 + Dy0 + 1.0 / dt;
      const double Tc = T[k], Tw = T[k - P], Te = T[k + P], Ts = T[k - 1], Tn = T[k + 1];
      const double uc = u[k], uw = u[k - P], ue = u[k + P], us = u[k - 1], un = u[k + 1];
      const double vc = v[k], vw = v[k - P], ve = v[k + P], vs = v[k - 1], vn = v[k + 1];
      // flux_integral
      const double fx_hi = (Tc * uc + Te * ue) / 2.0;  // uT_x_flux(i, j)
      const double fx_lo = (Tw * uw + Tc * uc) / 2.0;  // uT_x_flux(i-1, j)
      const double fy_hi = (Tc * vc + Tn * vn) / 2.0;  // vT_y_flux(i, j)
      const double fy_lo = (Ts * vs + Tc * vc) / 2.0;  // vT_y_flux(i, j-1)
      const double x_deriv = (fx_hi - fx_lo) / dx;
      const double y_deriv = (fy_hi - fy_lo) / dy;
      const double x2_deriv = ((Te - Tc) / dx - (Tc - Tw) / dx) / dx;
      const double y2_deriv = ((Tn - Tc) / dy - (Tc - Ts) / dy) / dy;
      const double lapl = (x2_deriv + y2_deriv) * g.inv_rp;
      const double flux = (-x_deriv - y_deriv) + g.diffusion * lapl;
      // source_fd
      const double u_dx = (ue - uw) / (2.0 * dx);
      const double v_dy = (vn - vs) / (2.0 * dy);
      const double u_dy = (un - us) / (2.0 * dy);
      const double v_dx = (ve - vw) / (2.0 * dx);
      const double cross = u_dy + v_dx;
      const double src = g.eckert / g.reynolds * (2.0 * (u_dx * u_dx + v_dy * v_dy) + cross * cross);
      rhs[k] = flux + src;
    } else {
      rhs[k] = 0.0;
      if (xin && jp == g.cy + 1) {  // top ghost: inner neighbour is k-1
        bs[k] = 1.0;
        bcn[k] = 2.0 * bc.weight[KRY_SIDE_TOP] - 1.0;
      } else if (xin && jp == 0) {  // bottom ghost: inner neighbour is k+1
        bn[k] = 1.0;
        bcn[k] = 2.0 * bc.weight[KRY_SIDE_BOTTOM] - 1.0;
      } else if (yin && ip == 0) {  // left ghost: inner neighbour is k+P
        be[k] = 1.0;
        bcn[k] = 2.0 * bc.weight[KRY_SIDE_LEFT] - 1.0;
      } else if (yin && ip == g.cx + 1) {  // right ghost: inner neighbour is k-P
        bw[k] = 1.0;
        bcn[k] = 2.0 * bc.weight[KRY_SIDE_LEFT] - 1.0;
      }
    }
  }
}

// State update of ImplicitEuler::timestep, include/time_disc.hpp:281-286:
// T += delta on interior cells, and the ingredients of the returned maximum.
//
// The reference folds `max_delta = std::max(std::abs(delta), max_delta)` over the
// interior cells in (i, j) order (:284).  std::max(a, b) is `(a < b) ? b : a`, and
// every comparison with a NaN is false: a NaN |delta| replaces the running maximum,
// and the next finite |delta| replaces a NaN maximum.  The fold therefore returns
// the maximum over the cells AFTER the last NaN cell, and NaN only when the last
// interior cell is NaN.  Here every workgroup records the largest |delta| that is
// not NaN (part[2 b]) and the largest position of a NaN (part[2 b + 1]; position =
// index in the padded layout of the whole mesh, which grows in (i, j) order; -1 if
// none); k_max_fold turns them into the reference's value.
__device__ __forceinline__ double max_skip_nan(double a, double m) { return a > m ? a : m; }

__global__ __launch_bounds__(kBlock) void k_update_max(IeGeom g, double *__restrict__ T,
                                                        const double *__restrict__ delta,
                                                        double *__restrict__ part) {
  __shared__ double s[2 * (kBlock / 64)];
  const int P = g.cy + 2;
  const int64_t nloc = (int64_t)g.rows * P;
  double m = 0.0, last_nan = -1.0;
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nloc; k += (int64_t)gridDim.x * kBlock) {
    const int ip = g.row0 + (int)(k / P), jp = (int)(k % P);
    if (ip >= 1 && ip <= g.cx && jp >= 1 && jp <= g.cy) {
      const double d = delta[k];
      T[k] = T[k] + d;
      const double a = fabs(d);
      m = max_skip_nan(a, m);
      if (a != a) last_nan = (double)((int64_t)g.row0 * P + k);  // positions grow along a thread's walk
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    m = max_skip_nan(__shfl_down(m, off, 64), m);
    last_nan = max_skip_nan(__shfl_down(last_nan, off, 64), last_nan);
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) {
    s[2 * wv] = m;
    s[2 * wv + 1] = last_nan;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < kBlock / 64; i++) {
      m = max_skip_nan(s[2 * i], m);
      last_nan = max_skip_nan(s[2 * i + 1], last_nan);
    }
    part[2 * blockIdx.x] = m;
    part[2 * blockIdx.x + 1] = last_nan;
  }
}

// The reference's fold from the partial results of k_update_max, in up to three
// phases (bits of `phases`) so that a row-block decomposition can reduce over
// the ranks in between; scal = {result, last NaN position, largest |delta| that is not NaN}.
//   1: collapse the workgroups' partial results into scal[1], scal[2]
//   2: scal[0] = scal[2] if no cell is NaN; otherwise the largest |delta| over this
//      rank's interior cells behind the last NaN position (one workgroup walks them:
//      only runs whose solve has broken down get here)
//   4: scal[0] = NaN if the last interior cell of the mesh is the last NaN position
constexpr int kFoldBlock = 1024;

__device__ __forceinline__ double fold_block_max(double v, double *s) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max_skip_nan(__shfl_down(v, off, 64), v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s[wv] = v;
  __syncthreads();
  double t = s[0];
  for (int i = 1; i < kFoldBlock / 64; i++) t = max_skip_nan(s[i], t);
  return t;  // the same value in every thread
}

__global__ __launch_bounds__(kFoldBlock) void k_max_fold(IeGeom g, const double *__restrict__ delta,
                                                          const double *__restrict__ part, int np,
                                                          double *__restrict__ scal, int phases) {
  __shared__ double s[kFoldBlock / 64];
  const int P = g.cy + 2;
  double last_nan, mx;
  if (phases & 1) {
    double m = 0.0, l = -1.0;
    for (int i = threadIdx.x; i < np; i += kFoldBlock) {
      m = max_skip_nan(part[2 * i], m);
      l = max_skip_nan(part[2 * i + 1], l);
    }
    mx = fold_block_max(m, s);
    last_nan = fold_block_max(l, s);
    if (threadIdx.x == 0) {
      scal[1] = last_nan;
      scal[2] = mx;
    }
  } else {
    last_nan = scal[1];
    mx = scal[2];
  }
  double res = mx;
  if (phases & 2) {
    if (last_nan >= 0.0) {
      const int64_t nloc = (int64_t)g.rows * P;
      int64_t k0 = (int64_t)last_nan + 1 - (int64_t)g.row0 * P;
      if (k0 < 0) k0 = 0;
      double m = 0.0;
      for (int64_t k = k0 + threadIdx.x; k < nloc; k += kFoldBlock) {
        const int ip = g.row0 + (int)(k / P), jp = (int)(k % P);
        if (ip >= 1 && ip <= g.cx && jp >= 1 && jp <= g.cy) m = max_skip_nan(fabs(delta[k]), m);
      }
      res = fold_block_max(m, s);
    }
    if (threadIdx.x == 0) scal[0] = res;
  } else {
    res = scal[0];
  }
  if ((phases & 4) && threadIdx.x == 0) {
    const double last_cell = (double)((int64_t)g.cx * P + g.cy);
    if (last_nan == last_cell) scal[0] = __builtin_nan("");
  }
}

// ---------------------------------------------------------------------------
// ImplicitEulerApproxFact (include/time_disc.hpp:388-540):
//   (I + dt Dy)(I + dt Dx) dT = dt (flux_integral + source_fd)
// First one tridiagonal system per mesh column along y (contiguous lines: the
// preconditioner's line solver does them), then one per mesh row along x.
// ---------------------------------------------------------------------------
// Lines along y in padded layout: coefficients of assemble_system_dy (:487-507)
// and the right-hand side of :450-458.  The two ghost rows of the padded array
// are not part of any system and become identity lines.
__global__ __launch_bounds__(kBlock) void k_af_rows_y(IeGeom g, IeBc bc, double dt,
                                                       const double *__restrict__ T,
                                                       const double *__restrict__ u,
                                                       const double *__restrict__ v,
                                                       double *__restrict__ lo, double *__restrict__ di,
                                                       double *__restrict__ up, double *__restrict__ rhs) {
  const int P = g.cy + 2;
  const int64_t nloc = (int64_t)g.rows * P;
  const double dx = g.dx, dy = g.dy;
  for (int64_t k = (int64_t)blockIdx.x * kBlock + threadIdx.x; k < nloc; k += (int64_t)gridDim.x * kBlock) {
    const int ip = g.row0 + (int)(k / P), jp = (int)(k % P);
    const bool xin = ip >= 1 && ip <= g.cx, yin = jp >= 1 && jp <= g.cy;
    double a = 0.0, d = 1.0, c = 0.0, r = 0.0;
    if (xin && yin) {
      const double ky = g.diffusion * g.inv_rp / dy;
      a = (-v[k - 1] / 2.0 - ky) / dy * dt;                             // Dy_m1 * dt
      d = 2.0 * g.diffusion * g.inv_rp / (dy * dy) * dt + 1.0;          // Dy_0 * dt + 1
      c = (v[k + 1] / 2.0 - ky) / dy * dt;                              // Dy_p1 * dt
      const double Tc = T[k], Tw = T[k - P], Te = T[k + P], Ts = T[k - 1], Tn = T[k + 1];
      const double uc = u[k], uw = u[k - P], ue = u[k + P], us = u[k - 1], un = u[k + 1];
      const double vc = v[k], vw = v[k - P], ve = v[k + P], vs = v[k - 1], vn = v[k + 1];
      const double fx_hi = (Tc * uc + Te * ue) / 2.0;
      const double fx_lo = (Tw * uw + Tc * uc) / 2.0;
      const double fy_hi = (Tc * vc + Tn * vn) / 2.0;
      const double fy_lo = (Ts * vs + Tc * vc) / 2.0;
      const double x_deriv = (fx_hi - fx_lo) / dx;
      const double y_deriv = (fy_hi - fy_lo) / dy;
      const double x2_deriv = ((Te - Tc) / dx - (Tc - Tw) / dx) / dx;
      const double y2_deriv = ((Tn - Tc) / dy - (Tc - Ts) / dy) / dy;
      const double lapl = (x2_deriv + y2_deriv) * g.inv_rp;
      const double flux = (-x_deriv - y_deriv) + g.diffusion * lapl;
      const double u_dx = (ue - uw) / (2.0 * dx);
      const double v_dy = (vn - vs) / (2.0 * dy);
      const double u_dy = (un - us) / (2.0 * dy);
      const double v_dx = (ve - vw) / (2.0 * dx);
      const double cross = u_dy + v_dx;
      const double src = g.eckert / g.reynolds * (2.0 * (u_dx * u_dx + v_dy * v_dy) + cross * cross);
      r = (flux + src) * dt;
    } else if (xin && jp == 0) {
      c = 2.0 * bc.weight[KRY_SIDE_BOTTOM] - 1.0;
    } else if (xin && jp == g.cy + 1) {
      a = 2.0 * bc.weight[KRY_SIDE_TOP] - 1.0;
    }
    lo[k] = a;
    di[k] = d;
    up[k] = c;
    rhs[k] = r;
  }
}

// Lines along x (include/time_disc.hpp:466-482 with assemble_system_dx,
// :509-529, and SparseTriDiagSolver::solve, include/solver.hpp:246-273).  One
// lane per mesh column j walks the padded rows; neighbouring lanes touch
// neighbouring addresses, so every access is coalesced without staging.
// rhs holds the y-sweep's result (padded layout); y and ud are scratch.  The
// back substitution adds the interior part of the solution to T and takes the
// step's maximum over the solution entries 0..cx-1 of every line, the index
// range the reference uses (its update reads entries 1..cx).
// Run as a pipeline: with one wavefront per 64 columns and every row step waiting for its own
// loads the sweep takes the memory latency times the number of rows (11.8 ms at 8192^2).
// Here a workgroup of nine wavefronts owns 64 columns and passes blocks of kAfB rows through
// LDS, one barrier per block:
//   wavefronts 1-3, 6-8  loaders, two per input taking every other block: each keeps kAfAhead
//                   of its blocks requested ahead and deposits block t in step t; in the
//                   forward elimination those that fetch u turn it into the coefficients
//                   Dx_m1 dt and Dx_p1 dt on the way;
//   wavefront 0     walks the recurrence that carries the dependence from row to row, one
//                   block behind the loaders: forward the pivots and the normalised upper
//                   diagonal (diag, ud), backward the solution x and T + x;
//   wavefront 5     one more block behind: forward the second recurrence, y, which needs
//                   the pivots but not the other way round (the two IEEE divisions of a row
//                   go to two wavefronts); backward the step's maximum over x;
//   wavefront 4     stores the finished block (y and ud, or T).
// Results replace inputs in place; four LDS slots per array.  The loaders' steady state has
// no conditions (rows past the end are clamped, what they fetch there is never used), so
// their waits name exactly the block to deposit.  Every role reads its block from LDS into
// registers first and writes last, so that LDS latency is paid once per block.  Measured at
// 8192^2: 1.24 ms; with one loader per input 1.40 ms (the
// loaders' 48 requests in flight each then set the pace), so it is the memory system's rate
// for 512-byte rows 65 KB apart that bounds it.
constexpr int kAfB = 12;
constexpr int kAfAhead = 4;
constexpr int kAfSlots = 4;
constexpr int kAfThreads = 576;

struct AfRings {
  double v[4][kAfSlots][kAfB][64];  // array, slot, row of the block, column
};

// One loader wavefront of the pipeline: blocks PHASE, PHASE + NL, ... of one input (rows
// first, first +- 1, ... of one column, clamped to [0, hi_row]), kAfAhead of its blocks
// requested ahead, deposited in the step that bears the block's number.  XF: 0 the values
// as they are, 1 / 2 turned into Dx_m1 dt / Dx_p1 dt (include/time_disc.hpp:509-529).
template <bool BACKWARD, int NL, int PHASE, int XF>
__device__ __forceinline__ void af_loader(double (&slots)[kAfSlots][kAfB][64], const double *base, int64_t P, int first,
                                          int hi_row, int steps, int lane, double kx, double dx, double dt) {
  constexpr int B = kAfB, A = kAfAhead;
  double reg[A][B];
  const auto request = [&](double(&r)[B], int block) {
#pragma unroll
    for (int q = 0; q < B; q++) {
      int row = BACKWARD ? first - (block * B + q) : first + (block * B + q);
      row = row < 0 ? 0 : row;
      row = row > hi_row ? hi_row : row;
      r[q] = base[(int64_t)row * P];
    }
  };
#pragma unroll
  for (int k = 0; k < A; k++) request(reg[k], PHASE + NL * k);
  for (int t0 = 0; t0 < steps; t0 += NL * A) {
#pragma unroll
    for (int kk = 0; kk < NL * A; kk++) {
      if (kk % NL == PHASE) {
        const int t = t0 + kk;
        double(&slot)[B][64] = slots[t % kAfSlots];
        double(&r)[B] = reg[kk / NL];
#pragma unroll
        for (int q = 0; q < B; q++)
          slot[q][lane] = XF == 1 ? (-r[q] / 2.0 - kx) / dx * dt : XF == 2 ? (r[q] / 2.0 - kx) / dx * dt : r[q];
        request(r, t + NL * A);
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(kAfThreads) void k_af_sweep_x2(IeGeom g, IeBc bc, double dt, double *__restrict__ T,
                                                             const double *__restrict__ u,
                                                             const double *__restrict__ rhs,
                                                             double *__restrict__ y, double *__restrict__ ud,
                                                             double *__restrict__ part) {
  extern __shared__ __align__(16) unsigned char af_lds[];
  AfRings &ring = *reinterpret_cast<AfRings *>(af_lds);
  constexpr int B = kAfB, A = kAfAhead, S = kAfSlots;
  const int P = g.cy + 2, cx = g.cx;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;  // interior column
  const bool live = j < g.cy;
  const int64_t col = (live ? j : g.cy - 1) + 1;  // lanes past the last column shadow it and never store
  double m = 0.0, last_nan = -1.0;
  __shared__ double carry[64];  // y of row cx+1, from the wavefront that walks y to the one that walks x

  // ---- forward elimination over the interior rows 1..cx: block b holds rows b B + 1 + q ----
  {
    const int nblk = (cx + B - 1) / B;
    const int steps = (nblk + 3 + 2 * A - 1) / (2 * A) * (2 * A);
    if (wave == 0) {
      const double d0 = 2.0 * g.diffusion * g.inv_rp / (g.dx * g.dx) * dt + 1.0;  // Dx_0 * dt + 1
      // row 0: (., 1, left term)
      double uprev = (2.0 * bc.weight[KRY_SIDE_LEFT] - 1.0) / 1.0;
      if (live) ud[col] = uprev;
      for (int t = 0; t < steps; t++) {
        const int b = t - 1;
        if (b >= 0 && b < nblk) {
          const int s = b % S;
          const int nb = cx - b * B < B ? cx - b * B : B;
          if (nb == B) {
            // all LDS reads of the block first, all writes last: the dependent chain in between
            // then runs out of registers
            double av[B], cv[B];
#pragma unroll
            for (int q = 0; q < B; q++) {
              av[q] = ring.v[0][s][q][lane];
              cv[q] = ring.v[1][s][q][lane];
            }
#pragma unroll
            for (int q = 0; q < B; q++) {
              const double diag = d0 - uprev * av[q];
              uprev = cv[q] / diag;
              av[q] = diag;
              cv[q] = uprev;
            }
#pragma unroll
            for (int q = 0; q < B; q++) {
              ring.v[3][s][q][lane] = av[q];
              ring.v[1][s][q][lane] = cv[q];
            }
          } else {
            for (int q = 0; q < nb; q++) {
              const double diag = d0 - uprev * ring.v[0][s][q][lane];
              uprev = ring.v[1][s][q][lane] / diag;
              ring.v[3][s][q][lane] = diag;
              ring.v[1][s][q][lane] = uprev;
            }
          }
        }
        __syncthreads();
      }
      // row cx+1: (right term, 1, .): its pivot goes to the wavefront that walks y
      {
        const double a = 2.0 * bc.weight[KRY_SIDE_RIGHT] - 1.0;
        const double diag = 1.0 - uprev * a;
        uprev = 0.0 / diag;
        carry[lane] = diag;
        if (live) ud[(int64_t)(cx + 1) * P + col] = uprev;
      }
    } else if (wave <= 3 || wave >= 6) {
      const double kx = g.diffusion * g.inv_rp / g.dx;
      // u in the row before / after the one in hand, and the right-hand side of the row in hand
      if (wave == 1) af_loader<false, 2, 0, 1>(ring.v[0], u + col, P, 0, cx + 1, steps, lane, kx, g.dx, dt);
      else if (wave == 2) af_loader<false, 2, 0, 2>(ring.v[1], u + col, P, 2, cx + 1, steps, lane, kx, g.dx, dt);
      else if (wave == 3) af_loader<false, 2, 0, 0>(ring.v[2], rhs + col, P, 1, cx + 1, steps, lane, kx, g.dx, dt);
      else if (wave == 6) af_loader<false, 2, 1, 1>(ring.v[0], u + col, P, 0, cx + 1, steps, lane, kx, g.dx, dt);
      else if (wave == 7) af_loader<false, 2, 1, 2>(ring.v[1], u + col, P, 2, cx + 1, steps, lane, kx, g.dx, dt);
      else af_loader<false, 2, 1, 0>(ring.v[2], rhs + col, P, 1, cx + 1, steps, lane, kx, g.dx, dt);
    } else if (wave == 5) {
      // row 0: right-hand side 0
      double vprev = 0.0 / 1.0;
      if (live) y[col] = vprev;
      for (int t = 0; t < steps; t++) {
        const int b = t - 2;
        if (b >= 0 && b < nblk) {
          const int s = b % S;
          const int nb = cx - b * B < B ? cx - b * B : B;
          if (nb == B) {
            double av[B], rv[B], dv[B];
#pragma unroll
            for (int q = 0; q < B; q++) {
              av[q] = ring.v[0][s][q][lane];
              rv[q] = ring.v[2][s][q][lane];
              dv[q] = ring.v[3][s][q][lane];
            }
#pragma unroll
            for (int q = 0; q < B; q++) {
              vprev = (rv[q] - vprev * av[q]) / dv[q];
              rv[q] = vprev;
            }
#pragma unroll
            for (int q = 0; q < B; q++) ring.v[2][s][q][lane] = rv[q];
          } else {
            for (int q = 0; q < nb; q++) {
              vprev = (ring.v[2][s][q][lane] - vprev * ring.v[0][s][q][lane]) / ring.v[3][s][q][lane];
              ring.v[2][s][q][lane] = vprev;
            }
          }
        }
        __syncthreads();
      }
      __syncthreads();  // the pivot of row cx+1 is in carry
      {
        const double a = 2.0 * bc.weight[KRY_SIDE_RIGHT] - 1.0;
        vprev = (0.0 - vprev * a) / carry[lane];
        if (live) y[(int64_t)(cx + 1) * P + col] = vprev;
      }
      __syncthreads();
      carry[lane] = vprev;  // where the back substitution starts
    } else {
      for (int t = 0; t < steps; t++) {
        const int b = t - 3;
        if (b >= 0 && b < nblk && live) {
          const int s = b % S;
          const int nb = cx - b * B < B ? cx - b * B : B;
          double *yk = y + (int64_t)(b * B + 1) * P + col, *uk = ud + (int64_t)(b * B + 1) * P + col;
          if (nb == B) {
            double yv[B], uv[B];
#pragma unroll
            for (int q = 0; q < B; q++) {
              yv[q] = ring.v[2][s][q][lane];
              uv[q] = ring.v[1][s][q][lane];
            }
#pragma unroll
            for (int q = 0; q < B; q++) {
              yk[(int64_t)q * P] = yv[q];
              uk[(int64_t)q * P] = uv[q];
            }
          } else {
            for (int q = 0; q < nb; q++) {
              yk[(int64_t)q * P] = ring.v[2][s][q][lane];
              uk[(int64_t)q * P] = ring.v[1][s][q][lane];
            }
          }
        }
        __syncthreads();
      }
    }
    if (wave != 5) {
      __syncthreads();
      __syncthreads();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wavefront's stores have left before the others read y and ud
  }
  __syncthreads();  // y and ud of this workgroup's columns are written, carry holds y of row cx+1

  // ---- back substitution over rows cx..0: block b holds rows cx - (b B + q) ----
  {
    const int nrows = cx + 1;
    const int nblk = (nrows + B - 1) / B;
    const int steps = (nblk + 2 + 2 * A - 1) / (2 * A) * (2 * A);
    if (wave == 0) {
      double next = carry[lane];  // the final value of row cx+1
      for (int t = 0; t < steps; t++) {
        const int b = t - 1;
        if (b >= 0 && b < nblk) {
          const int s = b % S;
          const int nb = nrows - b * B < B ? nrows - b * B : B;
          if (nb == B) {
            double yv[B], uv[B], tv[B];
#pragma unroll
            for (int q = 0; q < B; q++) {
              yv[q] = ring.v[0][s][q][lane];
              uv[q] = ring.v[1][s][q][lane];
              tv[q] = ring.v[2][s][q][lane];
            }
#pragma unroll
            for (int q = 0; q < B; q++) {
              next = yv[q] - next * uv[q];
              yv[q] = next;
              tv[q] = tv[q] + next;
            }
#pragma unroll
            for (int q = 0; q < B; q++) {
              ring.v[0][s][q][lane] = yv[q];
              ring.v[2][s][q][lane] = tv[q];
            }
          } else {
            for (int q = 0; q < nb; q++) {
              next = ring.v[0][s][q][lane] - next * ring.v[1][s][q][lane];
              ring.v[0][s][q][lane] = next;
              ring.v[2][s][q][lane] = ring.v[2][s][q][lane] + next;
            }
          }
        }
        __syncthreads();
      }
    } else if (wave <= 3 || wave >= 6) {
      const int arr = wave <= 3 ? wave - 1 : wave - 6;
      const double *src = (arr == 0 ? y : arr == 1 ? ud : T) + col;
      if (wave <= 3) af_loader<true, 2, 0, 0>(ring.v[arr], src, P, cx, cx + 1, steps, lane, 0.0, 1.0, 1.0);
      else af_loader<true, 2, 1, 0>(ring.v[arr], src, P, cx, cx + 1, steps, lane, 0.0, 1.0, 1.0);
    } else if (wave == 5) {
      // The step's maximum over the solution, one block behind the walker.  Walking backwards,
      // the first NaN met is the line's last one, and the maximum stops there.
      int nan_ip = -1;  // row of the line's last NaN among the entries that count
      for (int t = 0; t < steps; t++) {
        const int b = t - 2;
        if (b >= 0 && b < nblk) {
          const int s = b % S;
          const int nb = nrows - b * B < B ? nrows - b * B : B;
          const auto row = [&](int q, double x) {
            const int ip = cx - (b * B + q);
            const double a = fabs(x);
            const bool counts = live && ip <= cx - 1 && nan_ip < 0, is_nan = a != a;
            m = counts && !is_nan ? max_skip_nan(a, m) : m;
            nan_ip = counts && is_nan ? ip : nan_ip;
          };
          if (nb == B) {
            double xv[B];
#pragma unroll
            for (int q = 0; q < B; q++) xv[q] = ring.v[0][s][q][lane];
#pragma unroll
            for (int q = 0; q < B; q++) row(q, xv[q]);
          } else {
            for (int q = 0; q < nb; q++) row(q, ring.v[0][s][q][lane]);
          }
        }
        __syncthreads();
      }
      if (nan_ip >= 0) last_nan = (double)((int64_t)j * cx + nan_ip);
    } else {
      // T + x of the block before the walker's, rows 1..cx only (a row of T is loaded by
      // wavefront 3 several blocks before it is stored here)
      for (int t = 0; t < steps; t++) {
        const int b = t - 2;
        if (b >= 0 && b < nblk && live) {
          const int s = b % S;
          const int nb = nrows - b * B < B ? nrows - b * B : B;
          const int ip0 = cx - b * B;  // the block's first row; its rows run downwards
          if (nb == B && ip0 - (B - 1) >= 1) {
            double tv[B];
#pragma unroll
            for (int q = 0; q < B; q++) tv[q] = ring.v[2][s][q][lane];
            double *tk = T + (int64_t)ip0 * P + col;
#pragma unroll
            for (int q = 0; q < B; q++) tk[-(int64_t)q * P] = tv[q];
          } else {
            for (int q = 0; q < nb; q++)
              if (ip0 - q >= 1) T[(int64_t)(ip0 - q) * P + col] = ring.v[2][s][q][lane];
          }
        }
        __syncthreads();
      }
    }
  }
  if (wave != 5) return;
  // the workgroup's summary, by the wavefront that took the maximum: its last NaN position,
  // and its maximum behind that position (lanes of later lines have no NaN, or the position
  // would be theirs)
  double lb = last_nan;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) lb = max_skip_nan(__shfl_xor(lb, off, 64), lb);
  if (lb >= 0.0 && (int64_t)j < (int64_t)lb / g.cx) m = 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max_skip_nan(__shfl_down(m, off, 64), m);
  if (lane == 0) {
    part[2 * blockIdx.x] = m;
    part[2 * blockIdx.x + 1] = lb;
  }
}

// The fold of ImplicitEulerApproxFact::timestep (include/time_disc.hpp:478-479; same
// std::max semantics as above) from the workgroups' summaries of k_af_sweep_x2:
// the maximum behind the last NaN position, NaN if that is the last position.
__global__ __launch_bounds__(64) void k_af_max_final(IeGeom g, const double *__restrict__ part, int np,
                                                      double *__restrict__ out) {
  double l = -1.0;
  for (int i = threadIdx.x; i < np; i += 64) l = max_skip_nan(part[2 * i + 1], l);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) l = max_skip_nan(__shfl_xor(l, off, 64), l);
  const int first = l < 0.0 ? 0 : (int)(((int64_t)l / g.cx) / 64);  // workgroup of the line that holds the position
  double m = 0.0;
  for (int i = first + threadIdx.x; i < np; i += 64) m = max_skip_nan(part[2 * i], m);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max_skip_nan(__shfl_down(m, off, 64), m);
  if (threadIdx.x == 0) *out = l == (double)((int64_t)g.cx * g.cy - 1) ? __builtin_nan("") : m;
}

// ---------------------------------------------------------------------------
// Launchers
// ---------------------------------------------------------------------------
// Self-test of band_quotients against the division instruction sequence: `count` groups of four pseudo-random
// numerators over the divisors dx, dy; out[0] += quotients whose bits differ from n / d, out[1] += groups that
// took the five-operation path.  mode 0: numerators within 2^+-40 of 1, all mantissas; mode 1: every bit pattern
// (zeros, subnormals, infinities, NaNs: the guard's side); mode 2: quotients around the limits of the guarded
// range; mode 3: numerators whose quotient by dx is within a few units in the last place of a power of two or
// of a midpoint between two neighbouring numbers (the hard cases of correct rounding).
__device__ __forceinline__ uint64_t splitmix64(uint64_t &x) {
  uint64_t z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__global__ __launch_bounds__(kBlock) void k_selftest_division(double dx, double dy, double rdx, double rdy, uint64_t seed,
                                                               int64_t count, int mode, unsigned long long *out) {
  const bool ok = shared_divisor_in_range(dx) && shared_divisor_in_range(dy);
  const SharedDivisor sdx = shared_divisor(dx, rdx, ok), sdy = shared_divisor(dy, rdy, ok);
  const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  uint64_t st = seed ^ (0xD1B54A32D192ED03ull * (uint64_t)(tid + 1));
  unsigned long long bad = 0, fast = 0;
  for (int64_t it = tid; it < count; it += (int64_t)gridDim.x * kBlock) {
    double n[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const uint64_t r = splitmix64(st);
      const uint64_t frac = r & 0xFFFFFFFFFFFFFull, sign = r >> 63;
      uint64_t e;
      if (mode == 1) {
        e = (r >> 52) & 0x7FF;
      } else if (mode == 2) {
        const int off = (int)((r >> 52) % 41) - 20;  // 2^(+-800 +- 20) times the divisor
        const int de = (int)exponent_field((j == 0 || j == 3) ? dx : dy) - 1023;
        int ee = 1023 + de + (((r >> 60) & 1) ? 800 : -800) + off;
        e = (uint64_t)(ee < 0 ? 0 : (ee > 2047 ? 2047 : ee));
      } else {
        e = 1023 - 40 + (r >> 52) % 81;
      }
      n[j] = __longlong_as_double((long long)((sign << 63) | (e << 52) | frac));
      if (mode == 3) {
        // q0: a power of two or an arbitrary number, moved by up to four units in the last place, or by half a
        // unit (the product with the divisor is rounded, so the quotient lands next to, not on, the hard point)
        const double dd = (j == 0 || j == 3) ? dx : dy;
        uint64_t qb = ((r >> 59) & 1) ? ((sign << 63) | (e << 52)) : ((sign << 63) | (e << 52) | frac);
        qb += (uint64_t)((r >> 55) & 7) - 4;
        const double q0 = __longlong_as_double((long long)qb);
        const double h = ((r >> 58) & 1) ? 0.5 * (__longlong_as_double((long long)(qb + 1)) - q0) : 0.0;
        n[j] = fma(q0, dd, h * dd);
      }
    }
    double q[4];
    if (band_quotients(n[0], n[1], n[2], n[3], sdx, sdy, q[0], q[1], q[2], q[3])) fast++;
    const double t[4] = {n[0] / dx, n[1] / dy, n[2] / dy, n[3] / dx};
#pragma unroll
    for (int j = 0; j < 4; j++) bad += __double_as_longlong(q[j]) != __double_as_longlong(t[j]);
  }
  if (bad) atomicAdd(&out[0], bad);
  if (fast) atomicAdd(&out[1], fast);
}

void launch_selftest_division(hipStream_t st, double dx, double dy, uint64_t seed, int64_t count, int mode,
                              unsigned long long *out) {
  // the reciprocals as the library forms them for a launch: one IEEE division each on the host
  hipLaunchKernelGGL(k_selftest_division, dim3(2048), dim3(kBlock), 0, st, dx, dy, 1.0 / dx, 1.0 / dy, seed, count, mode,
                     out);
}

void launch_fill(hipStream_t st, double *x, double v, int64_t n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_fill, dim3(grid_for(n)), dim3(kBlock), 0, st, x, v, n);
}

// Launch shape of a streaming sweep over `streams` arrays.  Measured on
// MI355X: HBM efficiency is best when the number of concurrently advancing
// stream heads (streams x workgroups) stays near the number of DRAM banks
// (~4096); above 256 workgroups the grid must be a multiple of the 256 CUs
// (384 is slower than both 256 and 512), below that any count works (224 beats
// 256 for 18 streams); wide sweeps want one 16-byte element per lane and
// stream in flight, narrow ones four.
constexpr int sweep_unroll(int streams) { return streams >= 8 ? 1 : (streams >= 5 ? 2 : 4); }

int sweep_grid(int64_t n2, int streams, int unroll) {
  int64_t g = 4096 / streams;
  if (g >= 256) {
    g = g / 256 * 256;  // whole waves of workgroups over the 256 CUs: no CU gets an extra one
    if (g > 1024) g = 1024;
  } else if (streams <= 18) {
    g = g / 8 * 8;  // fewer workgroups than CUs: all resident at once, spread evenly over the 8 XCDs
  } else {
    g = 256;  // very wide sweeps (measured with 34 streams): one workgroup per CU beats fewer
  }
  const int64_t tiles = (n2 + (int64_t)kBlock * unroll - 1) / ((int64_t)kBlock * unroll);
  if (g > tiles) g = tiles;
  if (g < 1) g = 1;
  return (int)g;
}

static int stream_grid(int64_t n2, int unroll, int cap) {
  const int64_t tile = (int64_t)kBlock * unroll;
  int64_t g = (n2 + tile - 1) / tile;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

void launch_nrm2_part(hipStream_t st, const double *x, int64_t n, double *part, int *nparts) {
  const int g = stream_grid(n >> 1, 4, 1024);
  *nparts = g;
  hipLaunchKernelGGL(k_nrm2_part, dim3(g), dim3(kBlock), 0, st, x, n, part);
}

void launch_scale_from(hipStream_t st, const double *src, double *dst, const double *part,
                       int nparts, int64_t n, double *out, int64_t pitch, double *dst2) {
  const int g = stream_grid(n >> 1, 4, 1024);
  hipLaunchKernelGGL(k_scale_from, dim3(g), dim3(kBlock), 0, st, src, dst, part, nparts, n, out, t_stop, pitch, dst2);
}

template <int KC, bool HALO>
static void launch_spmv_stencil_t(hipStream_t st, const Bands &b, int64_t n, int cols, const double *x,
                                  double *y, const double *cbase, int64_t ld, double *part, int *nparts,
                                  const BandGen *gen, int64_t pitch) {
  const bool vec = (cols % 2 == 0) && (n % 2 == 0);
  // 13+ streams with cached x rows: two elements per lane, one workgroup per CU (measured optimum)
  constexpr int U = 2;
  int g = grid_for(n);
  if (vec) {
    const int64_t tiles = ((n >> 1) + (int64_t)kBlock * U - 1) / ((int64_t)kBlock * U);
    // (experimentation knob KRY_SPMV_GRID; with 32 partner vectors and generated bands on an 8192 x 8192 grid:
    // 128 workgroups 4.51 ms, 256 3.50, 512 3.53, 768 3.57, 1024 3.54; one element per lane instead of two: 3.49)
    static const int cap = getenv("KRY_SPMV_GRID") ? atoi(getenv("KRY_SPMV_GRID")) : 256;
    g = (int)(tiles < cap ? tiles : cap);
  }
  if (nparts) *nparts = g;
  if (vec && gen != nullptr)
    hipLaunchKernelGGL((k_spmv_stencil_v2<KC, HALO, U, true>), dim3(g), dim3(kBlock), 0, st, b, n, cols, x, y, cbase, ld, part,
                       t_stop, *gen, pitch);
  else if (vec)
    hipLaunchKernelGGL((k_spmv_stencil_v2<KC, HALO, U, false>), dim3(g), dim3(kBlock), 0, st, b, n, cols, x, y, cbase, ld, part,
                       t_stop, BandGen(), pitch);
  else
    hipLaunchKernelGGL((k_spmv_stencil<KC, HALO>), dim3(g), dim3(kBlock), 0, st, b, n, cols, x, y, cbase, ld, part, t_stop,
                       pitch);
}

// kc -> template argument (0..kMgsBlock)
template <bool HALO, int KC = kMgsBlock>
struct SpmvStencilKc {
  static void go(hipStream_t st, const Bands &b, int64_t n, int cols, const double *x, double *y, int kc,
                 const double *cbase, int64_t ld, double *part, int *nparts, const BandGen *gen, int64_t pitch) {
    if (kc == KC)
      launch_spmv_stencil_t<KC, HALO>(st, b, n, cols, x, y, cbase, ld, part, nparts, gen, pitch);
    else
      SpmvStencilKc<HALO, KC - 1>::go(st, b, n, cols, x, y, kc, cbase, ld, part, nparts, gen, pitch);
  }
};
template <bool HALO>
struct SpmvStencilKc<HALO, -1> {
  static void go(hipStream_t, const Bands &, int64_t, int, const double *, double *, int, const double *, int64_t,
                 double *, int *, const BandGen *, int64_t) {}
};

template <bool HALO>
static void launch_spmv_stencil_h(hipStream_t st, const Bands &b, int64_t n, int cols, const double *x,
                                  double *y, int kc, const double *cbase, int64_t ld, double *part, int *nparts,
                                  const BandGen *gen, int64_t pitch) {
  SpmvStencilKc<HALO>::go(st, b, n, cols, x, y, kc, cbase, ld, part, nparts, gen, pitch);
}

void launch_spmv_stencil(hipStream_t st, const Bands &b, int64_t n, int cols, const double *x,
                         double *y, int kc, const double *cbase, int64_t ld, double *part, int *nparts,
                         bool x_has_halo, const BandGen *gen, int64_t pitch) {
  if (x_has_halo)
    launch_spmv_stencil_h<true>(st, b, n, cols, x, y, kc, cbase, ld, part, nparts, gen, pitch);
  else
    launch_spmv_stencil_h<false>(st, b, n, cols, x, y, kc, cbase, ld, part, nparts, gen, pitch);
}

void launch_reduce_parts(hipStream_t st, const double *part, int nq, int np, double *tot) {
  hipLaunchKernelGGL(k_reduce_parts, dim3(1), dim3(kBlock), 0, st, part, nq, np, tot, t_stop);
}

template <int KC>
static void launch_spmv_csr_t(hipStream_t st, const int64_t *rowptr, const int32_t *col, const double *val,
                              int64_t n, const double *x, double *y, const double *cbase, int64_t ld,
                              double *part, int *nparts, int64_t pitch) {
  const int g = grid_for(n);
  if (nparts) *nparts = g;
  hipLaunchKernelGGL((k_spmv_csr<KC>), dim3(g), dim3(kBlock), 0, st, rowptr, col, val, n, x, y, cbase, ld, part, t_stop,
                     pitch);
}

template <int KC = kMgsBlock>
struct SpmvCsrKc {
  static void go(hipStream_t st, const int64_t *rowptr, const int32_t *col, const double *val, int64_t n,
                 const double *x, double *y, int kc, const double *cbase, int64_t ld, double *part, int *nparts,
                 int64_t pitch) {
    if (kc == KC)
      launch_spmv_csr_t<KC>(st, rowptr, col, val, n, x, y, cbase, ld, part, nparts, pitch);
    else
      SpmvCsrKc<KC - 1>::go(st, rowptr, col, val, n, x, y, kc, cbase, ld, part, nparts, pitch);
  }
};
template <>
struct SpmvCsrKc<-1> {
  static void go(hipStream_t, const int64_t *, const int32_t *, const double *, int64_t, const double *, double *, int,
                 const double *, int64_t, double *, int *, int64_t) {}
};

void launch_spmv_csr(hipStream_t st, const int64_t *rowptr, const int32_t *col, const double *val,
                     int64_t n, const double *x, double *y, int kc, const double *cbase, int64_t ld,
                     double *part, int *nparts, int64_t pitch) {
  SpmvCsrKc<>::go(st, rowptr, col, val, n, x, y, kc, cbase, ld, part, nparts, pitch);
}

struct MgsArgs {
  double *w;
  const double *pbase, *cbase;
  int64_t ld;
  const double *part_in;
  int np_in;
  double *part_out;
  int *np_out;
  double *hout;
  int hstride;
  int64_t n;
  double *G;
  int gld, gp;
  int64_t pitch;
};

template <int KP, int KC, bool NORM>
static void launch_mgs_block_t(hipStream_t st, const MgsArgs &a) {
  constexpr int streams = 2 + KP + KC;
  constexpr int U = sweep_unroll(streams);
  const int g = sweep_grid(a.n >> 1, streams, U);
  *a.np_out = g;
  hipLaunchKernelGGL((k_mgs_block<KP, KC, NORM, U>), dim3(g), dim3(kBlock), 0, st, a.w, a.pbase, a.cbase, a.ld,
                     a.part_in, a.np_in, a.part_out, a.hout, a.hstride, a.n, a.G, a.gld, a.gp, t_stop, a.pitch);
}

// (kp, kc) -> template arguments.  Groups are aligned, so a sweep either joins two
// full groups (kp = kc = K for the block size K in use), or its second group is the
// step's last, shorter one (0 < kc < kp = K), or it is the closing pass of a step
// (kc = 0, any kp <= K).  Only those combinations are instantiated, for the block
// sizes listed in mgs_block_supported().
template <int KP, int KC = KP>
struct MgsKc {
  static void go(hipStream_t st, int kc, const MgsArgs &a) {
    if (kc == KC)
      launch_mgs_block_t<KP, KC, false>(st, a);
    else
      MgsKc<KP, KC - 1>::go(st, kc, a);
  }
};
template <int KP>
struct MgsKc<KP, 0> {
  static void go(hipStream_t, int, const MgsArgs &) {}
};

template <int KP = kMgsBlock>
struct MgsClosing {
  static void go(hipStream_t st, int kp, const MgsArgs &a) {
    if (kp == KP)
      launch_mgs_block_t<KP, 0, true>(st, a);
    else
      MgsClosing<KP - 1>::go(st, kp, a);
  }
};
template <>
struct MgsClosing<0> {
  static void go(hipStream_t, int, const MgsArgs &) {}
};

bool mgs_block_supported(int k) { return k == 1 || k == 2 || k == 3 || k == 4 || k == 8 || k == 16 || k == 32; }

void launch_mgs_block(hipStream_t st, int kp, int kc, double *w, const double *pbase,
                      const double *cbase, int64_t ld, const double *part_in, int np_in,
                      double *part_out, int *np_out, double *hout, int hstride, int64_t n, double *G,
                      int gld, int gp, int64_t pitch) {
  const MgsArgs a{w, pbase, cbase, ld, part_in, np_in, part_out, np_out, hout, hstride, n, G, gld, gp, pitch};
  if (kc == 0) {
    MgsClosing<>::go(st, kp, a);
    return;
  }
  switch (kp) {
    case 1: MgsKc<1>::go(st, kc, a); break;
    case 2: MgsKc<2>::go(st, kc, a); break;
    case 3: MgsKc<3>::go(st, kc, a); break;
    case 4: MgsKc<4>::go(st, kc, a); break;
    case 8: MgsKc<8>::go(st, kc, a); break;
    case 16: MgsKc<16>::go(st, kc, a); break;
    case 32: MgsKc<32>::go(st, kc, a); break;
    default: break;  // excluded by mgs_block_supported() at the API
  }
}

void launch_small_solve(hipStream_t st, const double *H, const double *beta, int m, double *y,
                        double *work) {
  if (m <= kSmallLdsMaxM) {
    const size_t lds = sizeof(double) * (3 * (size_t)m * m + 3 * (size_t)m);
    static std::atomic<uint64_t> configured{0};
    allow_lds(&k_small_solve<true>, sizeof(double) * (3 * (size_t)kSmallLdsMaxM * kSmallLdsMaxM + 3 * kSmallLdsMaxM),
              configured, "LDS size of the reduced solve");
    hipLaunchKernelGGL(k_small_solve<true>, dim3(1), dim3(kBlock), lds, st, H, beta, m, y, work);
    return;
  }
  hipLaunchKernelGGL(k_small_solve<false>, dim3(1), dim3(kBlock), 0, st, H, beta, m, y, work);
}

void launch_combine(hipStream_t st, const double *V, int64_t ld, const double *y, int m, double *x,
                    int64_t n, const int *count, bool accumulate, const double *sigma, int64_t pitch) {
  const int g = grid_for((n + 1) / 2);
  hipLaunchKernelGGL(k_combine, dim3(g), dim3(kBlock), 0, st, V, ld, y, m, x, n, count, accumulate ? 1 : 0, sigma,
                     pitch);
}

void launch_add(hipStream_t st, double *x, const double *w, int64_t n) {
  const int g = stream_grid((n + 1) >> 1, 1, 1024);
  hipLaunchKernelGGL(k_add, dim3(g), dim3(kBlock), 0, st, x, w, n);
}

void launch_sub_nrm2(hipStream_t st, const double *a, const double *b, double *out, int64_t n,
                     double *part, int *nparts) {
  const int g = stream_grid((n + 1) >> 1, 1, 1024);
  *nparts = g;
  hipLaunchKernelGGL(k_sub_nrm2, dim3(g), dim3(kBlock), 0, st, a, b, out, n, part);
}

void launch_ext_begin(hipStream_t st, const double *beta, ExtState *state, double *gv, int m, bool first,
                      double rtol) {
  hipLaunchKernelGGL(k_ext_begin, dim3(1), dim3(64), 0, st, beta, state, gv, m, first ? 1 : 0, rtol);
}

void launch_givens(hipStream_t st, const double *H, int m, int j, double *R, double *cs, double *sn,
                   double *gv, ExtState *state) {
  hipLaunchKernelGGL(k_givens, dim3(1), dim3(64), 0, st, H, m, j, R, cs, sn, gv, state);
}

void launch_ext_solve(hipStream_t st, const double *R, const double *gv, int m, const ExtState *state,
                      double *y) {
  hipLaunchKernelGGL(k_ext_solve, dim3(1), dim3(64), 0, st, R, gv, m, state, y);
}

// lines per wavefront: the largest power of two <= 16 that still leaves about
// two wavefronts per compute unit
static int tdp_lines_per_wave(int64_t nlines) {
  static const int forced = [] {
    const char *e = getenv("KRY_TDP_LPW");
    return e ? atoi(e) : 0;
  }();
  if (forced == 1 || forced == 2 || forced == 4 || forced == 8 || forced == 16) return forced;
  int lpw = 16;
  while (lpw > 1 && (nlines + lpw - 1) / lpw < 512) lpw >>= 1;
  return lpw;
}

template <int LPW>
static void launch_tdp_factor_t(hipStream_t st, const double *lo, const double *di, const double *up,
                                double *dd, double *ud, int64_t nlines, int64_t L) {
  const int g = (int)((nlines + LPW - 1) / LPW);
  hipLaunchKernelGGL(k_tdp_factor<LPW>, dim3(g), dim3(kTdpC), 0, st, lo, di, up, dd, ud, nlines, L);
}

template <int LPW>
static void launch_tdp_apply_t(hipStream_t st, const double *lo, const double *dd, const double *ud,
                               const double *sol, double *out, int64_t nlines, int64_t L) {
  const int g = (int)((nlines + LPW - 1) / LPW);
  hipLaunchKernelGGL(k_tdp_apply<LPW>, dim3(g), dim3(kTdpC), 0, st, lo, dd, ud, sol, out, nlines, L, t_stop);
}

void launch_tdp_factor(hipStream_t st, const double *lo, const double *di, const double *up,
                       double *dd, double *ud, int64_t nlines, int64_t L) {
  switch (tdp_lines_per_wave(nlines)) {
    case 16: launch_tdp_factor_t<16>(st, lo, di, up, dd, ud, nlines, L); break;
    case 8: launch_tdp_factor_t<8>(st, lo, di, up, dd, ud, nlines, L); break;
    case 4: launch_tdp_factor_t<4>(st, lo, di, up, dd, ud, nlines, L); break;
    case 2: launch_tdp_factor_t<2>(st, lo, di, up, dd, ud, nlines, L); break;
    default: launch_tdp_factor_t<1>(st, lo, di, up, dd, ud, nlines, L); break;
  }
}

void launch_tdp_apply(hipStream_t st, const double *lo, const double *dd, const double *ud,
                      const double *sol, double *out, int64_t nlines, int64_t L) {
  static const int pairs = [] {
    const char *e = getenv("KRY_TDP_PAIRS");
    return e ? atoi(e) : 1;
  }();
  const auto aligned16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  if (pairs && tdp_lines_per_wave(nlines) == 16 && L % 2 == 0 && aligned16(lo) && aligned16(dd) && aligned16(ud) &&
      aligned16(sol) && aligned16(out)) {
    const int g = (int)((nlines + 15) / 16);
    hipLaunchKernelGGL((k_tdp_apply2<16, false>), dim3(g), dim3(kTdpC), 0, st, lo, dd, ud, sol, out, nlines, L, t_stop,
                       TdpScale());
    return;
  }
  switch (tdp_lines_per_wave(nlines)) {
    case 16: launch_tdp_apply_t<16>(st, lo, dd, ud, sol, out, nlines, L); break;
    case 8: launch_tdp_apply_t<8>(st, lo, dd, ud, sol, out, nlines, L); break;
    case 4: launch_tdp_apply_t<4>(st, lo, dd, ud, sol, out, nlines, L); break;
    case 2: launch_tdp_apply_t<2>(st, lo, dd, ud, sol, out, nlines, L); break;
    default: launch_tdp_apply_t<1>(st, lo, dd, ud, sol, out, nlines, L); break;
  }
}

bool tdp_scaled_supported(const double *lo, const double *dd, const double *ud, const double *sol, const double *out,
                          const double *vout, int64_t nlines, int64_t L) {
  const auto aligned16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  return tdp_lines_per_wave(nlines) == 16 && L % 2 == 0 && L >= 8 * kTdpC && aligned16(lo) && aligned16(dd) &&
         aligned16(ud) && aligned16(sol) && aligned16(out) && aligned16(vout);
}

void launch_tdp_apply_scaled(hipStream_t st, const double *lo, const double *dd, const double *ud, const double *sol,
                             double *out, int64_t nlines, int64_t L, const TdpScale &sc) {
  const int g = (int)((nlines + 15) / 16);
  hipLaunchKernelGGL((k_tdp_apply2<16, true>), dim3(g), dim3(kTdpC), 0, st, lo, dd, ud, sol, out, nlines, L, t_stop, sc);
}

void launch_tdp_extract_csr(hipStream_t st, const int64_t *rowptr, const int32_t *col,
                            const double *val, int64_t n, double *lo, double *di, double *up) {
  hipLaunchKernelGGL(k_tdp_extract_csr, dim3(grid_for(n)), dim3(kBlock), 0, st, rowptr, col, val, n, lo, di, up);
}

void launch_pack_wrap(hipStream_t st, const IeGeom &g, const double *T, const double *u,
                      const double *v, double *wrap) {
  const int total = 3 * g.cy;
  hipLaunchKernelGGL(k_pack_wrap, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), 0, st, g, T, u, v, wrap);
}

void launch_bc_fill(hipStream_t st, const IeGeom &g, const IeBc &bc, double *T, double *u,
                    double *v, const double *bcvals, const double *wrap) {
  const int total = 3 * 2 * (g.cx + g.cy);
  hipLaunchKernelGGL(k_bc_fill, dim3((total + kBlock - 1) / kBlock), dim3(kBlock), 0, st, g, bc, T, u, v, bcvals, wrap);
}

void launch_offdiag(hipStream_t st, const IeGeom &g, const double *u, const double *v, double *bw,
                    double *bs, double *bc, double *bn, double *be) {
  const int64_t nloc = (int64_t)g.rows * (g.cy + 2);
  hipLaunchKernelGGL(k_offdiag, dim3(grid_for(nloc)), dim3(kBlock), 0, st, g, u, v, bw, bs, bc, bn, be);
}

void launch_assemble(hipStream_t st, const IeGeom &g, const IeBc &bc, double dt, const double *T,
                     const double *u, const double *v, double *bw, double *bs, double *bcn,
                     double *bn, double *be, double *rhs) {
  const int64_t nloc = (int64_t)g.rows * (g.cy + 2);
  hipLaunchKernelGGL(k_assemble, dim3(grid_for(nloc)), dim3(kBlock), 0, st, g, bc, dt, T, u, v, bw, bs, bcn, bn, be, rhs);
}

void launch_update_max(hipStream_t st, const IeGeom &g, double *T, const double *delta,
                       double *part, int *nparts) {
  const int64_t nloc = (int64_t)g.rows * (g.cy + 2);
  const int gr = grid_for(nloc);
  *nparts = gr;
  hipLaunchKernelGGL(k_update_max, dim3(gr), dim3(kBlock), 0, st, g, T, delta, part);
}

void launch_af_rows_y(hipStream_t st, const IeGeom &g, const IeBc &bc, double dt, const double *T,
                      const double *u, const double *v, double *lo, double *di, double *up, double *rhs) {
  const int64_t nloc = (int64_t)g.rows * (g.cy + 2);
  hipLaunchKernelGGL(k_af_rows_y, dim3(grid_for(nloc)), dim3(kBlock), 0, st, g, bc, dt, T, u, v, lo, di, up, rhs);
}

void launch_af_sweep_x(hipStream_t st, const IeGeom &g, const IeBc &bc, double dt, double *T, const double *u,
                       const double *rhs, double *y, double *ud, double *part, int *nparts) {
  const int gr = (g.cy + 63) / 64;
  *nparts = gr;
  static std::atomic<uint64_t> configured{0};
  allow_lds(&k_af_sweep_x2, sizeof(AfRings), configured, "LDS size of the sweep along x");
  hipLaunchKernelGGL(k_af_sweep_x2, dim3(gr), dim3(kAfThreads), sizeof(AfRings), st, g, bc, dt, T, u, rhs, y, ud, part);
}

void launch_max_fold(hipStream_t st, const IeGeom &g, const double *delta, const double *part, int nparts,
                     double *scal, int phases) {
  hipLaunchKernelGGL(k_max_fold, dim3(1), dim3(kFoldBlock), 0, st, g, delta, part, nparts, scal, phases);
}

void launch_af_max_final(hipStream_t st, const IeGeom &g, const double *part, int nparts, double *out) {
  hipLaunchKernelGGL(k_af_max_final, dim3(1), dim3(64), 0, st, g, part, nparts, out);
}

}  // namespace kry
