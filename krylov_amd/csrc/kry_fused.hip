// One-sweep Arnoldi step for the 5-band stencil operator (gfx950).
//
// The plain chain of include/solver.hpp:196-211 reads the basis twice per
// step: once for the dot products <w, v_k>, once for w -= h_k v_k.  The sweep
// below reads it once.  The vectors it produces stay UN-NORMALISED in the basis
// array: slot k holds s_k = sigma_k v_k with sigma_k = ||s_k|| kept in a small
// array (sigma_k = 1 for the vectors the chain stored).  Per step i the sweep
// consumes s_0..s_i and t_i = A s_i that the previous sweep left behind, and in
// ONE pass over the grid
//     w       = t_i / sigma_i                          (= A v_i up to rounding)
//     s_{i+1} = w - c_0 s_0 - ... - c_i s_i,  c_k = h_k / sigma_k
//                                              (modified Gram-Schmidt order)
//     t_{i+1} = A s_{i+1}
// and accumulates <t_{i+1}, s_k>, <s_k, s_{i+1}> (k <= i), <t_{i+1}, s_{i+1}>
// and <s_{i+1}, s_{i+1}>.  From those sums the next step's coefficients follow
// without touching a vector again (k_fused_coeffs): with n = sigma_{i+1},
//     <A v_{i+1}, v_k> = <t, s_k> / (sigma_k n),   <A v_{i+1}, v_{i+1}> = <t, s> / n^2,
//     <v_k, v_{i+1}>   = <s_k, s> / (sigma_k n)    (the carried Gram column),
// and the modified-Gram-Schmidt coefficients are recovered from the dots of the
// un-updated w and the Gram matrix, h_k = d_k - sum_{a<k} h_a g_ak, exactly as
// the blocked sweeps do within one group.  Whoever needs v_k divides by sigma_k
// (kry_gmres_subspace_vector), or divides the coefficient that multiplies it
// (the final combination x = sum y_k v_k).
//
// t = A u needs u at the four neighbours, so the sweep is tiled in two
// dimensions: a workgroup owns a strip of at most 254 columns (thread = column,
// one halo column on either side) and walks down a band of rows, keeping the
// last rows of u in LDS; u is recomputed on the band's two halo rows and the
// strip's two halo columns (about 1.5 % extra reads).  The basis values of a row
// are needed once more one row later (for the dots with t), and wait in LDS.
//
// On several ranks (row-block decomposition) the rows next to a rank's block
// belong to its neighbours: the host first evaluates u_{i+1} on the block's
// first and last row (k_arnoldi_close on those rows), exchanges them, and the
// sweep reads the neighbours' rows from the halo of u_out instead of
// recomputing them (u_halo).
//
// Requirements, checked by the host: 5-band operator whose couplings across
// the ends of a grid row (band s in column 0, band n in the last column) are
// zero, m <= kFusedMaxM.
#include "kry_device.h"

namespace kry {

namespace {

constexpr int kStripMax = kBlock - 2;  // owned columns per workgroup

struct FusedGeom {
  int rows, P;     // padded grid: rows x P, unknown k = row * P + column
  int W, nstrips;  // owned columns per strip, strips per row
  int R, nbands;   // rows per band, bands
};

// Stencil row sum in ascending column order starting from +0.0, as k_spmv_stencil.
__device__ __forceinline__ double row_sum(double cw, double cs, double cc, double cn, double ce, double xw,
                                          double xs, double xc, double xn, double xe) {
  double r = 0.0;
  r = r + cw * xw;
  r = r + cs * xs;
  r = r + cc * xc;
  r = r + cn * xn;
  r = r + ce * xe;
  return r;
}

// Value of x in lane `lane` (a compile-time constant after unrolling), as a wave-uniform scalar.
__device__ __forceinline__ double lane_value(double x, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), lane);
  return __hiloint2double(hi, lo);
}

// NV: step number; s_0..s_{NV-1} are read from V, s_NV from u_in (it is slot NV of the
// same array; a separate argument keeps the register arrays NV long).
// u: the new vector s_{NV+1}.
// part layout (q * gridDim.x + workgroup): q <= NV: <t, s_q>;  NV+1: <t, u>;
// NV+2+a, a <= NV: <s_a, u>;  2 NV + 3: <u, u>.
template <int NV>
__global__ __launch_bounds__(kBlock) void k_arnoldi_fused(FusedGeom geo, Bands b,
                                                           const double *__restrict__ V, int64_t ld,
                                                           const double *__restrict__ t_in,
                                                           const double *__restrict__ u_in,
                                                           const double *__restrict__ coef,
                                                           double *__restrict__ u_out,
                                                           double *__restrict__ t_out,
                                                           double *__restrict__ part,
                                                           const double *__restrict__ u_halo,
                                                           const int *__restrict__ stop) {
  if (stopped(stop)) return;
  constexpr int NQ = 2 * NV + 4;
  extern __shared__ double lds[];
  double *keep = lds;                              // (NV + 1) x kBlock: v_k and v_NV of the previous row
  double *ring = keep + (size_t)(NV + 1) * kBlock;  // 4 x kBlock: the last rows of u
  double *red = ring + 4 * kBlock;                 // NQ x 4
  const int t = threadIdx.x;
  const int strip = blockIdx.x % geo.nstrips, band = blockIdx.x / geo.nstrips;
  const int cs = strip * geo.W;
  const int ce = cs + geo.W < geo.P ? cs + geo.W : geo.P;
  const int c = cs - 1 + t;
  const bool colvalid = c >= 0 && c < geo.P && t <= (ce - cs) + 1;
  const bool colowned = t >= 1 && c < ce;
  const int rs = band * geo.R;
  const int re = rs + geo.R < geo.rows ? rs + geo.R : geo.rows;
  const int P = geo.P;

  // the coefficients live in the lanes of one register pair (lane k: h_k)
  static_assert(NV + 1 <= 64, "one coefficient per lane");
  const double hpack = coef[(t & 63) <= NV ? (t & 63) : 0];
  const double nrm = coef[kFusedNormSlot];

  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = 0.0;
#pragma unroll
  for (int k = 0; k <= NV; k++) keep[k * kBlock + t] = 0.0;

  // Loads run one row ahead of the arithmetic: the values of row rr + 1 are requested
  // before row rr is worked on.
  double vn[NV > 0 ? NV : 1];
  double tin_n = 0.0, uin_n = 0.0;
  double bn[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  auto fetch = [&](int rr) {
    const bool rowvalid = rr >= 0 && rr < geo.rows;
    const int64_t idx = (int64_t)rr * P + c;
    tin_n = 0.0;
    uin_n = 0.0;
#pragma unroll
    for (int k = 0; k < NV; k++) vn[k] = 0.0;
    if (rowvalid && colvalid) {
      tin_n = __builtin_nontemporal_load(&t_in[idx]);
      uin_n = __builtin_nontemporal_load(&u_in[idx]);
#pragma unroll
      for (int k = 0; k < NV; k++) vn[k] = __builtin_nontemporal_load(&V[(int64_t)k * ld + idx]);
    } else if (u_halo != nullptr && !rowvalid && c >= 0 && c < P) {
      // a neighbouring rank's boundary row of the new vector itself (zeros beyond the
      // mesh); u_halo is u_out, read only outside the rows this kernel writes.  The read
      // sits with the other requests of the row: a load inside the arithmetic below would
      // make the compiler drain all outstanding requests there.
      uin_n = u_halo[idx];
    }
    // operator coefficients of row rr - 1, whose application completes one round later
    if (rr - 1 >= rs && rr - 1 < re && colowned) {
      const int64_t idq = idx - P;
      bn[0] = __builtin_nontemporal_load(&b.w[idq]);
      bn[1] = __builtin_nontemporal_load(&b.s[idq]);
      bn[2] = __builtin_nontemporal_load(&b.c[idq]);
      bn[3] = __builtin_nontemporal_load(&b.n[idq]);
      bn[4] = __builtin_nontemporal_load(&b.e[idq]);
    }
  };
  fetch(rs - 1);

  for (int rr = rs - 1; rr <= re; rr++) {
    const bool rowvalid = rr >= 0 && rr < geo.rows;
    const int64_t idx = (int64_t)rr * P + c;
    const int q = rr - 1;  // the row whose operator application completes in this round
    const bool emit = q >= rs && colowned;
    const int64_t idq = idx - P;
    double vk[NV > 0 ? NV : 1];
#pragma unroll
    for (int k = 0; k < NV; k++) vk[k] = vn[k];
    const double tin = tin_n, uin = uin_n;
    const double cw = bn[0], cso = bn[1], cc = bn[2], cn = bn[3], cea = bn[4];
    if (rr < re) fetch(rr + 1);
    __builtin_amdgcn_sched_barrier(0);  // keep the requests ahead of the arithmetic below
    double vi = 0.0, u = 0.0;
    if (rowvalid && colvalid) {
      vi = uin;
      u = tin / nrm;
#pragma unroll
      for (int k = 0; k < NV; k++) u = u - lane_value(hpack, k) * vk[k];
      u = u - lane_value(hpack, NV) * vi;
    } else {
      u = uin;  // zero, or a neighbouring rank's row of the new vector
    }
    if (rr >= rs && rr < re && colowned) {
      __builtin_nontemporal_store(u, &u_out[idx]);
#pragma unroll
      for (int k = 0; k < NV; k++) acc[NV + 2 + k] = fma(vk[k], u, acc[NV + 2 + k]);
      acc[2 * NV + 2] = fma(vi, u, acc[2 * NV + 2]);
      acc[2 * NV + 3] = fma(u, u, acc[2 * NV + 3]);
    }
    const int slot = (rr - rs + 1) & 3;
    ring[slot * kBlock + t] = u;
    __syncthreads();
    if (emit) {
      const double *rq = ring + ((slot + 3) & 3) * kBlock;   // row q
      const double *rm = ring + ((slot + 2) & 3) * kBlock;   // row q - 1
      const double *rp = ring + slot * kBlock;               // row q + 1
      // the first and last column of a grid row have no left / right neighbour in that row
      const double xs = c > 0 ? rq[t - 1] : 0.0;
      const double xn = c + 1 < P ? rq[t + 1] : 0.0;
      const double uq = rq[t];
      const double tv = row_sum(cw, cso, cc, cn, cea, rm[t], xs, uq, xn, rp[t]);
      __builtin_nontemporal_store(tv, &t_out[idq]);
#pragma unroll
      for (int k = 0; k <= NV; k++) acc[k] = fma(keep[k * kBlock + t], tv, acc[k]);
      acc[NV + 1] = fma(uq, tv, acc[NV + 1]);
    }
#pragma unroll
    for (int k = 0; k < NV; k++) keep[k * kBlock + t] = vk[k];
    keep[NV * kBlock + t] = vi;
  }
  reduce_store_multi<NQ>(acc, red, part);
}

// Last step of a cycle: s_{i+1} and <s_{i+1}, s_{i+1}> only (no operator
// application follows).  part[workgroup].
template <int NV>
__global__ __launch_bounds__(kBlock) void k_arnoldi_close(const double *__restrict__ V, int64_t ld,
                                                           const double *__restrict__ t_in,
                                                           const double *__restrict__ u_in,
                                                           const double *__restrict__ coef,
                                                           double *__restrict__ u_out, int64_t n,
                                                           double *__restrict__ part,
                                                           const int *__restrict__ stop) {
  if (stopped(stop)) return;
  __shared__ double s[kBlock / 64];
  double h[NV + 1];
#pragma unroll
  for (int k = 0; k <= NV; k++) h[k] = coef[k];
  const double nrm = coef[kFusedNormSlot];
  double a = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    double vk[NV > 0 ? NV : 1];
    const double tin = __builtin_nontemporal_load(&t_in[i]);
    const double uin = __builtin_nontemporal_load(&u_in[i]);
#pragma unroll
    for (int k = 0; k < NV; k++) vk[k] = __builtin_nontemporal_load(&V[(int64_t)k * ld + i]);
    double u = tin / nrm;
#pragma unroll
    for (int k = 0; k < NV; k++) u = u - h[k] * vk[k];
    u = u - h[NV] * uin;
    __builtin_nontemporal_store(u, &u_out[i]);
    a = fma(u, u, a);
  }
  const double tot = block_sum(a, s);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// tot[q] = sum of part[q * np .. q * np + np): one workgroup per quantity.
__global__ __launch_bounds__(kBlock) void k_reduce_parts_wide(const double *__restrict__ part, int np,
                                                               double *__restrict__ tot,
                                                               const int *__restrict__ stop) {
  if (stopped(stop)) return;
  __shared__ double s[kBlock / 64];
  const int q = blockIdx.x;
  double a = 0.0;
  for (int i = threadIdx.x; i < np; i += kBlock) a += part[(size_t)q * np + i];
  const double r = block_sum(a, s);
  if (threadIdx.x == 0) tot[q] = r;
}

// Coefficients of step i from the totals of the sweep that produced s_i and
// t_i = A s_i (layout of k_arnoldi_fused<i-1>), or, with boot, from the totals
// of the operator kernel applied to the normalised v_i (i + 1 dots, then the i
// Gram entries of v_i; sigma_0..sigma_i := 1).  Stores sigma_i = ||s_i||, column i
// of G and of H, H[i][i-1] = sigma_i, coef[k] = h_k / sigma_k (k <= i) and
// coef[kFusedNormSlot] = sigma_i.
__global__ __launch_bounds__(64) void k_fused_coeffs(const double *__restrict__ tot, int i, int boot,
                                                      double *__restrict__ G, int gld,
                                                      double *__restrict__ H, int m,
                                                      double *__restrict__ coef,
                                                      double *__restrict__ sigma,
                                                      const int *__restrict__ stop) {
  if (stopped(stop)) return;
  __shared__ double gs[kFusedMaxM * kFusedMaxM];
  const int lane = threadIdx.x;
  const double nrm = boot ? 1.0 : sqrt(tot[2 * i + 1]);
  const double sk = (boot || lane >= i) ? (lane == i ? nrm : 1.0) : sigma[lane];  // sigma of "my" vector
  // earlier Gram columns
  for (int e = lane; e < i * i; e += 64) {
    const int a = e / i, k = e % i;
    gs[a * kFusedMaxM + k] = a < k ? G[(size_t)a * gld + k] : 0.0;
  }
  if (lane < i) {
    const double g = boot ? tot[i + 1 + lane] : tot[i + 1 + lane] / sk / nrm;
    gs[lane * kFusedMaxM + i] = g;
    G[(size_t)lane * gld + i] = g;
  }
  __syncthreads();
  double v = 0.0;
  if (lane <= i) {
    if (boot)
      v = tot[lane];
    else
      v = tot[lane] / sk / nrm;  // lane == i: <t, s_i> / sigma_i^2
  }
  for (int a = 0; a < i; a++) {
    const double ha = __shfl(v, a, 64);
    if (lane > a && lane <= i) v = v - ha * gs[a * kFusedMaxM + lane];
  }
  if (lane <= i) {
    coef[lane] = v / sk;
    H[(size_t)lane * m + i] = v;
    if (boot || lane == i) sigma[lane] = sk;
  }
  if (lane == 0) {
    coef[kFusedNormSlot] = nrm;
    if (!boot && i > 0) H[(size_t)i * m + (i - 1)] = nrm;
  }
}

// End of a cycle: sigma_m = ||s_m|| from the closing kernel's total, H[m][m-1] = sigma_m.
__global__ __launch_bounds__(64) void k_fused_finish(const double *__restrict__ tot, int m, double *__restrict__ H,
                                                      double *__restrict__ sigma, const int *__restrict__ stop) {
  if (stopped(stop)) return;
  if (threadIdx.x == 0) {
    const double nrm = sqrt(tot[0]);
    sigma[m] = nrm;
    H[(size_t)m * m + (m - 1)] = nrm;
  }
}

// dst = src / sigma[k] (true division, as normalize() in include/matrix.hpp:54-60)
__global__ __launch_bounds__(kBlock) void k_unscale(const double *__restrict__ src, const double *__restrict__ sigma,
                                                     int k, double *__restrict__ dst, int64_t n) {
  const double sg = sigma[k];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    dst[i] = src[i] / sg;
}

FusedGeom fused_geom(int rows, int P) {
  FusedGeom g;
  g.rows = rows;
  g.P = P;
  g.nstrips = (P + kStripMax - 1) / kStripMax;
  g.W = (P + g.nstrips - 1) / g.nstrips;
  // about four workgroups per compute unit, bands of at least 32 rows
  int nb = (4 * 256) / g.nstrips;
  if (nb < 1) nb = 1;
  while (nb > 1 && (rows + nb - 1) / nb < 32) nb--;
  g.R = (rows + nb - 1) / nb;
  g.nbands = (rows + g.R - 1) / g.R;
  return g;
}

template <int NV>
void launch_fused_t(hipStream_t st, const FusedGeom &geo, const Bands &b, const double *V, int64_t ld,
                    const double *t_in, const double *u_in, const double *coef, double *u_out,
                    double *t_out, double *part, const double *u_halo) {
  constexpr size_t lds = sizeof(double) * ((size_t)(NV + 1) * kBlock + 4 * kBlock + (2 * NV + 4) * (kBlock / 64));
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_arnoldi_fused<NV>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    configured = true;
  }
  hipLaunchKernelGGL(k_arnoldi_fused<NV>, dim3(geo.nstrips * geo.nbands), dim3(kBlock), lds, st, geo, b, V, ld,
                     t_in, u_in, coef, u_out, t_out, part, u_halo, launch_predicate_current());
}

template <int NV>
void launch_close_t(hipStream_t st, const double *V, int64_t ld, const double *t_in, const double *u_in,
                    const double *coef, double *u_out, int64_t n, double *part, int grid) {
  hipLaunchKernelGGL(k_arnoldi_close<NV>, dim3(grid), dim3(kBlock), 0, st, V, ld, t_in, u_in, coef, u_out, n, part,
                     launch_predicate_current());
}

template <int NV = kFusedMaxM - 1>
struct FusedTable {
  static void sweep(int nv, hipStream_t st, const FusedGeom &geo, const Bands &b, const double *V, int64_t ld,
                    const double *t_in, const double *u_in, const double *coef, double *u_out,
                    double *t_out, double *part, const double *u_halo) {
    if (nv == NV)
      launch_fused_t<NV>(st, geo, b, V, ld, t_in, u_in, coef, u_out, t_out, part, u_halo);
    else
      FusedTable<NV - 1>::sweep(nv, st, geo, b, V, ld, t_in, u_in, coef, u_out, t_out, part, u_halo);
  }
  static void close(int nv, hipStream_t st, const double *V, int64_t ld, const double *t_in, const double *u_in,
                    const double *coef, double *u_out, int64_t n, double *part, int grid) {
    if (nv == NV)
      launch_close_t<NV>(st, V, ld, t_in, u_in, coef, u_out, n, part, grid);
    else
      FusedTable<NV - 1>::close(nv, st, V, ld, t_in, u_in, coef, u_out, n, part, grid);
  }
};

template <>
struct FusedTable<-1> {
  static void sweep(int, hipStream_t, const FusedGeom &, const Bands &, const double *, int64_t, const double *,
                    const double *, const double *, double *, double *, double *, const double *) {}
  static void close(int, hipStream_t, const double *, int64_t, const double *, const double *, const double *,
                    double *, int64_t, double *, int) {}
};

}  // namespace

int fused_parts(int rows, int cols) {
  const FusedGeom g = fused_geom(rows, cols);
  return g.nstrips * g.nbands;
}

void launch_arnoldi_fused(hipStream_t st, int nv, int rows, int cols, const Bands &b, const double *V, int64_t ld,
                          const double *t_in, const double *coef, double *t_out, double *part, int *nparts,
                          bool has_halo) {
  const FusedGeom g = fused_geom(rows, cols);
  double *u_out = const_cast<double *>(V) + (int64_t)(nv + 1) * ld;
  FusedTable<>::sweep(nv, st, g, b, V, ld, t_in, V + (int64_t)nv * ld, coef, u_out, t_out, part,
                      has_halo ? u_out : nullptr);
  *nparts = g.nstrips * g.nbands;
}

void launch_arnoldi_close(hipStream_t st, int nv, const double *V, int64_t ld, const double *t_in, const double *coef,
                          int64_t n, double *part, int *nparts) {
  int grid = grid_for(n);
  if (grid > 1024) grid = 1024;
  FusedTable<>::close(nv, st, V, ld, t_in, V + (int64_t)nv * ld, coef, const_cast<double *>(V) + (int64_t)(nv + 1) * ld,
                      n, part, grid);
  *nparts = grid;
}

void launch_reduce_parts_wide(hipStream_t st, const double *part, int nq, int np, double *tot) {
  hipLaunchKernelGGL(k_reduce_parts_wide, dim3(nq), dim3(kBlock), 0, st, part, np, tot, launch_predicate_current());
}

void launch_fused_coeffs(hipStream_t st, const double *tot, int i, bool boot, double *G, int gld, double *H,
                         int m, double *coef, double *sigma) {
  hipLaunchKernelGGL(k_fused_coeffs, dim3(1), dim3(64), 0, st, tot, i, boot ? 1 : 0, G, gld, H, m, coef, sigma,
                     launch_predicate_current());
}

void launch_fused_finish(hipStream_t st, const double *tot, int m, double *H, double *sigma) {
  hipLaunchKernelGGL(k_fused_finish, dim3(1), dim3(64), 0, st, tot, m, H, sigma, launch_predicate_current());
}

void launch_unscale(hipStream_t st, const double *src, const double *sigma, int k, double *dst, int64_t n) {
  hipLaunchKernelGGL(k_unscale, dim3(grid_for(n)), dim3(kBlock), 0, st, src, sigma, k, dst, n);
}

}  // namespace kry
