// One-sweep Arnoldi step for the 5-band stencil operator (gfx950).
//
// The plain chain of include/solver.hpp:196-211 reads the basis twice per
// step: once for the dot products <w, v_k>, once for w -= h_k v_k.  The sweep
// below reads it once.  The vectors it produces stay UN-NORMALISED in the basis
// array: slot k holds s_k = sigma_k v_k with sigma_k = ||s_k|| kept in a small
// array (sigma_k = 1 for the vectors the chain stored).  Per step i the sweep
// consumes s_0..s_i and in ONE pass over the grid
//     w       = (A s_i) / sigma_i                      (= A v_i up to rounding)
//     s_{i+1} = w - c_0 s_0 - ... - c_i s_i,  c_k = h_k / sigma_k
//                                              (modified Gram-Schmidt order)
//     t_{i+1} = A s_{i+1}
// and accumulates <t_{i+1}, s_k>, <s_k, s_{i+1}> (k <= i), <t_{i+1}, s_{i+1}>,
// <s_{i+1}, s_{i+1}> and <t_{i+1}, t_{i+1}>.  From those sums the next step's coefficients follow
// without touching a vector again (k_fused_coeffs): with n = sigma_{i+1},
//     <A v_{i+1}, v_k> = <t, s_k> / (sigma_k n),   <A v_{i+1}, v_{i+1}> = <t, s> / n^2,
//     <v_k, v_{i+1}>   = <s_k, s> / (sigma_k n)    (the carried Gram column),
// and the modified-Gram-Schmidt coefficients are recovered from the dots of the
// un-updated w and the Gram matrix, h_k = d_k - sum_{a<k} h_a g_ak, exactly as
// the blocked sweeps do within one group.  Whoever needs v_k divides by sigma_k
// (kry_gmres_subspace_vector), or divides the coefficient that multiplies it
// (the final combination x = sum y_k v_k).
// The last step of a cycle only needs the NORM of its vector (H(m, m-1); the vector
// itself is not part of x): modified Gram-Schmidt takes ||w||^2 down by h_k^2 per
// link, so that norm is sqrt(||A v_{m-1}||^2 - sum_k h_k^2) with ||A v_{m-1}||^2 =
// <t, t> / sigma_{m-1}^2 from the last sweep.  k_fused_coeffs forms it and raises a
// device flag that turns the closing kernel (k_arnoldi_close on the whole vector) into
// a no-op, unless the subtraction cancels (less than 1/64 of ||A v||^2 left).
//
// t = A u needs u at the four neighbours, so the sweep is tiled in two
// dimensions: a workgroup owns a strip of at most 252 columns (thread = column,
// two halo columns on either side) and walks down a band of rows, keeping the
// last rows of s_i and of the new vector in LDS; the new vector is recomputed on
// the band's two halo rows and the strip's inner halo columns (about 1.5 %
// extra reads).  The basis values of a row
// are needed once more one row later (for the dots with t), and wait in LDS.
//
// On several ranks (row-block decomposition) the rows next to a rank's block
// belong to its neighbours: the host first evaluates s_{i+1} on the block's
// first and last row (k_arnoldi_close on those rows), exchanges them, and the
// sweep reads the neighbours' rows from the halo of slot i + 1 instead of
// recomputing them (u_halo); the neighbours' rows of s_i are in the halo of
// slot i since the step before.
//
// Two kernels do this: k_arnoldi_fused (256 threads per workgroup, every step) and, from
// kSplitFrom earlier vectors on, k_arnoldi_split (512 threads whose two halves split the
// basis vectors, so that two wavefronts fit a SIMD where the first kernel runs with one);
// both produce the same bits.
//
// Three things about how the sweeps use the memory system and the compute units:
//  * the rows of the new vector wait in an LDS ring and are stored when a device-wide clock passes a
//    multiple of a period (FusedGeom::tick), so that all workgroups write at the same moments;
//  * the basis vectors and velocity snapshots may be stored strip by strip (FusedGeom::sm, StripMap):
//    a workgroup then reads and writes whole aligned 2 KB lines that follow each other in memory;
//  * from kGroupedFrom earlier vectors on the update is evaluated in two groups, one per half of
//    the 512-thread workgroup, so that the halves work on it at the same time.
//
// Requirements, checked by the host: 5-band operator whose couplings across
// the ends of a grid row (band s in column 0, band n in the last column) are
// zero, m <= kFusedMaxM.
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "kry_device.h"

namespace kry {

namespace {

constexpr int kStripMax = kBlock - 4;  // owned columns per workgroup (two halo columns on either side)
// From this many earlier vectors on the update s_{NV+1} = w - sum_k c_k s_k is evaluated in two groups,
//   (w - c_0 s_0 - ... - c_{NA-1} s_{NA-1}) - (c_NA s_NA + ... + c_NV s_NV),   NA = (NV + 1) / 2,
// by every kernel that forms the new vector (both sweeps and the closing kernel: same bits everywhere), so that
// the two halves of a 512-thread workgroup can work on their groups at the same time.
constexpr int kGroupedFrom = 12;
// Rows of the new vector stored together = rows of it kept in LDS (a power of two, at least 4).
// Longer bursts are better as long as the LDS they take does not cost a resident workgroup
// (4 workgroups per compute unit are resident up to 5 earlier vectors, 3 up to 10, 2 up to 21
// by registers; a workgroup needs (nv + 1 + rows + 4) x 2 KB + the reduction scratch of the
// 160 KB); the steps were measured per instantiation on an 8192 x 8192 grid.
constexpr int store_rows(int nv) { return nv <= 7 ? 8 : (nv <= 18 ? 16 : (nv <= 21 ? 8 : (nv <= 29 ? 32 : 16))); }

// Whether the sweep of step nv also accumulates <t, t> (the last of its partial sums), from which the norm of a
// cycle's last vector follows without the closing kernel.  The 512-thread kernel for 41 earlier vectors has no
// register left for one more accumulator (it would spill), so a cycle whose last sweep is that one (m = 43)
// keeps the closing kernel.  A property of the step, not of the kernel variant that serves it.
constexpr bool sweep_has_tt(int nv) { return nv != 41; }

struct FusedGeom {
  int rows, P;     // padded grid: rows x P, unknown k = row * P + column
  int W, nstrips;  // owned columns per strip, strips per row
  int R, nbands;   // rows per band, bands
  // A launch may cover a subset of the bands: workgroup b of the launch works on band
  // band_first + (b / nstrips) * band_step.  Its partial sums go to the slot it would have in
  // a launch over all bands, so the subsets' launches fill one array between them.
  int band_first = 0, band_step = 1;
  // Stores on a common clock: period in ticks of the device-wide 100 MHz counter (0: off).  Every wavefront
  // stores the rows of the new vector it holds when that counter passes a multiple of the period (or its
  // ring is full), so that the writes of all workgroups reach the memory controllers in a few dense batches
  // instead of a steady trickle among the reads.
  int tick = 0;
  // Strip-major vectors (sm != 0): the basis vectors and the velocity snapshots are stored strip by strip.
  // Strip s is a block of rows + 2 rows: one grid row in front, the rows, one behind; a row holds the kBlock
  // columns the strip's workgroup touches (lane t at position t: two columns of the strip to the left, the W
  // owned ones, two of the strip to the right), so a workgroup reads and writes whole aligned 2 KB lines as it
  // walks down its band.  The edge columns are stored twice (by their owner and by its neighbour).
  // The velocity snapshots are single vectors (rows kBlock apart, blocks of SBg values); in the basis the rows
  // of all vectors are interleaved (StripMap): rows `pitch` = slots x kBlock apart, vectors kBlock apart (the
  // `ld` the kernel is given), blocks of SB values.
  int sm = 0, pitch = 0, SBg = 0;
  int64_t SB = 0;
};

// The counter all compute units share (s_memrealtime), low 32 bits: only differences are used.
__device__ __forceinline__ uint32_t common_clock() { return (uint32_t)wall_clock64(); }
// First multiple of the period after `now`, in the counter's low 32 bits.
__device__ __forceinline__ uint32_t first_deadline(int period) {
  const uint64_t now = wall_clock64();
  return (uint32_t)((now / (uint64_t)period + 1) * (uint64_t)period);
}

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

// The value itself, behind an empty asm statement.  The compiler turns a select on the
// result of an expensive operation (a division) into a branch around the operation,
// and the waits for the operation's operands then sit inside that branch: on the path
// around it the requests count as still in flight, and every later wait is made to
// allow for them.  A select on a value of unknown origin stays a select.
__device__ __forceinline__ double opaque(double x) {
  asm volatile("" : "+v"(x));
  return x;
}

// The diagonal entry all interior cells share.
__device__ __forceinline__ double gdiag_of(const BandGen &gen) { return *gen.cdiag; }

// Coefficients of step i from the totals of the sweep that produced s_i and
// t_i = A s_i (layout of k_arnoldi_fused<i-1>), or, with boot, from the totals
// of the operator kernel applied to the normalised v_i (i + 1 dots, then the i
// Gram entries of v_i; sigma_0..sigma_i := 1).  Stores sigma_i = ||s_i||, column i
// of G and of H, H[i][i-1] = sigma_i, coef[k] = h_k / sigma_k (k <= i) and
// coef[kFusedNormSlot] = sigma_i.  Called by every thread of a workgroup; the first
// wavefront does the recurrence.  gs: i (i + 1) doubles of LDS.
constexpr double kNormFromSumsMin = 1.0 / 64.0;  // share of ||A v||^2 the norm from sums has to keep

// write_global: store the results (G, H, coef, sigma, the norm flag); cf != nullptr (LDS, 65 doubles): also leave
// coef[k] in cf[k] and sigma_i in cf[64] for the calling workgroup.  A sweep that evaluates its own coefficients
// calls this in every workgroup (same totals, same expression tree, hence the same bits everywhere) and lets its
// first workgroup store.  Contains a barrier: to be called by all threads of the workgroup.
__device__ __forceinline__ void fused_coeffs_body(const double *tot, int i, int boot, double *G, int gld, double *H,
                                                  int m, double *coef, double *sigma, double *gs, int *norm_flag,
                                                  bool write_global = true, double *cf = nullptr) {
  const int lane = threadIdx.x;  // only lanes of the first wavefront own a vector
  const int gp = i + 1;          // row pitch of gs
  const double nrm = boot ? 1.0 : sqrt(tot[2 * i + 1]);
  const double sk = (boot || lane >= i) ? (lane == i ? nrm : 1.0) : sigma[lane < kFusedCoef ? lane : 0];  // sigma of "my" vector
  // earlier Gram columns
  for (int e = threadIdx.x; e < i * i; e += blockDim.x) {
    const int a = e / i, k = e % i;
    gs[a * gp + k] = a < k ? G[(size_t)a * gld + k] : 0.0;
  }
  if (lane < i) {
    const double g = boot ? tot[i + 1 + lane] : tot[i + 1 + lane] / sk / nrm;
    gs[lane * gp + i] = g;
    if (write_global) G[(size_t)lane * gld + i] = g;
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  double v = 0.0;
  if (lane <= i) {
    if (boot)
      v = tot[lane];
    else
      v = tot[lane] / sk / nrm;  // lane == i: <t, s_i> / sigma_i^2
  }
  for (int a = 0; a < i; a++) {
    const double ha = __shfl(v, a, 64);
    if (lane > a && lane <= i) v = v - ha * gs[a * gp + lane];
  }
  if (lane <= i) {
    const double c = v / sk;
    if (cf != nullptr) cf[lane] = c;
    if (write_global) {
      coef[lane] = c;
      H[(size_t)lane * m + i] = v;
      if (boot || lane == i) sigma[lane] = sk;
    }
  }
  if (lane == 0) {
    if (cf != nullptr) cf[64] = nrm;
    if (write_global) {
      coef[kFusedNormSlot] = nrm;
      if (!boot && i > 0) H[(size_t)i * m + (i - 1)] = nrm;
    }
  }
  if (norm_flag != nullptr) {
    // ||A v_i - sum_k h_k v_k||^2 = ||A v_i||^2 - sum_k h_k^2;  <t, t> is the last total of the previous sweep
    double hs = lane <= i ? v * v : 0.0;
    hs = __shfl(wave_sum(hs), 0, 64);
    const double ww = boot ? 0.0 : tot[2 * i + 2] / nrm / nrm;
    const double r2 = ww - hs;
    const bool ok = !boot && r2 >= kNormFromSumsMin * ww && ww > 0.0 && ww <= 1.7e308;  // false for anything not finite
    if (lane == 0 && write_global) {
      *norm_flag = ok ? 1 : 0;
      if (ok) {
        const double nn = sqrt(r2);
        sigma[i + 1] = nn;
        H[(size_t)(i + 1) * m + i] = nn;
      }
    }
  }
}

// NV: step number; s_0..s_{NV-1} are read from V, s_NV from u_in (it is slot NV of the
// same array; a separate argument keeps the register arrays NV long).
// u: the new vector s_{NV+1}.  t_NV = A s_NV is not stored by the previous sweep but
// recomputed here from s_NV, whose last rows live in a second LDS ring: that is one
// vector read and one written less per sweep, at the price of a second halo column on
// either side of the strip and a second halo row at either end of the band (s_NV only).
// s_halo: the rows in front of the first / behind the last grid row exist in memory
// (a neighbouring rank's rows of s_NV, zeros beyond the mesh).
// part layout (q * nstrips * nbands + band * nstrips + strip): q <= NV: <t, s_q>;  NV+1: <t, u>;
// NV+2+a, a <= NV: <s_a, u>;  2 NV + 3: <u, u>;  2 NV + 4: <t, t>   (t = A u).
// GEN: the band entries of interior cells are evaluated from the recipe `gen` (velocity
// snapshots; the operations of k_offdiag, so the same bits) and those of ghost rows and
// columns follow from their position (a 1 towards the interior, the diagonal 2w - 1 of
// their side): two streamed arrays instead of five, no stored band is read.
//
// Memory requests form a rolling window.  Requesting a whole row, working on the
// previous one and then waiting for everything lets the number of requests in flight
// drop to zero once per row; with one wavefront per SIMD (more than 256 registers, from
// 22 earlier vectors on) nothing else covers that gap.  So every request is issued
// unconditionally -- from an address clamped into the grid, the value of a lane outside
// it is discarded by a select when it is used -- and in exactly the order in which the
// next round consumes the values: each value is consumed and the request for its
// successor issued in the same place, and the wait for any one of them leaves all
// younger requests in flight.
// S: register sets for the requested values; the requests run S - 1 rows ahead of the row being worked on.
template <int NV, bool GEN, int S = 2>
__global__ __launch_bounds__(kBlock) void k_arnoldi_fused(FusedGeom geo, Bands b, BandGen gen,
                                                          const double *__restrict__ V, int64_t ld,
                                                          const double *__restrict__ u_in,
                                                          const double *__restrict__ coef,
                                                          double *__restrict__ u_out,
                                                          double *__restrict__ part,
                                                          const double *__restrict__ u_halo, int s_halo,
                                                          const int *__restrict__ stop, CoefIn cfin) {
  if (stopped(stop)) return;
  constexpr bool TT = sweep_has_tt(NV);
  constexpr int NQ = 2 * NV + 4 + (TT ? 1 : 0);
  extern __shared__ double lds[];
  double *keep = lds;                               // (NV + 1) x kBlock: s_k of the previous row
  constexpr int kStoreRows = store_rows(NV), kRing = kStoreRows;
  double *ring = keep + (size_t)(NV + 1) * kBlock;  // kRing x kBlock: the last rows of u
  double *sring = ring + kRing * kBlock;            // 4 x kBlock: rows rr-1 .. rr+2 of s_NV
  double *red = sring + 4 * kBlock;                 // NQ x 4
  const int t = threadIdx.x;
  const int tm = t > 0 ? t - 1 : 0, tp = t + 1 < kBlock ? t + 1 : kBlock - 1;
  const int strip = blockIdx.x % geo.nstrips, band = geo.band_first + (blockIdx.x / geo.nstrips) * geo.band_step;
  const int cs = strip * geo.W;
  const int ce = cs + geo.W < geo.P ? cs + geo.W : geo.P;
  const int wown = ce - cs;
  const int P = geo.P, rows = geo.rows;
  const int c = cs - 2 + t;
  const bool cin = c >= 0 && c < P;
  const bool scol = cin && t <= wown + 3;            // s_NV is needed here
  const bool ucol = cin && t >= 1 && t <= wown + 2;  // the new vector is evaluated here
  const bool colowned = t >= 2 && t <= wown + 1;     // ... and stored / summed here
  const int rs = band * geo.R;
  const int re = rs + geo.R < rows ? rs + geo.R : rows;
  // clamped coordinates for the unconditional requests
  const int cc = c < 0 ? 0 : (c >= P ? P - 1 : c);
  const int ccm = cc > 0 ? cc - 1 : 0, ccp = cc + 1 < P ? cc + 1 : P - 1;
  auto rowc = [&](int r) -> int64_t { return r < 0 ? 0 : (r >= rows ? rows - 1 : r); };  // a grid row
  auto rowh = [&](int r) -> int64_t { return r < -1 ? -1 : (r > rows ? rows : r); };     // ... or a halo row
  // position of this lane's column in a row of a basis vector (off) and of the lane's column and its two
  // neighbours in a row of a velocity snapshot (goff, goffm, goffp), and the distances between rows
  const bool sm = geo.sm != 0;
  const int pitch = sm ? geo.pitch : P, gpitch = sm ? kBlock : P;
  int off = cc, soff = cc, goff = cc, goffm = ccm, goffp = ccp;
  if (sm) {
    // the basis pointers move to this strip's block (its size does not fit 32 bits times the strip number)
    V += (int64_t)strip * geo.SB;
    u_in += (int64_t)strip * geo.SB;
    u_out += (int64_t)strip * geo.SB;
    const int base = geo.pitch;                           // row 0 of the block
    const int lt = t <= wown + 3 ? t : wown + 3;          // lanes beyond the strip repeat its last position
    off = base + lt;
    const int gbase = strip * geo.SBg + kBlock;           // row 0 of this strip's block of a snapshot
    goff = gbase + lt;
    goffm = gbase + (lt > 0 ? lt - 1 : 0);
    goffp = gbase + (lt + 1 < kBlock ? lt + 1 : kBlock - 1);
    // the outermost columns of s_NV are the only values a workgroup needs that it did not store
    // itself one step earlier: they come from their owners' blocks
    soff = off;
    if (t == 0 && strip > 0) soff = base - (int)geo.SB + geo.W;
    if (t == wown + 3 && strip + 1 < geo.nstrips) soff = base + (int)geo.SB + 3;
  }
  auto s_addr = [&](int r) -> const double * { return &u_in[(s_halo ? rowh(r) : rowc(r)) * pitch + soff]; };
  auto s_ok = [&](int r) -> bool {
    return scol && ((r >= 0 && r < rows) || (s_halo != 0 && (r == -1 || r == rows)));
  };

  static_assert(NV + 1 <= 64, "one coefficient per lane");
  double hpack, nrm;
  if (cfin.tot != nullptr) {
    // this step's coefficients from the previous sweep's totals; the scratch lives where the rows of the basis
    // vectors will wait (not in use before the loop)
    double *gs = keep, *cf = keep + NV * (NV + 1);
    fused_coeffs_body(cfin.tot, NV, 0, cfin.G, cfin.gld, cfin.H, cfin.m, const_cast<double *>(coef), cfin.sigma, gs, nullptr,
                      blockIdx.x == 0, cf);
    __syncthreads();
    hpack = cf[(t & 63) <= NV ? (t & 63) : 0];
    nrm = cf[64];
  } else {
    hpack = coef[(t & 63) <= NV ? (t & 63) : 0];
    nrm = coef[kFusedNormSlot];
  }

  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = 0.0;

  {
    const double d0 = __builtin_nontemporal_load(s_addr(rs - 2));
    const double d1 = __builtin_nontemporal_load(s_addr(rs - 1));
    const double d2 = __builtin_nontemporal_load(s_addr(rs));
    // sring slot of row r: (r - rs + 2) & 3
    sring[0 * kBlock + t] = s_ok(rs - 2) ? d0 : 0.0;
    sring[1 * kBlock + t] = s_ok(rs - 1) ? d1 : 0.0;
    sring[2 * kBlock + t] = s_ok(rs) ? d2 : 0.0;
  }

  const double gdiag = GEN ? gdiag_of(gen) : 0.0;
  // the mesh widths with their reciprocals, for the quotients in the band entries (see div_shared_fast)
  const SharedDivisor sdx = shared_divisor(gen.dx, gen.rdx, gen.div_ok != 0), sdy = shared_divisor(gen.dy, gen.rdy, gen.div_ok != 0);
  // The values requested for a round.  S sets take turns (the loop below is unrolled S
  // times): a value requested into the registers a later round reads needs no copy at
  // the end of the round, which would have to wait for it.  The round that works on row
  // rr requests row rr + S - 1 into the set the round before it has finished with.
  static_assert(S >= 2 && S <= 4, "two to four register sets");
  constexpr int D = S - 1;
  struct Req {
    double v[NV > 0 ? NV : 1];  // s_0 .. s_{NV-1}
    double b[5];                // five band entries, or (GEN) the two velocities next to the cell in its row
    double gu;                  // GEN: the velocity one row further
    double s, uh;               // s_NV two rows further; a neighbouring rank's row of the new vector
  };
  Req qa, qb, qc, qd;
  double gu0 = 0.0, gu1 = 0.0;  // GEN: velocity of rows rr - 1 and rr
  const bool yin = GEN && c >= 1 && c <= gen.cy;
  auto issue_op = [&](Req &n, int r) __attribute__((always_inline)) {  // what the operator coefficients of row r are made of
    const int64_t ro = rowc(r) * P;
    if (GEN) {
      const int64_t rg = rowc(r) * gpitch;
      n.b[1] = gen.v[rg + goffm];
      n.b[3] = gen.v[rg + goffp];
      n.gu = __builtin_nontemporal_load(&gen.u[rowh(r + 1) * gpitch + goff]);
    } else {
      n.b[0] = __builtin_nontemporal_load(&b.w[ro + cc]);
      n.b[1] = __builtin_nontemporal_load(&b.s[ro + cc]);
      n.b[2] = __builtin_nontemporal_load(&b.c[ro + cc]);
      n.b[3] = __builtin_nontemporal_load(&b.n[ro + cc]);
      n.b[4] = __builtin_nontemporal_load(&b.e[ro + cc]);
    }
  };
  auto issue_tail = [&](Req &n, int r) __attribute__((always_inline)) {
    // a neighbouring rank's boundary row of the new vector itself (u_halo is u_out, read only
    // outside the rows this kernel writes); without one, a request that costs nothing
    n.uh = *(u_halo != nullptr ? &u_halo[rowh(r) * pitch + off] : coef);
    n.s = __builtin_nontemporal_load(s_addr(r + 2));
  };
  if (GEN) {
    gu0 = __builtin_nontemporal_load(&gen.u[rowh(rs - 2) * gpitch + goff]);
    gu1 = __builtin_nontemporal_load(&gen.u[rowh(rs - 1) * gpitch + goff]);
  }
  auto issue_all = [&](Req &n, int r) __attribute__((always_inline)) {  // everything the round of row r consumes
    issue_op(n, r);
    const int64_t i0 = rowc(r) * pitch + off;
#pragma unroll
    for (int k = 0; k < NV; k++) n.v[k] = __builtin_nontemporal_load(&V[(int64_t)k * ld + i0]);
    issue_tail(n, r);
  };
  issue_all(qa, rs - 1);
  if (D > 1) issue_all(qb, rs);
  if (D > 2) issue_all(qc, rs + 1);
  double bp[5] = {0.0, 0.0, 0.0, 0.0, 0.0};  // operator coefficients of the previous row
  int p0 = rs;  // first row of the new vector not stored yet (the same in every lane of a wavefront)
  uint32_t deadline = geo.tick > 0 ? first_deadline(geo.tick) : 0u;
  // The first row's values are awaited here (vmcnt(0)): a request still in flight on entry
  // to the loop would make every wait inside it allow for that request's position in the
  // queue as well, which drains the window once per round.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  // one round: consumes the values of set o (row rr), requests those of row rr + D into set n
  auto round = [&](const int rr, Req &o, Req &n) __attribute__((always_inline)) {
    const bool rowvalid = rr >= 0 && rr < rows;
    const bool active = rowvalid && ucol;
    const int64_t idxn = rowc(rr + D) * pitch + off;
    const int q = rr - 1;  // the row whose operator application (to the new vector) completes in this round
    const bool emit = q >= rs && colowned;
    // operator coefficients of row rr
    double bc[5];
    if (GEN) {
      const double gum = gu0, gup = o.gu;
      gu0 = gu1;
      gu1 = gup;
      issue_op(n, rr + D);
      const int ip = gen.row0 + rr;
      const bool xin = ip >= 1 && ip <= gen.cx;
      // SecondOrderCentered::Dx_m1 / Dy_m1 / Dy_p1 / Dx_p1, include/space_disc.hpp:83-113, as k_offdiag forms them
      double iw, is, in, ie;
      band_quotients(-gum / 2.0 - gen.kx, -o.b[1] / 2.0 - gen.ky, o.b[3] / 2.0 - gen.ky, gup / 2.0 - gen.kx, sdx, sdy, iw, is,
                     in, ie);
      const bool inner = xin && yin;
      // ghost rows as k_assemble leaves them (include/time_disc.hpp:304-337)
      bc[0] = inner ? iw : ((yin && ip == gen.cx + 1) ? 1.0 : 0.0);
      bc[1] = inner ? is : ((xin && c == gen.cy + 1) ? 1.0 : 0.0);
      bc[3] = inner ? in : ((xin && c == 0) ? 1.0 : 0.0);
      bc[4] = inner ? ie : ((yin && ip == 0) ? 1.0 : 0.0);
      bc[2] = inner ? gdiag : (xin ? (c == 0 ? gen.dg_bottom : gen.dg_top) : (yin ? gen.dg_left : 1.0));
    } else {
#pragma unroll
      for (int j = 0; j < 5; j++) bc[j] = o.b[j];
      issue_op(n, rr + D);
    }
    const int sl = (rr - rs + 2) & 3;  // sring slot of row rr
    const double *s0 = sring + sl * kBlock;
    const double vi = s0[t];
    // the first and last column of a grid row have no left / right neighbour in that row
    const double xs = c > 0 ? s0[tm] : 0.0;
    const double xn = c + 1 < P ? s0[tp] : 0.0;
    const double tin = row_sum(bc[0], bc[1], bc[2], bc[3], bc[4], sring[((sl + 3) & 3) * kBlock + t], xs, vi, xn,
                               sring[((sl + 1) & 3) * kBlock + t]);
    double u = tin / nrm;
    constexpr int NG = NV >= kGroupedFrom ? (NV + 1) / 2 : NV + 1;  // vectors subtracted one by one
    double pg = 0.0;                                                // the second group's sum
#pragma unroll
    for (int k = 0; k < NV; k++) {
      n.v[k] = __builtin_nontemporal_load(&V[(int64_t)k * ld + idxn]);
      const double term = lane_value(hpack, k) * o.v[k];
      if (k < NG)
        u = u - term;
      else
        pg = k == NG ? term : pg + term;
    }
    if (NG <= NV) {
      const double term = lane_value(hpack, NV) * vi;
      pg = NG == NV ? term : pg + term;
      u = opaque(u - pg);
    } else {
      u = opaque(u - lane_value(hpack, NV) * vi);
    }
    issue_tail(n, rr + D);
    // outside the columns and rows evaluated here: zero, or a neighbouring rank's row of the new vector
    u = active ? u : ((u_halo != nullptr && !rowvalid && cin) ? o.uh : 0.0);
    const int slot = (rr - rs + 1) & (kRing - 1);
    ring[slot * kBlock + t] = u;
    // The new vector is stored kStoreRows rows at a time, from the ring (a thread reads back
    // what it wrote itself): the memory side handles a few long bursts of writes among
    // the reads better than a steady trickle.
    bool flush = rr - p0 + 1 == kStoreRows || rr == re - 1;  // the ring is full, or the band ends
    if (geo.tick > 0 && (int32_t)(common_clock() - deadline) >= 0) {
      flush = true;
      do deadline += (uint32_t)geo.tick;
      while ((int32_t)(common_clock() - deadline) >= 0);
    }
    if (rr >= rs && rr < re && flush) {
      // strip-major: the workgroup stores the edge columns it evaluated as well (their owners store the same bits)
      if (sm ? (t >= 1 && t <= wown + 2) : colowned) {
        const int so = sm ? off : c;
#pragma unroll
        for (int j = 0; j < kStoreRows; j++) {
          const int r = p0 + j;
          if (r <= rr) __builtin_nontemporal_store(ring[((r - rs + 1) & (kRing - 1)) * kBlock + t], &u_out[(int64_t)r * pitch + so]);
        }
      }
      p0 = rr + 1;
    }
    sring[((sl + 2) & 3) * kBlock + t] = s_ok(rr + 2) ? o.s : 0.0;  // row rr + 2 of s_NV
    __syncthreads();
    if (emit) {
      const double *rq = ring + ((slot + kRing - 1) & (kRing - 1)) * kBlock;  // row q
      const double *rm = ring + ((slot + kRing - 2) & (kRing - 1)) * kBlock;  // row q - 1
      const double *rp = ring + slot * kBlock;                                // row q + 1
      const double us = c > 0 ? rq[tm] : 0.0;
      const double un = c + 1 < P ? rq[tp] : 0.0;
      const double uq = rq[t];
      const double tv = row_sum(bp[0], bp[1], bp[2], bp[3], bp[4], rm[t], us, uq, un, rp[t]);
      // both sets of dot products of row q from the parked values: the registers that
      // received s_k are then free as soon as the update has used them
#pragma unroll
      for (int k = 0; k <= NV; k++) {
        const double sk = keep[k * kBlock + t];
        acc[k] = fma(sk, tv, acc[k]);
        acc[NV + 2 + k] = fma(sk, uq, acc[NV + 2 + k]);
      }
      acc[NV + 1] = fma(uq, tv, acc[NV + 1]);
      acc[2 * NV + 3] = fma(uq, uq, acc[2 * NV + 3]);
      if (TT) acc[NQ - 1] = fma(tv, tv, acc[NQ - 1]);
    }
#pragma unroll
    for (int k = 0; k < NV; k++) keep[k * kBlock + t] = o.v[k];
    keep[NV * kBlock + t] = vi;
#pragma unroll
    for (int j = 0; j < 5; j++) bp[j] = bc[j];
  };
  for (int rr = rs - 1; rr <= re; rr += S) {
    if (S == 2) {
      round(rr, qa, qb);
      if (rr + 1 > re) break;
      round(rr + 1, qb, qa);
    } else if (S == 3) {
      round(rr, qa, qc);
      if (rr + 1 > re) break;
      round(rr + 1, qb, qa);
      if (rr + 2 > re) break;
      round(rr + 2, qc, qb);
    } else {
      round(rr, qa, qd);
      if (rr + 1 > re) break;
      round(rr + 1, qb, qa);
      if (rr + 2 > re) break;
      round(rr + 2, qc, qb);
      if (rr + 3 > re) break;
      round(rr + 3, qd, qc);
    }
  }
  reduce_store_multi<NQ>(acc, red, part, band * geo.nstrips + strip, geo.nstrips * geo.nbands);
}

// The same sweep by a workgroup of 512 threads whose two halves share the columns of the
// strip and split the basis vectors: half 0 evaluates the operator coefficients, w and the
// first NA updates, half 1 the remaining updates (in the same order, continuing from half
// 0's value, so every bit of the new vector is the same), stores the result and applies the
// operator to it; each half requests, parks and accumulates the dot products of its own
// vectors.  Per lane that is half the registers, so two wavefronts fit a SIMD where the
// kernel above runs with one, and the arithmetic of one overlaps the waiting of the other.
constexpr int kSplitBlock = 2 * kBlock;
constexpr int kUniformHalfUpto = 34;
constexpr int kSplitFrom = 18;  // first step served by it (measured per instantiation on an 8192 x 8192 grid)
// rows stored together: as many as the LDS left beside the parked vectors holds
constexpr int split_store_rows(int nv) { return nv <= 40 ? 32 : 16; }
// Vectors per half whose parked values wait in registers instead of LDS, as many as the registers
// hold: fewer LDS round trips in the dot-product phase (2 % up to 34 earlier vectors), and the LDS
// for 32-row bursts up to 42 earlier vectors.
constexpr int split_park_regs(int nv) { return nv < 18 ? 0 : (nv <= 24 ? 9 : (nv <= 30 ? 8 : (nv <= 33 ? 6 : (nv <= 40 ? 4 : 0)))); }

template <int NV, bool GEN>
__global__ __launch_bounds__(kSplitBlock) void k_arnoldi_split(FusedGeom geo, Bands b, BandGen gen,
                                                                const double *__restrict__ V, int64_t ld,
                                                                const double *__restrict__ u_in,
                                                                const double *__restrict__ coef,
                                                                double *__restrict__ u_out,
                                                                double *__restrict__ part,
                                                                const double *__restrict__ u_halo, int s_halo,
                                                                const int *__restrict__ stop, CoefIn cfin) {
  if (stopped(stop)) return;
  constexpr bool TT = sweep_has_tt(NV);
  constexpr int NQ = 2 * NV + 4 + (TT ? 1 : 0);
  constexpr int NA = (NV + 1) / 2, NB = NV - NA;  // vectors of half 0: [0, NA), of half 1: [NA, NV) and s_NV
  static_assert(NB >= 1, "both halves need a vector");
  constexpr int kStoreRows = split_store_rows(NV), kRing = kStoreRows;
  constexpr int PK = split_park_regs(NV);  // vectors per half parked in registers
  static_assert(PK <= NB, "a half parks its own vectors");
  extern __shared__ double lds[];
  // s_k of the previous row: (NV + 1 - 2 PK) x kBlock -- half 0's vectors PK .. NA-1, half 1's
  // vectors PK .. NB-1, then s_NV
  double *keep = lds;
  double *ring = keep + (size_t)(NV + 1 - 2 * PK) * kBlock;  // kRing x kBlock: the last rows of u
  double *sring = ring + kRing * kBlock;            // 4 x kBlock: rows rr-1 .. rr+2 of s_NV
  double *xa = sring + 4 * kBlock;                  // kBlock: half 0's value of the update chain
  double *bcs = xa + kBlock;                        // 5 x kBlock: operator coefficients of row rr
  double *tvs = bcs + 5 * kBlock;                   // kBlock: the operator applied to the new vector, row rr - 1
  double *red = tvs + kBlock;                       // NQ x 4
  // which half: the same in every lane of a wavefront; up to kUniformHalfUpto earlier vectors it is read as
  // such (from the first lane), so that the halves' loops, their counters, predicates and row addresses live in
  // scalar registers (1-2 % faster; beyond, the register allocation of that variant needs scratch)
  const int half = NV <= kUniformHalfUpto ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 8) : (int)(threadIdx.x >> 8);
  const int t = threadIdx.x & (kBlock - 1);
  const int lane = t & 63, wv = t >> 6;
  const int tm = t > 0 ? t - 1 : 0, tp = t + 1 < kBlock ? t + 1 : kBlock - 1;
  const int strip = blockIdx.x % geo.nstrips, band = geo.band_first + (blockIdx.x / geo.nstrips) * geo.band_step;
  const int cs = strip * geo.W;
  const int ce = cs + geo.W < geo.P ? cs + geo.W : geo.P;
  const int wown = ce - cs;
  const int P = geo.P, rows = geo.rows;
  const int c = cs - 2 + t;
  const bool cin = c >= 0 && c < P;
  const bool scol = cin && t <= wown + 3;
  const bool ucol = cin && t >= 1 && t <= wown + 2;
  const bool colowned = t >= 2 && t <= wown + 1;
  const int rs = band * geo.R;
  const int re = rs + geo.R < rows ? rs + geo.R : rows;
  const int cc = c < 0 ? 0 : (c >= P ? P - 1 : c);
  const int ccm = cc > 0 ? cc - 1 : 0, ccp = cc + 1 < P ? cc + 1 : P - 1;
  auto rowc = [&](int r) -> int64_t { return r < 0 ? 0 : (r >= rows ? rows - 1 : r); };
  auto rowh = [&](int r) -> int64_t { return r < -1 ? -1 : (r > rows ? rows : r); };
  // positions within a row of a basis vector / velocity snapshot, as in k_arnoldi_fused
  const bool sm = geo.sm != 0;
  const int pitch = sm ? geo.pitch : P, gpitch = sm ? kBlock : P;
  int off = cc, soff = cc, goff = cc, goffm = ccm, goffp = ccp;
  if (sm) {
    V += (int64_t)strip * geo.SB;
    u_in += (int64_t)strip * geo.SB;
    u_out += (int64_t)strip * geo.SB;
    const int base = geo.pitch;
    const int lt = t <= wown + 3 ? t : wown + 3;
    off = base + lt;
    const int gbase = strip * geo.SBg + kBlock;
    goff = gbase + lt;
    goffm = gbase + (lt > 0 ? lt - 1 : 0);
    goffp = gbase + (lt + 1 < kBlock ? lt + 1 : kBlock - 1);
    soff = off;
    if (t == 0 && strip > 0) soff = base - (int)geo.SB + geo.W;
    if (t == wown + 3 && strip + 1 < geo.nstrips) soff = base + (int)geo.SB + 3;
  }
  auto s_addr = [&](int r) -> const double * { return &u_in[(s_halo ? rowh(r) : rowc(r)) * pitch + soff]; };
  auto s_ok = [&](int r) -> bool {
    return scol && ((r >= 0 && r < rows) || (s_halo != 0 && (r == -1 || r == rows)));
  };
  const int kb = half ? NA : 0;  // first vector of this half
  // lane k: the coefficient of this half's k-th vector (half 1: and that of s_NV in lane NB)
  double hpack, nrm;
  if (cfin.tot != nullptr) {
    // this step's coefficients from the previous sweep's totals (scratch: the start of the LDS, not in use yet)
    double *gs = lds, *cf = lds + NV * (NV + 1);
    fused_coeffs_body(cfin.tot, NV, 0, cfin.G, cfin.gld, cfin.H, cfin.m, const_cast<double *>(coef), cfin.sigma, gs, nullptr,
                      blockIdx.x == 0, cf);
    __syncthreads();
    hpack = cf[kb + lane <= NV ? kb + lane : 0];
    nrm = cf[64];
    __syncthreads();  // sring, which the lines below fill, may lie inside that scratch
  } else {
    hpack = coef[kb + lane <= NV ? kb + lane : 0];
    nrm = coef[kFusedNormSlot];
  }
  const double *Vh = V + (int64_t)kb * ld;

  if (half == 0) {
    const double d0 = __builtin_nontemporal_load(s_addr(rs - 2));
    const double d1 = __builtin_nontemporal_load(s_addr(rs - 1));
    const double d2 = __builtin_nontemporal_load(s_addr(rs));
    sring[0 * kBlock + t] = s_ok(rs - 2) ? d0 : 0.0;
    sring[1 * kBlock + t] = s_ok(rs - 1) ? d1 : 0.0;
    sring[2 * kBlock + t] = s_ok(rs) ? d2 : 0.0;
  }
  // accumulators of this half: <t, s_k> and <s_k, u> of its vectors (index NB of half 1: s_NV),
  // half 0 also <t, u> and <u, u>, half 1 <t, t>
  double aT[NA + 1], aS[NA + 1], aTU = 0.0, aUU = 0.0, aTT = 0.0;
#pragma unroll
  for (int k = 0; k <= NA; k++) aT[k] = aS[k] = 0.0;

  struct Req {
    double v[NA];
    double b[5];
    double gu;
    double s, uh;
  };
  Req qa, qb;
  const double gdiag = GEN ? gdiag_of(gen) : 0.0;
  // the mesh widths with their reciprocals, for the quotients in the band entries (see div_shared_fast)
  const SharedDivisor sdx = shared_divisor(gen.dx, gen.rdx, gen.div_ok != 0), sdy = shared_divisor(gen.dy, gen.rdy, gen.div_ok != 0);
  double gu0 = 0.0, gu1 = 0.0;
  const bool yin = GEN && c >= 1 && c <= gen.cy;
  auto issue_op = [&](Req &n, int r) __attribute__((always_inline)) {
    const int64_t ro = rowc(r) * P;
    if (GEN) {
      const int64_t rg = rowc(r) * gpitch;
      n.b[1] = gen.v[rg + goffm];
      n.b[3] = gen.v[rg + goffp];
      n.gu = __builtin_nontemporal_load(&gen.u[rowh(r + 1) * gpitch + goff]);
    } else {
      n.b[0] = __builtin_nontemporal_load(&b.w[ro + cc]);
      n.b[1] = __builtin_nontemporal_load(&b.s[ro + cc]);
      n.b[2] = __builtin_nontemporal_load(&b.c[ro + cc]);
      n.b[3] = __builtin_nontemporal_load(&b.n[ro + cc]);
      n.b[4] = __builtin_nontemporal_load(&b.e[ro + cc]);
    }
  };
  // this half's vectors of one row
  auto issue_row = [&](Req &n, int64_t idx) __attribute__((always_inline)) {
#pragma unroll
    for (int k = 0; k < NB; k++) n.v[k] = __builtin_nontemporal_load(&Vh[(int64_t)k * ld + idx]);
    if (NA > NB && half == 0) n.v[NA - 1] = __builtin_nontemporal_load(&Vh[(int64_t)(NA - 1) * ld + idx]);
  };
  {
    const int64_t i0 = rowc(rs - 1) * pitch + off;
    if (half == 0) {
      if (GEN) {
        gu0 = __builtin_nontemporal_load(&gen.u[rowh(rs - 2) * gpitch + goff]);
        gu1 = __builtin_nontemporal_load(&gen.u[rowh(rs - 1) * gpitch + goff]);
      }
      issue_op(qa, rs - 1);
      issue_row(qa, i0);
      qa.s = __builtin_nontemporal_load(s_addr(rs + 1));
    } else {
      issue_row(qa, i0);
      qa.uh = *(u_halo != nullptr ? &u_halo[rowh(rs - 1) * pitch + off] : coef);
    }
  }
  double bp[5] = {0.0, 0.0, 0.0, 0.0, 0.0};  // half 1: operator coefficients of the previous row
  int p0 = rs;  // half 1: first row of the new vector not stored yet (the same in every lane of a wavefront)
  uint32_t deadline = geo.tick > 0 ? first_deadline(geo.tick) : 0u;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();

  // dot products of row q = rr - 1 from the parked values, then this row's values are parked
  double pk[PK > 0 ? PK : 1];  // this half's first PK vectors of the previous row
#pragma unroll
  for (int k = 0; k < (PK > 0 ? PK : 1); k++) pk[k] = 0.0;
  // LDS row of this half's k-th vector (k >= PK)
  const int krow = (half ? NA - PK : 0) - PK;
  constexpr int kRowS = NV - 2 * PK;  // ... and of s_NV
  auto emit_park = [&](const int rr, Req &o, const double vi) __attribute__((always_inline)) {
    const int q = rr - 1;
    const int slot = (rr - rs + 1) & (kRing - 1);
    if (q >= rs && colowned) {
      const double tv = tvs[t];
      const double uq = ring[((slot + kRing - 1) & (kRing - 1)) * kBlock + t];
#pragma unroll
      for (int k = 0; k < NB; k++) {
        const double sk = k < PK ? pk[k] : keep[(krow + k) * kBlock + t];
        aT[k] = fma(sk, tv, aT[k]);
        aS[k] = fma(sk, uq, aS[k]);
      }
      if (half == 0) {
        if (NA > NB) {
          const double sk = keep[(krow + NA - 1) * kBlock + t];
          aT[NA - 1] = fma(sk, tv, aT[NA - 1]);
          aS[NA - 1] = fma(sk, uq, aS[NA - 1]);
        }
        aTU = fma(uq, tv, aTU);
        aUU = fma(uq, uq, aUU);
      } else {
        const double sk = keep[kRowS * kBlock + t];
        aT[NB] = fma(sk, tv, aT[NB]);
        aS[NB] = fma(sk, uq, aS[NB]);
        if (TT) aTT = fma(tv, tv, aTT);
      }
    }
#pragma unroll
    for (int k = 0; k < NB; k++) {
      if (k < PK)
        pk[k] = o.v[k];
      else
        keep[(krow + k) * kBlock + t] = o.v[k];
    }
    if (half == 0) {
      if (NA > NB) keep[(krow + NA - 1) * kBlock + t] = o.v[NA - 1];
    } else {
      keep[kRowS * kBlock + t] = vi;
    }
  };

  // Operator coefficients of grid row `row` from the velocities next to the cell (GEN):
  // SecondOrderCentered::Dx_m1 / Dy_m1 / Dy_p1 / Dx_p1, include/space_disc.hpp:83-113, as k_offdiag forms them;
  // ghost rows as k_assemble leaves them (include/time_disc.hpp:304-337)
  auto form_coeffs = [&](const int row, const double gum, const double gup, const double vm, const double vp,
                         double *out) __attribute__((always_inline)) {
    const int ip = gen.row0 + row;
    const bool xin = ip >= 1 && ip <= gen.cx;
    double iw, is, in, ie;
    band_quotients(-gum / 2.0 - gen.kx, -vm / 2.0 - gen.ky, vp / 2.0 - gen.ky, gup / 2.0 - gen.kx, sdx, sdy, iw, is, in, ie);
    const bool inner = xin && yin;
    out[0] = inner ? iw : ((yin && ip == gen.cx + 1) ? 1.0 : 0.0);
    out[1] = inner ? is : ((xin && c == gen.cy + 1) ? 1.0 : 0.0);
    out[3] = inner ? in : ((xin && c == 0) ? 1.0 : 0.0);
    out[4] = inner ? ie : ((yin && ip == 0) ? 1.0 : 0.0);
    out[2] = inner ? gdiag : (xin ? (c == 0 ? gen.dg_bottom : gen.dg_top) : (yin ? gen.dg_left : 1.0));
  };
  // GEN, half 0: the coefficients of the row the next round works on.  They are formed one round ahead,
  // between the round's two barriers, while half 1 continues the update chain and half 0 would only wait:
  // their four divisions are then off the path from one row to the next.  On entry to round rr:
  // bcn = coefficients of row rr, gu0 / gu1 = velocity of rows rr and rr + 1.
  // (Up to 44 earlier vectors; beyond, the registers this takes cost more than it saves.)
  constexpr bool AHEAD = GEN && NV <= 44;
  constexpr bool W_AHEAD = false;  // also w = (A s_NV) / sigma of the next row (measured: slower, half 0 becomes the longer one)
  // So is w = (A s_NV) / sigma on that row (wn): the rows of s_NV it needs are in the ring by then.
  double bcn[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  double wn = 0.0;
  auto form_w = [&](const int row, const double *cf) __attribute__((always_inline)) -> double {
    const int sl = (row - rs + 2) & 3;
    const double *s0 = sring + sl * kBlock;
    const double xs = c > 0 ? s0[tm] : 0.0;
    const double xn = c + 1 < P ? s0[tp] : 0.0;
    const double tin = row_sum(cf[0], cf[1], cf[2], cf[3], cf[4], sring[((sl + 3) & 3) * kBlock + t], xs, s0[t], xn,
                               sring[((sl + 1) & 3) * kBlock + t]);
    return tin / nrm;
  };
  if (AHEAD && half == 0) {
    form_coeffs(rs - 1, gu0, qa.gu, qa.b[1], qa.b[3], bcn);
    gu0 = gu1;
    gu1 = qa.gu;
    if (W_AHEAD) wn = form_w(rs - 1, bcn);
  }

  auto round0 = [&](const int rr, Req &o, Req &n) __attribute__((always_inline)) {
    const int64_t idxn = rowc(rr + 1) * pitch + off;
    double bc[5];
    if (GEN && !AHEAD) {
      const double gum = gu0, gup = o.gu;
      gu0 = gu1;
      gu1 = gup;
      issue_op(n, rr + 1);
      form_coeffs(rr, gum, gup, o.b[1], o.b[3], bc);
    } else {
#pragma unroll
      for (int j = 0; j < 5; j++) bc[j] = AHEAD ? bcn[j] : o.b[j];
      issue_op(n, rr + 1);
    }
    const int sl = (rr - rs + 2) & 3;
    double u = (AHEAD && W_AHEAD) ? wn : form_w(rr, bc);
#pragma unroll
    for (int k = 0; k < NA; k++) {
      n.v[k] = __builtin_nontemporal_load(&Vh[(int64_t)k * ld + idxn]);
      u = u - lane_value(hpack, k) * o.v[k];
    }
    xa[t] = opaque(u);
#pragma unroll
    for (int j = 0; j < 5; j++) bcs[j * kBlock + t] = bc[j];
    const double sval = o.s;
    n.s = __builtin_nontemporal_load(s_addr(rr + 3));
    sring[((sl + 2) & 3) * kBlock + t] = s_ok(rr + 2) ? sval : 0.0;  // row rr + 2 of s_NV
    __syncthreads();
    if (AHEAD) {
      // row rr + 1: its own velocities were requested at the top of this round (n.b), the row behind it too (n.gu)
      form_coeffs(rr + 1, gu0, n.gu, n.b[1], n.b[3], bcn);
      gu0 = gu1;
      gu1 = n.gu;
      if (W_AHEAD) wn = form_w(rr + 1, bcn);  // rows rr .. rr + 2 of s_NV are in the ring (rr + 2 since this round's first barrier)
    }
    __syncthreads();
    emit_park(rr, o, 0.0);
  };

  auto round1 = [&](const int rr, Req &o, Req &n) __attribute__((always_inline)) {
    const bool rowvalid = rr >= 0 && rr < rows;
    const bool active = rowvalid && ucol;
    const int64_t idxn = rowc(rr + 1) * pitch + off;
    // this half's group of the update (kGroupedFrom), summed while half 0 works on its own
    const int sl = (rr - rs + 2) & 3;
    const double vi = sring[sl * kBlock + t];  // row rr of s_NV: in the ring since two rounds
    double pg = 0.0;
#pragma unroll
    for (int k = 0; k < NB; k++) {
      n.v[k] = __builtin_nontemporal_load(&Vh[(int64_t)k * ld + idxn]);
      const double term = lane_value(hpack, k) * o.v[k];
      pg = k == 0 ? term : pg + term;
    }
    pg = pg + lane_value(hpack, NB) * vi;
    // The operator applied to the new vector on row q = rr - 1: rows q - 1 and q are in the ring since
    // earlier rounds, so the first four terms of the row sum are formed here, before half 0's value arrives;
    // the term of row q + 1 (this round's value) follows behind the barrier.
    const int slot = (rr - rs + 1) & (kRing - 1);
    double tvp;
    {
      const double *rq = ring + ((slot + kRing - 1) & (kRing - 1)) * kBlock;
      const double *rm = ring + ((slot + kRing - 2) & (kRing - 1)) * kBlock;
      const double us = c > 0 ? rq[tm] : 0.0;
      const double un = c + 1 < P ? rq[tp] : 0.0;
      double r = 0.0;
      r = r + bp[0] * rm[t];
      r = r + bp[1] * us;
      r = r + bp[2] * rq[t];
      r = r + bp[3] * un;
      tvp = r;
    }
    __syncthreads();
    double u = opaque(xa[t] - pg);
    const double uhv = o.uh;
    n.uh = *(u_halo != nullptr ? &u_halo[rowh(rr + 1) * pitch + off] : coef);
    u = active ? u : ((u_halo != nullptr && !rowvalid && cin) ? uhv : 0.0);
    ring[slot * kBlock + t] = u;
    tvs[t] = tvp + bp[4] * u;
#pragma unroll
    for (int j = 0; j < 5; j++) bp[j] = bcs[j * kBlock + t];
    __syncthreads();
    emit_park(rr, o, vi);
    // The stores come last: nothing in the round waits for them (a row stays in the ring for kRing rounds).
    bool flush = rr - p0 + 1 == kStoreRows || rr == re - 1;
    if (geo.tick > 0 && (int32_t)(common_clock() - deadline) >= 0) {
      flush = true;
      do deadline += (uint32_t)geo.tick;
      while ((int32_t)(common_clock() - deadline) >= 0);
    }
    if (rr >= rs && rr < re && flush) {
      // strip-major: the workgroup stores the edge columns it evaluated as well (their owners store the same bits)
      if (sm ? (t >= 1 && t <= wown + 2) : colowned) {
        const int so = sm ? off : c;
#pragma unroll
        for (int j = 0; j < kStoreRows; j++) {
          const int r = p0 + j;
          if (r <= rr) __builtin_nontemporal_store(ring[((r - rs + 1) & (kRing - 1)) * kBlock + t], &u_out[(int64_t)r * pitch + so]);
        }
      }
      p0 = rr + 1;
    }
  };

  if (half == 0) {
    for (int rr = rs - 1; rr <= re; rr += 2) {
      round0(rr, qa, qb);
      if (rr + 1 > re) break;
      round0(rr + 1, qb, qa);
    }
  } else {
    for (int rr = rs - 1; rr <= re; rr += 2) {
      round1(rr, qa, qb);
      if (rr + 1 > re) break;
      round1(rr + 1, qb, qa);
    }
  }

  // partial sums in the layout of k_arnoldi_fused; every quantity belongs to one half
  auto put = [&](int q, double v) __attribute__((always_inline)) {
    const double tsum = wave_sum(v);
    if (lane == 0) red[q * 4 + wv] = tsum;
  };
#pragma unroll
  for (int k = 0; k < NB; k++) {
    put(kb + k, aT[k]);
    put(NV + 2 + kb + k, aS[k]);
  }
  if (half == 0) {
    if (NA > NB) {
      put(NA - 1, aT[NA - 1]);
      put(NV + 2 + NA - 1, aS[NA - 1]);
    }
    put(NV + 1, aTU);
    put(2 * NV + 3, aUU);
  } else {
    put(NV, aT[NB]);
    put(2 * NV + 2, aS[NB]);
    if (TT) put(2 * NV + 4, aTT);
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    double tot = red[threadIdx.x * 4];
#pragma unroll
    for (int i = 1; i < 4; i++) tot += red[threadIdx.x * 4 + i];
    part[(size_t)threadIdx.x * (geo.nstrips * geo.nbands) + (band * geo.nstrips + strip)] = tot;
  }
}

// The update alone, on the cnt unknowns from k0 on (the whole vector: last step of a
// cycle; one grid row: the rows a neighbouring rank needs): s_{NV+1} and its squared
// norm (part[workgroup]).  n: unknowns of the local vector; s_halo: entries in front of
// unknown 0 and behind unknown n - 1 exist in memory (one grid row each).
// Position of unknown (r, c) of a grid in a strip-major vector (see FusedGeom::sm): in the block of the
// strip that owns column c.
__device__ __forceinline__ int64_t strip_pos(const StripMap &m, int r, int c) {
  const int s = c / m.W;
  return (int64_t)s * m.SB + (int64_t)(r + 1) * m.pitch + (c - s * m.W + 2);
}

// map.on: the basis vectors (V, u_in, u_out) are strip-major; the bands stay in grid order.
template <int NV>
__global__ __launch_bounds__(kBlock) void k_arnoldi_close(Bands b, const double *__restrict__ V, int64_t ld,
                                                           const double *__restrict__ u_in,
                                                           const double *__restrict__ coef,
                                                           double *__restrict__ u_out, int64_t n, int P,
                                                           int64_t k0, int64_t cnt, int s_halo,
                                                           double *__restrict__ part,
                                                           const int *__restrict__ stop, StripMap map,
                                                           const int *__restrict__ skip) {
  if (stopped(stop) || stopped(skip)) return;
  __shared__ double s[kBlock / 64];
  double h[NV + 1];
#pragma unroll
  for (int k = 0; k <= NV; k++) h[k] = coef[k];
  const double nrm = coef[kFusedNormSlot];
  double a = 0.0;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < cnt; j += (int64_t)gridDim.x * kBlock) {
    const int64_t i = k0 + j;
    const int col = (int)(i % P);
    // e: position of the unknown in a basis vector; rp: distance to the same column one grid row on
    const int64_t e = map.on ? strip_pos(map, (int)(i / P), col) : i;
    const int64_t rp = map.on ? map.pitch : P;
    double vk[NV > 0 ? NV : 1];
#pragma unroll
    for (int k = 0; k < NV; k++) vk[k] = __builtin_nontemporal_load(&V[(int64_t)k * ld + e]);
    const double xc = u_in[e];
    const double xw = (s_halo || i >= P) ? u_in[e - rp] : 0.0;
    const double xe = (s_halo || i + P < n) ? u_in[e + rp] : 0.0;
    const double xs = col > 0 ? u_in[e - 1] : 0.0;
    const double xn = col + 1 < P ? u_in[e + 1] : 0.0;
    const double tin = row_sum(b.w[i], b.s[i], b.c[i], b.n[i], b.e[i], xw, xs, xc, xn, xe);
    double u = tin / nrm;
    constexpr int NG = NV >= kGroupedFrom ? (NV + 1) / 2 : NV + 1;  // as the sweeps group the update
    double pg = 0.0;
#pragma unroll
    for (int k = 0; k < NV; k++) {
      const double term = h[k] * vk[k];
      if (k < NG)
        u = u - term;
      else
        pg = k == NG ? term : pg + term;
    }
    if (NG <= NV) {
      const double term = h[NV] * xc;
      pg = NG == NV ? term : pg + term;
      u = u - pg;
    } else {
      u = u - h[NV] * xc;
    }
    __builtin_nontemporal_store(u, &u_out[e]);
    a = fma(u, u, a);
  }
  const double tot = block_sum(a, s);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}

// dst (strip-major, all kBlock positions of every row incl. the edge columns stored twice; zero outside the
// grid) = src (grid order, rows x P).  src_halo: the grid rows in front of / behind src exist and are copied
// into the blocks' first / last row; otherwise those rows are set to zero.
__global__ __launch_bounds__(kBlock) void k_to_strips(const double *__restrict__ src, double *__restrict__ dst, int rows,
                                                       int P, StripMap map, int src_halo) {
  const int lt = threadIdx.x;
  const int64_t total = (int64_t)map.nstrips * (rows + 2);
  for (int64_t q = blockIdx.x; q < total; q += gridDim.x) {
    const int s = (int)(q / (rows + 2)), r = (int)(q % (rows + 2)) - 1;
    const int c = s * map.W - 2 + lt;
    const int wown = (s + 1) * map.W <= P ? map.W : P - s * map.W;
    const bool ok = c >= 0 && c < P && lt <= wown + 3 && ((r >= 0 && r < rows) || src_halo != 0);
    __builtin_nontemporal_store(ok ? src[(int64_t)r * P + c] : 0.0, &dst[(int64_t)s * map.SB + (int64_t)(r + 1) * map.pitch + lt]);
  }
}

// v = src / sqrt(sum of part), stored twice in one pass over src: strip-major (every position of every row of
// every block, zero outside the grid and in the blocks' first and last row) and in grid order (the hand-over's
// operator kernel reads that copy).  True division, as normalize() in include/matrix.hpp:54-60; *out = the norm.
__global__ __launch_bounds__(kBlock) void k_scale_to_strips(const double *__restrict__ src, double *__restrict__ grid_dst,
                                                             double *__restrict__ strip_dst,
                                                             const double *__restrict__ part, int np, int rows, int P,
                                                             StripMap map, double *__restrict__ out,
                                                             const int *__restrict__ stop) {
  if (stopped(stop)) return;
  __shared__ double sh[kBlock / 64];
  const double nrm = sqrt(sum_parts(part, np, sh));
  if (blockIdx.x == 0 && threadIdx.x == 0 && out) *out = nrm;
  const int lt = threadIdx.x;
  const int64_t total = (int64_t)map.nstrips * (rows + 2);
  for (int64_t q = blockIdx.x; q < total; q += gridDim.x) {
    const int s = (int)(q / (rows + 2)), r = (int)(q % (rows + 2)) - 1;
    const int c = s * map.W - 2 + lt;
    const int wown = (s + 1) * map.W <= P ? map.W : P - s * map.W;
    const bool ok = c >= 0 && c < P && lt <= wown + 3 && r >= 0 && r < rows;
    double v = 0.0;
    if (ok) v = __builtin_nontemporal_load(&src[(int64_t)r * P + c]) / nrm;
    __builtin_nontemporal_store(v, &strip_dst[(int64_t)s * map.SB + (int64_t)(r + 1) * map.pitch + lt]);
    if (ok && lt >= 2 && lt < wown + 2) __builtin_nontemporal_store(v, &grid_dst[(int64_t)r * P + c]);
  }
}

// x (grid order) = sum_k y_k V_k of strip-major vectors; per element the operations of k_combine.  A
// workgroup takes one row of one strip at a time (lane = position in the row), so every read is one
// aligned 2 KB line.  (Lanes that move 16 bytes, two rows per workgroup, were measured slower: 5.62
// against 5.39 ms on an 8192 x 8192 grid with 50 vectors.)  The default is k_combine_strips_rows below.
template <int U>
__global__ __launch_bounds__(kBlock) void k_combine_strips(const double *__restrict__ V, int64_t ld,
                                                            const double *__restrict__ y, int m,
                                                            double *__restrict__ x, int rows, int P,
                                                            const int *__restrict__ count, int accumulate,
                                                            const double *__restrict__ sigma, StripMap map) {
  __shared__ double ys[kMaxM];
  if (count != nullptr) {
    const int c = *count;
    if (c < m) m = c;
  }
  for (int i = threadIdx.x; i < m; i += kBlock) ys[i] = sigma != nullptr ? y[i] / sigma[i] : y[i];
  __syncthreads();
  const int lt = threadIdx.x;
  const int64_t total = (int64_t)map.nstrips * rows;
  for (int64_t q = blockIdx.x; q < total; q += gridDim.x) {
    const int s = (int)(q / rows), r = (int)(q % rows);
    const int c = s * map.W - 2 + lt;
    const int wown = (s + 1) * map.W <= P ? map.W : P - s * map.W;
    if (lt < 2 || lt >= wown + 2) continue;  // the strip's own columns only
    const int64_t e = (int64_t)s * map.SB + (int64_t)(r + 1) * map.pitch + lt;
    const int64_t i = (int64_t)r * P + c;
    double acc = accumulate ? x[i] : 0.0;
#pragma unroll U
    for (int k = 0; k < m; k++) acc = acc + ys[k] * __builtin_nontemporal_load(&V[(int64_t)k * ld + e]);
    __builtin_nontemporal_store(acc, &x[i]);
  }
}

// The same with R consecutive rows of a strip per visit (a lane keeps R running sums): with the rows of all
// vectors interleaved, the R rows of the m vectors a workgroup reads per visit are one contiguous piece of memory.
template <int R>
__global__ __launch_bounds__(kBlock) void k_combine_strips_rows(const double *__restrict__ V, int64_t ld,
                                                                 const double *__restrict__ y, int m,
                                                                 double *__restrict__ x, int rows, int P,
                                                                 const int *__restrict__ count, int accumulate,
                                                                 const double *__restrict__ sigma, StripMap map) {
  __shared__ double ys[kMaxM];
  if (count != nullptr) {
    const int c = *count;
    if (c < m) m = c;
  }
  for (int i = threadIdx.x; i < m; i += kBlock) ys[i] = sigma != nullptr ? y[i] / sigma[i] : y[i];
  __syncthreads();
  const int lt = threadIdx.x;
  const int groups = (rows + R - 1) / R;
  const int64_t total = (int64_t)map.nstrips * groups;
  for (int64_t q = blockIdx.x; q < total; q += gridDim.x) {
    const int s = (int)(q / groups), r0 = (int)(q % groups) * R;
    const int c = s * map.W - 2 + lt;
    const int wown = (s + 1) * map.W <= P ? map.W : P - s * map.W;
    const bool own = lt >= 2 && lt < wown + 2;  // the strip's own columns only
    const int64_t e = (int64_t)s * map.SB + (int64_t)(r0 + 1) * map.pitch + lt;
    double acc[R];
#pragma unroll
    for (int j = 0; j < R; j++) acc[j] = (accumulate && own && r0 + j < rows) ? x[(int64_t)(r0 + j) * P + c] : 0.0;
#pragma unroll 2
    for (int k = 0; k < m; k++) {
      const double yk = ys[k];
      const double *vk = &V[(int64_t)k * ld + e];
#pragma unroll
      for (int j = 0; j < R; j++) {
        // rows behind the last one: the row of zeros / the next strip's block, read but not used
        const double v = __builtin_nontemporal_load(&vk[(int64_t)(r0 + j < rows ? j : 0) * map.pitch]);
        acc[j] = acc[j] + yk * v;
      }
    }
    if (own) {
#pragma unroll
      for (int j = 0; j < R; j++)
        if (r0 + j < rows) __builtin_nontemporal_store(acc[j], &x[(int64_t)(r0 + j) * P + c]);
    }
  }
}

// dst (grid order) = src (strip-major) / sigma[k]
__global__ __launch_bounds__(kBlock) void k_unscale_strips(const double *__restrict__ src, const double *__restrict__ sigma,
                                                            int k, double *__restrict__ dst, int64_t n, int P, StripMap map) {
  const double sg = sigma != nullptr ? sigma[k] : 1.0;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    dst[i] = src[strip_pos(map, (int)(i / P), (int)(i % P))] / sg;
}

// tot[q] = sum of part[q * np .. q * np + np): one workgroup per quantity.
__global__ __launch_bounds__(kBlock) void k_reduce_parts_wide(const double *__restrict__ part, int np,
                                                               double *__restrict__ tot,
                                                               const int *__restrict__ stop,
                                                               const int *__restrict__ skip) {
  if (stopped(stop) || stopped(skip)) return;
  __shared__ double s[kBlock / 64];
  const int q = blockIdx.x;
  double a = 0.0;
  for (int i = threadIdx.x; i < np; i += kBlock) a += part[(size_t)q * np + i];
  const double r = block_sum(a, s);
  if (threadIdx.x == 0) tot[q] = r;
}

// The coefficients of step i as a kernel of its own (see fused_coeffs_body): the hand-over
// from the chain (boot), and every step on several ranks, where the totals are all-reduced.
__global__ __launch_bounds__(64) void k_fused_coeffs(const double *__restrict__ tot, int i, int boot,
                                                      double *__restrict__ G, int gld,
                                                      double *__restrict__ H, int m,
                                                      double *__restrict__ coef,
                                                      double *__restrict__ sigma,
                                                      const int *__restrict__ stop, int *__restrict__ norm_flag) {
  if (stopped(stop)) return;
  __shared__ double gs[kFusedMaxM * (kFusedMaxM + 1)];
  fused_coeffs_body(tot, i, boot, G, gld, H, m, coef, sigma, gs, norm_flag);
}

// End of a cycle: sigma_m = ||s_m|| from the closing kernel's total, H[m][m-1] = sigma_m.
__global__ __launch_bounds__(64) void k_fused_finish(const double *__restrict__ tot, int m, double *__restrict__ H,
                                                      double *__restrict__ sigma, const int *__restrict__ stop,
                                                      const int *__restrict__ skip) {
  if (stopped(stop) || stopped(skip)) return;
  if (threadIdx.x == 0) {
    const double nrm = sqrt(tot[0]);
    sigma[m] = nrm;
    H[(size_t)m * m + (m - 1)] = nrm;
  }
}

// dst = src / sigma[k] (true division, as normalize() in include/matrix.hpp:54-60); without sigma a plain copy.
// pitch != 0: src is a vector of a chunk-interleaved basis.
__global__ __launch_bounds__(kBlock) void k_unscale(const double *__restrict__ src, const double *__restrict__ sigma,
                                                     int k, double *__restrict__ dst, int64_t n, int64_t pitch) {
  if (sigma == nullptr) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
      dst[i] = src[il1(i, pitch)];
    return;
  }
  const double sg = sigma[k];
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock)
    dst[i] = src[il1(i, pitch)] / sg;
}

// first step served by the 512-thread kernel (set_split_from, else env KRY_SPLIT_FROM, else
// kSplitFrom; instantiated from 12 on; beyond the last step: never)
std::atomic<int> g_split_from{-1};
int split_from() {
  int v = g_split_from.load(std::memory_order_relaxed);
  if (v < 0) {
    v = getenv("KRY_SPLIT_FROM") ? atoi(getenv("KRY_SPLIT_FROM")) : kSplitFrom;
    g_split_from.store(v, std::memory_order_relaxed);
  }
  return v < 12 ? 12 : v;
}

// Last step served by the 512-thread kernel.  With 47 and 48 earlier vectors it needs scratch memory; with 47 it
// is still the faster one (5.38 against 5.60 ms on an 8192 x 8192 grid), with 48 the 256-thread kernel is (5.66
// against 5.77 ms).  Env KRY_SPLIT_UPTO overrides.
int split_upto() {
  static std::atomic<int> v{-1};
  int r = v.load(std::memory_order_relaxed);
  if (r < 0) {
    r = getenv("KRY_SPLIT_UPTO") ? atoi(getenv("KRY_SPLIT_UPTO")) : 47;
    v.store(r, std::memory_order_relaxed);
  }
  return r;
}

// Strips and bands for step nv.  A band of R rows costs about kBandOverhead rows more than
// its share (halo rows at either end, filling the request window, the final reduction), and
// workgroups run in rounds of `cap` (the number resident at once: 4 per compute unit up to 128
// registers per lane, 3 up to 168, 2 up to 256, 1 for the 512-thread kernel): among the band
// counts that fill one to four rounds, the one with the best product of occupancy and
// useful share is taken (8192^2: 31 bands in four rounds; 2048^2 and the row blocks of
// eight ranks: fewer, taller bands).  At most 1024 workgroups (the partial-sum buffers).
// The choice depends on nv alone, not on which kernel variant a knob selects.
constexpr int kBandOverhead = 6;
// Period of the common store clock for step nv, in 10 ns ticks (0: stores when the ring is full only).
// Measured per instantiation on an 8192 x 8192 grid: 20 us up to 7 earlier vectors, 31 us up to 12, 41 us up
// to 18, 61 us up to 27, 82 us beyond; the shortest sweeps gain nothing.  Env KRY_STORE_TICK=period[,first_nv] overrides.
int store_tick(int nv) {
  static std::atomic<int> period{-2}, from{0};
  if (period.load(std::memory_order_relaxed) == -2) {
    const char *e = getenv("KRY_STORE_TICK");
    int p = -1, f = 0;
    if (e) {
      p = atoi(e);
      if (p < 0) p = 0;
      const char *c = strchr(e, ',');
      if (c) f = atoi(c + 1);
    }
    from.store(f, std::memory_order_relaxed);
    period.store(p, std::memory_order_relaxed);
  }
  const int p = period.load(std::memory_order_relaxed);
  if (p >= 0) return nv >= from.load(std::memory_order_relaxed) ? p : 0;
  // (41 to 47 earlier vectors: the sweeps with 16-row rings, re-measured with the interleaved basis)
  if (nv >= 41) return nv <= 44 ? 4096 : (nv <= 47 ? 12288 : 8192);
  return nv <= 3 ? 0 : (nv <= 7 ? 2048 : (nv <= 12 ? 3072 : (nv <= 18 ? 4096 : (nv <= 27 ? 6144 : 8192))));
}
// Register sets of the 256-thread kernel for step nv (its requests run sets - 1 rows ahead).  Sweeps over few
// vectors have few requests per row and registers to spare: more rows in flight.  Env KRY_FUSED_SETS=s0,s1,...
// (one number per step from 0 on, 2 to 4) overrides; steps beyond kDeepSetsUpto always use two.
constexpr int kDeepSetsUpto = 17;
int fused_sets(int nv) {
  static std::atomic<int> table[kDeepSetsUpto + 1];
  static std::atomic<int> ready{0};
  if (!ready.load(std::memory_order_acquire)) {
    // measured per step on an 8192 x 8192 grid: three sets pay with 8 to 10 earlier vectors (4 to 6 %), nowhere else
    for (int i = 0; i <= kDeepSetsUpto; i++) table[i].store(i >= 8 && i <= 10 ? 3 : 2, std::memory_order_relaxed);
    if (const char *e = getenv("KRY_FUSED_SETS")) {
      int i = 0;
      for (const char *p = e; *p && i <= kDeepSetsUpto; i++) {
        int v = atoi(p);
        table[i].store(v < 2 ? 2 : (v > 4 ? 4 : v), std::memory_order_relaxed);
        p = strchr(p, ',');
        if (!p) break;
        p++;
      }
    }
    ready.store(1, std::memory_order_release);
  }
  return nv >= 0 && nv <= kDeepSetsUpto ? table[nv].load(std::memory_order_relaxed) : 2;
}

// Workgroups of the 256-thread kernel for step nv that a compute unit holds at once (registers and LDS),
// asked of the runtime once per step; 0 until then.
std::atomic<int> g_resident[kFusedMaxM];
int fused_resident_query(int nv);  // below, behind the kernel table

FusedGeom fused_geom(int rows, int P, int nv) {
  FusedGeom g;
  g.rows = rows;
  g.P = P;
  g.nstrips = (P + kStripMax - 1) / kStripMax;
  g.W = (P + g.nstrips - 1) / g.nstrips;
  int per_cu = nv <= 5 ? 4 : (nv <= 10 ? 3 : (nv < kSplitFrom ? 2 : 1));
  if (nv <= kDeepSetsUpto && fused_sets(nv) > 2) {
    const int q = fused_resident_query(nv);
    if (q >= 1 && q <= 4) per_cu = q;
  }
  const int cap = 256 * per_cu;
  int best_nb = 1;
  double best = -1.0;
  // `slots` workgroups run at a time, `r` rounds of them
  auto consider = [&](int slots, int r) {
    int nb = (slots * r) / g.nstrips;
    if (nb > rows / 32) nb = rows / 32;  // bands of at least 32 rows
    if (nb < 1) nb = 1;
    const int R = (rows + nb - 1) / nb, nbands = (rows + R - 1) / R;
    const int count = g.nstrips * nbands, rounds = (count + slots - 1) / slots;
    const double score = (double)count / ((double)slots * rounds) * (double)R / (double)(R + kBandOverhead);
    if (score > best) {
      best = score;
      best_nb = nb;
    }
  };
  for (int r = 1; r <= 4 && cap * r <= 1024; r++) consider(cap, r);
  // On a small grid the bands cannot get short enough to fill every slot (2048^2 with three slots per
  // compute unit: 567 workgroups of 33 rows, three on some compute units and two on others); fewer, taller
  // bands that give every compute unit the same number (two each: 504 of 37 rows) are faster then.
  for (int k = 2; 256 * k < cap; k++) consider(256 * k, 1);
  g.R = (rows + best_nb - 1) / best_nb;
  g.nbands = (rows + g.R - 1) / g.R;
  g.tick = store_tick(nv);
  return g;
}

template <int NV>
constexpr size_t fused_lds() {
  return sizeof(double) * ((size_t)(NV + 1) * kBlock + (store_rows(NV) + 4) * kBlock + (2 * NV + 5) * (kBlock / 64));
}

// the 256-thread kernel with S register sets
template <int NV, bool GEN, int S>
void launch_fused_sets(hipStream_t st, const FusedGeom &geo, const Bands &b, const BandGen &gen, const double *V,
                       int64_t ld, const double *u_in, const double *coef, double *u_out, double *part,
                       const double *u_halo, int s_halo, int launch_bands, const CoefIn &cin) {
  constexpr size_t lds = fused_lds<NV>();
  static std::atomic<uint64_t> configured{0};
  allow_lds(&k_arnoldi_fused<NV, GEN, S>, lds, configured, "LDS size of the one-sweep kernel");
  hipLaunchKernelGGL((k_arnoldi_fused<NV, GEN, S>), dim3(geo.nstrips * launch_bands), dim3(kBlock), lds, st, geo, b, gen, V,
                     ld, u_in, coef, u_out, part, u_halo, s_halo, launch_predicate_current(), cin);
}

template <int NV, int S>
int resident_t() {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k_arnoldi_fused<NV, true, S>, kBlock, fused_lds<NV>()) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

template <int NV, bool GEN>
void launch_fused_t(hipStream_t st, const FusedGeom &geo, const Bands &b, const BandGen &gen, const double *V,
                    int64_t ld, const double *u_in, const double *coef, double *u_out, double *part,
                    const double *u_halo, int s_halo, int launch_bands, const CoefIn &cin) {
  constexpr size_t lds = fused_lds<NV>();
  static_assert(lds <= 160 * 1024, "LDS of one compute unit");
  if constexpr (GEN && NV <= kDeepSetsUpto) {
    const int sets = fused_sets(NV);
    if (sets == 3)
      return launch_fused_sets<NV, GEN, 3>(st, geo, b, gen, V, ld, u_in, coef, u_out, part, u_halo, s_halo, launch_bands, cin);
    if (sets == 4)
      return launch_fused_sets<NV, GEN, 4>(st, geo, b, gen, V, ld, u_in, coef, u_out, part, u_halo, s_halo, launch_bands, cin);
  }
  static std::atomic<uint64_t> configured{0};
  allow_lds(&k_arnoldi_fused<NV, GEN>, lds, configured, "LDS size of the one-sweep kernel");
  // (with stored bands the 512-thread kernel needs scratch from 47 earlier vectors on and is slower there)
  if constexpr (NV >= 12 && (GEN || NV <= 46)) {
    if (NV >= split_from() && NV <= split_upto()) {
      constexpr size_t lds2 = sizeof(double) * ((size_t)(NV + 1 - 2 * split_park_regs(NV)) * kBlock +
                                                (split_store_rows(NV) + 4 + 7) * kBlock + (2 * NV + 5) * 4);
      static_assert(lds2 <= 160 * 1024, "LDS of one compute unit");
      static std::atomic<uint64_t> configured2{0};
      allow_lds(&k_arnoldi_split<NV, GEN>, lds2, configured2, "LDS size of the 512-thread one-sweep kernel");
      hipLaunchKernelGGL((k_arnoldi_split<NV, GEN>), dim3(geo.nstrips * launch_bands), dim3(kSplitBlock), lds2, st, geo, b,
                         gen, V, ld, u_in, coef, u_out, part, u_halo, s_halo, launch_predicate_current(), cin);
      return;
    }
  }
  hipLaunchKernelGGL((k_arnoldi_fused<NV, GEN>), dim3(geo.nstrips * launch_bands), dim3(kBlock), lds, st, geo, b, gen, V,
                     ld, u_in, coef, u_out, part, u_halo, s_halo, launch_predicate_current(), cin);
}

struct CloseArgs {
  Bands b;
  const double *V;
  int64_t ld;
  const double *coef;
  int64_t n;
  int P;
  int64_t k0, cnt;
  int s_halo;
  double *part;
  int grid;
  StripMap map;
  const int *skip;
};

template <int NV>
void launch_close_t(hipStream_t st, const CloseArgs &a) {
  hipLaunchKernelGGL(k_arnoldi_close<NV>, dim3(a.grid), dim3(kBlock), 0, st, a.b, a.V, a.ld, a.V + (int64_t)NV * a.ld,
                     a.coef, const_cast<double *>(a.V) + (int64_t)(NV + 1) * a.ld, a.n, a.P, a.k0, a.cnt, a.s_halo,
                     a.part, launch_predicate_current(), a.map, a.skip);
}

template <int NV = kFusedMaxM - 1>
struct FusedTable {
  static void sweep(int nv, hipStream_t st, const FusedGeom &geo, const Bands &b, const BandGen *gen, const double *V,
                    int64_t ld, const double *coef, double *part, bool u_halo, int s_halo, int launch_bands,
                    const CoefIn &cin) {
    if (nv == NV) {
      double *u_out = const_cast<double *>(V) + (int64_t)(NV + 1) * ld;
      const double *uh = u_halo ? u_out : nullptr;
      if (gen)
        launch_fused_t<NV, true>(st, geo, b, *gen, V, ld, V + (int64_t)NV * ld, coef, u_out, part, uh, s_halo, launch_bands, cin);
      else
        launch_fused_t<NV, false>(st, geo, b, BandGen(), V, ld, V + (int64_t)NV * ld, coef, u_out, part, uh, s_halo, launch_bands,
                                  cin);
    } else {
      FusedTable<NV - 1>::sweep(nv, st, geo, b, gen, V, ld, coef, part, u_halo, s_halo, launch_bands, cin);
    }
  }
  static void close(int nv, hipStream_t st, const CloseArgs &a) {
    if (nv == NV)
      launch_close_t<NV>(st, a);
    else
      FusedTable<NV - 1>::close(nv, st, a);
  }
  static int resident(int nv, int sets) {
    if constexpr (NV <= kDeepSetsUpto) {
      if (nv == NV) return sets == 4 ? resident_t<NV, 4>() : (sets == 3 ? resident_t<NV, 3>() : resident_t<NV, 2>());
    }
    return FusedTable<NV - 1>::resident(nv, sets);
  }
};

template <>
struct FusedTable<-1> {
  static void sweep(int, hipStream_t, const FusedGeom &, const Bands &, const BandGen *, const double *, int64_t,
                    const double *, double *, bool, int, int, const CoefIn &) {}
  static void close(int, hipStream_t, const CloseArgs &) {}
  static int resident(int, int) { return 0; }
};

int fused_resident_query(int nv) {
  int r = g_resident[nv].load(std::memory_order_relaxed);
  if (r == 0) {
    r = FusedTable<>::resident(nv, fused_sets(nv));
    if (r < 1) r = -1;  // the runtime would not say: keep the rule of thumb
    g_resident[nv].store(r, std::memory_order_relaxed);
  }
  return r;
}

}  // namespace

bool fused_sweep_has_tt(int nv) { return sweep_has_tt(nv); }

int get_split_upto() { return split_upto(); }

void set_split_from(int first_step) { g_split_from.store(first_step < 0 ? kSplitFrom : first_step); }
int get_split_from() { return split_from(); }

int fused_parts(int rows, int cols) {
  int most = 0;  // the largest count over all steps
  for (int nv = 0; nv < kFusedMaxM; nv++) {
    const FusedGeom g = fused_geom(rows, cols, nv);
    if (g.nstrips * g.nbands > most) most = g.nstrips * g.nbands;
  }
  return most;
}

StripMap strip_map(int rows, int cols, int slots) {
  StripMap m;
  m.on = 1;
  // strips depend on the row length alone (as in fused_geom)
  m.nstrips = (cols + kStripMax - 1) / kStripMax;
  m.W = (cols + m.nstrips - 1) / m.nstrips;
  m.pitch = slots * kBlock;
  m.SB = (int64_t)(rows + 2) * m.pitch;
  return m;
}

void launch_to_strips(hipStream_t st, const double *src, double *dst, int rows, int cols, bool src_has_halo_rows) {
  const StripMap m = strip_map(rows, cols);
  const int64_t total = (int64_t)m.nstrips * (rows + 2);
  hipLaunchKernelGGL(k_to_strips, dim3((unsigned)(total < 65536 ? total : 65536)), dim3(kBlock), 0, st, src, dst, rows, cols, m,
                     src_has_halo_rows ? 1 : 0);
}

void launch_scale_to_strips(hipStream_t st, const double *src, double *grid_dst, double *strip_dst, const double *part,
                            int nparts, int rows, int cols, int slots, double *out) {
  const StripMap m = strip_map(rows, cols, slots);
  const int64_t total = (int64_t)m.nstrips * (rows + 2);
  hipLaunchKernelGGL(k_scale_to_strips, dim3((unsigned)(total < 2048 ? total : 2048)), dim3(kBlock), 0, st, src, grid_dst,
                     strip_dst, part, nparts, rows, cols, m, out, launch_predicate_current());
}

void launch_combine_strips(hipStream_t st, const double *V, const double *y, int m, double *x, int rows, int cols,
                           int slots, const int *count, bool accumulate, const double *sigma) {
  const StripMap map = strip_map(rows, cols, slots);
  const int64_t ld = kBlock;  // vector k starts k rows of a strip into the interleaved basis
  const int64_t total = (int64_t)map.nstrips * rows;
  // Sixteen rows of a strip per visit and one workgroup per compute unit, measured on an 8192 x 8192 grid with 50
  // vectors (KRY_COMBINE_ROWS x KRY_COMBINE_GRID): 1 x 1024 (one row per visit) 5.40 ms, 4 x 1024 5.22, 8 x 1024
  // 5.08, 16 x 1024 4.94, 8 x 256 4.85, 16 x 256 4.70 (5.83 TB/s), 16 x 128 6.70, 16 x 384 5.40, 32 rows: spills.
  static const int rows_env = getenv("KRY_COMBINE_ROWS") ? atoi(getenv("KRY_COMBINE_ROWS")) : 16;
  static const int grid_env = getenv("KRY_COMBINE_GRID") ? atoi(getenv("KRY_COMBINE_GRID")) : (rows_env > 1 ? 256 : 1024);
  static const int unroll_env = getenv("KRY_COMBINE_UNROLL") ? atoi(getenv("KRY_COMBINE_UNROLL")) : 8;
  const unsigned grid = (unsigned)(total < grid_env ? total : grid_env);
  if (rows_env == 4 || rows_env == 8 || rows_env == 16) {
    const int64_t tot = (int64_t)map.nstrips * ((rows + rows_env - 1) / rows_env);
    const unsigned gr = (unsigned)(tot < grid_env ? tot : grid_env);
    if (rows_env == 4)
      hipLaunchKernelGGL(k_combine_strips_rows<4>, dim3(gr), dim3(kBlock), 0, st, V, ld, y, m, x, rows, cols, count,
                         accumulate ? 1 : 0, sigma, map);
    else if (rows_env == 8)
      hipLaunchKernelGGL(k_combine_strips_rows<8>, dim3(gr), dim3(kBlock), 0, st, V, ld, y, m, x, rows, cols, count,
                         accumulate ? 1 : 0, sigma, map);

    else
      hipLaunchKernelGGL(k_combine_strips_rows<16>, dim3(gr), dim3(kBlock), 0, st, V, ld, y, m, x, rows, cols, count,
                         accumulate ? 1 : 0, sigma, map);
    return;
  }
  if (unroll_env >= 16)
    hipLaunchKernelGGL(k_combine_strips<16>, dim3(grid), dim3(kBlock), 0, st, V, ld, y, m, x, rows, cols, count,
                       accumulate ? 1 : 0, sigma, map);
  else if (unroll_env <= 4)
    hipLaunchKernelGGL(k_combine_strips<4>, dim3(grid), dim3(kBlock), 0, st, V, ld, y, m, x, rows, cols, count,
                       accumulate ? 1 : 0, sigma, map);
  else
    hipLaunchKernelGGL(k_combine_strips<8>, dim3(grid), dim3(kBlock), 0, st, V, ld, y, m, x, rows, cols, count,
                       accumulate ? 1 : 0, sigma, map);
}

void launch_unscale_strips(hipStream_t st, const double *src, const double *sigma, int k, double *dst, int rows, int cols,
                           int slots) {
  const int64_t n = (int64_t)rows * cols;
  hipLaunchKernelGGL(k_unscale_strips, dim3(grid_for(n)), dim3(kBlock), 0, st, src, sigma, k, dst, n, cols,
                     strip_map(rows, cols, slots));
}

void launch_arnoldi_fused(hipStream_t st, int nv, int rows, int cols, const Bands &b, const double *V, int64_t ld,
                          const double *coef, double *part, int *nparts, bool has_halo, const BandGen *gen, int which,
                          int sm_slots, const CoefIn *cin) {
  FusedGeom g = fused_geom(rows, cols, nv);
  if (sm_slots > 0) {
    g.sm = 1;
    g.pitch = sm_slots * kBlock;
    g.SB = (int64_t)(rows + 2) * g.pitch;
    g.SBg = (rows + 2) * kBlock;
  }
  *nparts = g.nstrips * g.nbands;
  int launch_bands = g.nbands;
  if (which == kSweepInterior) {
    if (g.nbands < 3) return;  // nothing between the first and the last band
    g.band_first = 1;
    launch_bands = g.nbands - 2;
  } else if (which == kSweepEdges) {
    if (g.nbands < 3) {
      launch_bands = g.nbands;  // no interior launch has run: everything here
    } else {
      g.band_step = g.nbands - 1;
      launch_bands = 2;
    }
  }
  FusedTable<>::sweep(nv, st, g, b, gen, V, ld, coef, part, has_halo, has_halo ? 1 : 0, launch_bands, cin ? *cin : CoefIn());
}

bool fused_bands_splittable(int nv, int rows, int cols) { return fused_geom(rows, cols, nv).nbands >= 3; }

void launch_arnoldi_close(hipStream_t st, int nv, const Bands &b, const double *V, int64_t ld, const double *coef,
                          int64_t n, int cols, int64_t k0, int64_t cnt, bool has_halo, double *part, int *nparts,
                          const StripMap *map, const int *skip) {
  int grid = grid_for(cnt);
  if (grid > 1024) grid = 1024;
  const CloseArgs a{b, V, ld, coef, n, cols, k0, cnt, has_halo ? 1 : 0, part, grid, map ? *map : StripMap(), skip};
  FusedTable<>::close(nv, st, a);
  *nparts = grid;
}

void launch_reduce_parts_wide(hipStream_t st, const double *part, int nq, int np, double *tot, const int *skip) {
  hipLaunchKernelGGL(k_reduce_parts_wide, dim3(nq), dim3(kBlock), 0, st, part, np, tot, launch_predicate_current(), skip);
}

void launch_fused_coeffs(hipStream_t st, const double *tot, int i, bool boot, double *G, int gld, double *H,
                         int m, double *coef, double *sigma, int *norm_flag) {
  hipLaunchKernelGGL(k_fused_coeffs, dim3(1), dim3(64), 0, st, tot, i, boot ? 1 : 0, G, gld, H, m, coef, sigma,
                     launch_predicate_current(), norm_flag);
}

void launch_fused_finish(hipStream_t st, const double *tot, int m, double *H, double *sigma, const int *skip) {
  hipLaunchKernelGGL(k_fused_finish, dim3(1), dim3(64), 0, st, tot, m, H, sigma, launch_predicate_current(), skip);
}

void launch_unscale(hipStream_t st, const double *src, const double *sigma, int k, double *dst, int64_t n,
                    int64_t pitch) {
  hipLaunchKernelGGL(k_unscale, dim3(grid_for(n)), dim3(kBlock), 0, st, src, sigma, k, dst, n, pitch);
}

}  // namespace kry
