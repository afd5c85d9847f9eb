// Device-side helpers shared by the kernel translation units.
#ifndef KRY_DEVICE_H_
#define KRY_DEVICE_H_

#include "kry_internal.h"

namespace kry {

// true when the solver this launch belongs to has already met its tolerance
__device__ __forceinline__ bool stopped(const int *__restrict__ stop) { return stop != nullptr && *stop != 0; }

// ---------------------------------------------------------------------------
// Reduction helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // lane 0 holds the wave total
}

// Sum over the workgroup, returned to every thread.  `s` holds kBlock/64 doubles.
__device__ __forceinline__ double block_sum(double v, double *s) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s[wv] = v;
  __syncthreads();
  double t = s[0];
#pragma unroll
  for (int i = 1; i < kBlock / 64; i++) t += s[i];
  return t;
}

// Total of a partial-sum array; every workgroup of every kernel evaluates the
// same expression tree, so all of them obtain the same bits.
__device__ __forceinline__ double sum_parts(const double *__restrict__ part, int np, double *s) {
  double a = 0.0;
  for (int i = threadIdx.x; i < np; i += kBlock) a += part[i];
  return block_sum(a, s);
}

typedef double d2 __attribute__((ext_vector_type(2)));

// Chunk-interleaved basis (the Gram-Schmidt chain on one rank): the vectors are cut into chunks of kChunk values
// and chunk c of vector 0, chunk c of vector 1, ... follow each other in memory, so that a sweep over many
// vectors reads ONE contiguous piece per chunk instead of one piece per vector, each in a different memory page
// (more than about thirty pages in rotation overflow the translation caches of a compute unit).  Vector k starts
// k x kChunk values into the array; `pitch` is the distance between two chunks of one vector (slots x kChunk),
// 0 for vectors that are contiguous.  il1 / il2: position of value i / of 16-byte element i2 within its vector.
__device__ __forceinline__ int64_t il1(int64_t i, int64_t pitch) {
  return pitch ? (i / kChunk) * pitch + (i % kChunk) : i;
}
__device__ __forceinline__ int64_t il2(int64_t i2, int64_t pitch) {
  return pitch ? (i2 / (kChunk / 2)) * (pitch / 2) + (i2 % (kChunk / 2)) : i2;
}

// ---------------------------------------------------------------------------
// a / d, correctly rounded, for a divisor that many divisions share.
//
// With y = RN(1 / d) (one true division), q0 = RN(a y) is within two units in the last place of a / d; one
// correction q1 = RN(q0 + (a - d q0) y), the residual formed by a fused multiply-add, is within half a unit plus
// a relative 2^-104, i.e. it is one of the two neighbours of a / d; and for such a q1, with the residual
// r1 = a - d q1 exact, RN(q1 + r1 y) IS RN(a / d) (Markstein's theorem on division by correction with a
// correctly rounded reciprocal).  Five multiply-adds at full rate instead of the division sequence with its
// quarter-rate reciprocal.  The proof needs every intermediate to be a normal number and r1 to be exact, which
// holds when 2^-100 <= |d| < 2^101 and 2^-800 <= |q| < 2^801 (then the quantum of d q1, 2^-104 |d q|, is far above
// the smallest subnormal); callers test both (shared_divisor().ok for d, div_shared_in_range for the quotient;
// zero, infinite and NaN quotients fail the test) and take the true division for a whole wavefront when any
// lane fails, so every result has the bits of a / d.
// ---------------------------------------------------------------------------
struct SharedDivisor {
  double d, y;
  bool ok;
};
__device__ __forceinline__ uint32_t exponent_field(double x) { return ((uint32_t)__double2hiint(x) >> 20) & 0x7ffu; }
// y: RN(1 / d), from one true division (on the host for divisors that come with the launch, which keeps d and y
// in scalar registers); ok: shared_divisor_in_range(d)
__device__ __forceinline__ SharedDivisor shared_divisor(double d, double y, bool ok) {
  SharedDivisor s;
  s.d = d;
  s.y = y;
  s.ok = ok;
  return s;
}
__device__ __forceinline__ double div_shared_fast(double a, const SharedDivisor &s) {
  double q = a * s.y;
  double r = fma(-s.d, q, a);
  q = fma(r, s.y, q);
  r = fma(-s.d, q, a);
  return fma(r, s.y, q);
}
__device__ __forceinline__ bool div_shared_in_range(double q) { return exponent_field(q) - 223u <= 1600u; }

// (the value itself behind an empty asm statement: a select on it stays a select instead of becoming a branch
// around the arithmetic that produced it)
__device__ __forceinline__ double div_opaque(double x) {
  asm volatile("" : "+v"(x));
  return x;
}

// The four quotients in a row's band entries (numerators over the two mesh widths, which the whole sweep
// shares) by div_shared_fast; a wavefront in which any quotient leaves the range in which that is proven exact
// takes the true divisions instead.  Either way the bits of n / d, as k_offdiag forms them.
__device__ __forceinline__ bool band_quotients(double n0, double n1, double n2, double n3, const SharedDivisor &dx,
                                               const SharedDivisor &dy, double &iw, double &is, double &in, double &ie) {
  iw = div_shared_fast(n0, dx);
  is = div_shared_fast(n1, dy);
  in = div_shared_fast(n2, dy);
  ie = div_shared_fast(n3, dx);
  const bool ok = dx.ok && dy.ok && div_shared_in_range(iw) && div_shared_in_range(is) && div_shared_in_range(in) &&
                  div_shared_in_range(ie);
  const bool fast = __all(ok);
  if (!fast) {
    iw = n0 / dx.d;
    is = n1 / dy.d;
    in = n2 / dy.d;
    ie = n3 / dx.d;
  }
  iw = div_opaque(iw);
  is = div_opaque(is);
  in = div_opaque(in);
  ie = div_opaque(ie);
  return fast;  // (for the self-test)
}

__device__ __forceinline__ d2 ld_nt(const d2 *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_nt(d2 v, d2 *p) { __builtin_nontemporal_store(v, p); }

// partial-sum arrays of a group of k vectors: k dot products with the work vector and the
// k - 1 Gram entries <c_a, c_{k-1}> of the group's last vector
constexpr int nred(int k) { return k > 0 ? 2 * k - 1 : 0; }

// Totals of NQ partial-sum arrays, evaluated by the same expression tree in
// every workgroup: wave q % 4 reduces array q with its 64 lanes.
template <int NQ>
__device__ __forceinline__ void sum_parts_multi(const double *__restrict__ part, int np, double *tot /* LDS, NQ */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int q = wv; q < NQ; q += kBlock / 64) {
    double a = 0.0;
    for (int i = lane; i < np; i += 64) a += part[(size_t)q * np + i];
    a = wave_sum(a);
    if (lane == 0) tot[q] = a;
  }
  __syncthreads();
}

// Block-level reduction of NQ accumulators into part[q * np + wg] (by default: one slot per
// workgroup of the launch; a launch that covers part of a larger set of workgroups passes
// its workgroup's number in that set and the size of the set).
template <int NQ>
__device__ __forceinline__ void reduce_store_multi(const double (&acc)[NQ > 0 ? NQ : 1], double *red /* LDS NQ*4 */,
                                                    double *__restrict__ part, int wg = -1, int np = 0) {
  if (wg < 0) {
    wg = blockIdx.x;
    np = gridDim.x;
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const double t = wave_sum(acc[q]);
    if (lane == 0) red[q * (kBlock / 64) + wv] = t;
  }
  __syncthreads();
  if (threadIdx.x < NQ) {
    double t = red[threadIdx.x * (kBlock / 64)];
#pragma unroll
    for (int i = 1; i < kBlock / 64; i++) t += red[threadIdx.x * (kBlock / 64) + i];
    part[(size_t)threadIdx.x * np + wg] = t;
  }
}

}  // namespace kry

#endif  // KRY_DEVICE_H_
