// Stand-alone micro-benchmark of streaming-kernel structures for the
// Gram-Schmidt link (w -= h*p ; partial <w, c>): which loop shape, grid size,
// unroll and cache policy get closest to the HBM roof on MI355X.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/kbench.hip -o /tmp/kbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cstring>

typedef double d2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
template <int BLK>
__device__ __forceinline__ double block_sum(double v, double *s) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) s[wv] = v;
  __syncthreads();
  double t = s[0];
  for (int i = 1; i < BLK / 64; i++) t += s[i];
  return t;
}

// V0: grid-stride, two streams one grid-stride apart (the first version)
__global__ __launch_bounds__(256) void v0(double *w, const double *p, const double *c, double h, double *part, long n2) {
  __shared__ double s[4];
  d2 *w2 = (d2 *)w; const d2 *p2 = (const d2 *)p; const d2 *c2 = (const d2 *)c;
  double a0 = 0, a1 = 0;
  const long stride = (long)gridDim.x * 256;
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  for (; i + stride < n2; i += 2 * stride) {
    d2 wa = w2[i], wb = w2[i + stride]; d2 pa = p2[i], pb = p2[i + stride]; d2 ca = c2[i], cb = c2[i + stride];
    wa = wa - h * pa; wb = wb - h * pb; w2[i] = wa; w2[i + stride] = wb;
    a0 = fma(wa.x, ca.x, a0); a1 = fma(wa.y, ca.y, a1); a0 = fma(wb.x, cb.x, a0); a1 = fma(wb.y, cb.y, a1);
  }
  if (i < n2) { d2 wa = w2[i]; d2 pa = p2[i]; d2 ca = c2[i]; wa = wa - h * pa; w2[i] = wa; a0 = fma(wa.x, ca.x, a0); a1 = fma(wa.y, ca.y, a1); }
  double t = block_sum<256>(a0 + a1, s);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// V1: each workgroup walks contiguous tiles of BLK*U double2; NT selects
// non-temporal loads/stores for the streamed operands.
template <int BLK, int U, bool NT>
__global__ __launch_bounds__(BLK) void v1(double *w, const double *p, const double *c, double h, double *part, long n2) {
  __shared__ double s[BLK / 64];
  d2 *w2 = (d2 *)w; const d2 *p2 = (const d2 *)p; const d2 *c2 = (const d2 *)c;
  double a0 = 0, a1 = 0;
  const long tile = (long)BLK * U;
  for (long base = (long)blockIdx.x * tile; base < n2; base += (long)gridDim.x * tile) {
    d2 wv[U], pv[U], cv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        if (NT) { wv[u] = __builtin_nontemporal_load(&w2[i]); pv[u] = __builtin_nontemporal_load(&p2[i]); cv[u] = __builtin_nontemporal_load(&c2[i]); }
        else { wv[u] = w2[i]; pv[u] = p2[i]; cv[u] = c2[i]; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        d2 r = wv[u] - h * pv[u];
        if (NT) __builtin_nontemporal_store(r, &w2[i]); else w2[i] = r;
        a0 = fma(r.x, cv[u].x, a0); a1 = fma(r.y, cv[u].y, a1);
      }
    }
  }
  double t = block_sum<BLK>(a0 + a1, s);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// V2: two links per pass (5 reads, 1 write, 3 dots): w -= h1 p1 ; w -= h2 p2 ; <w,c1>, <w,c2>, <c1,c2>
template <int BLK, int U, bool NT>
__global__ __launch_bounds__(BLK) void v2(double *w, const double *p1, const double *p2_, const double *c1, const double *c2_, double h1, double h2, double *part, long n2) {
  __shared__ double s[BLK / 64];
  d2 *w2 = (d2 *)w; const d2 *P1 = (const d2 *)p1; const d2 *P2 = (const d2 *)p2_; const d2 *C1 = (const d2 *)c1; const d2 *C2 = (const d2 *)c2_;
  double a = 0, b = 0, g = 0;
  const long tile = (long)BLK * U;
  for (long base = (long)blockIdx.x * tile; base < n2; base += (long)gridDim.x * tile) {
    d2 wv[U], p1v[U], p2v[U], c1v[U], c2v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        if (NT) { wv[u] = __builtin_nontemporal_load(&w2[i]); p1v[u] = __builtin_nontemporal_load(&P1[i]); p2v[u] = __builtin_nontemporal_load(&P2[i]); c1v[u] = __builtin_nontemporal_load(&C1[i]); c2v[u] = __builtin_nontemporal_load(&C2[i]); }
        else { wv[u] = w2[i]; p1v[u] = P1[i]; p2v[u] = P2[i]; c1v[u] = C1[i]; c2v[u] = C2[i]; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        d2 r = wv[u] - h1 * p1v[u];
        r = r - h2 * p2v[u];
        if (NT) __builtin_nontemporal_store(r, &w2[i]); else w2[i] = r;
        a = fma(r.x, c1v[u].x, a); a = fma(r.y, c1v[u].y, a);
        b = fma(r.x, c2v[u].x, b); b = fma(r.y, c2v[u].y, b);
        g = fma(c1v[u].x, c2v[u].x, g); g = fma(c1v[u].y, c2v[u].y, g);
      }
    }
  }
  double ta = block_sum<BLK>(a, s), tb = block_sum<BLK>(b, s), tg = block_sum<BLK>(g, s);
  if (threadIdx.x == 0) { part[blockIdx.x] = ta; part[gridDim.x + blockIdx.x] = tb; part[2 * gridDim.x + blockIdx.x] = tg; }
}


// V1 with the real prologue: every workgroup first sums NP partials of the previous kernel
template <int BLK, int U, int POL>  // POL 0: all NT; 1: NT loads of p,c only; 2: NT loads all, plain store
__global__ __launch_bounds__(BLK) void v1p(double *w, const double *p, const double *c, const double *pin, int np, double *part, long n2) {
  __shared__ double s[BLK / 64];
  double hh = 0.0;
  for (int i = threadIdx.x; i < np; i += BLK) hh += pin[i];
  const double h = block_sum<BLK>(hh, s) * 1e-30 + 1e-3;
  d2 *w2 = (d2 *)w; const d2 *p2 = (const d2 *)p; const d2 *c2 = (const d2 *)c;
  double a0 = 0, a1 = 0;
  const long tile = (long)BLK * U;
  for (long base = (long)blockIdx.x * tile; base < n2; base += (long)gridDim.x * tile) {
    d2 wv[U], pv[U], cv[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        wv[u] = (POL == 1) ? w2[i] : __builtin_nontemporal_load(&w2[i]);
        pv[u] = __builtin_nontemporal_load(&p2[i]); cv[u] = __builtin_nontemporal_load(&c2[i]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        d2 r = wv[u] - h * pv[u];
        if (POL == 0) __builtin_nontemporal_store(r, &w2[i]); else w2[i] = r;
        a0 = fma(r.x, cv[u].x, a0); a1 = fma(r.y, cv[u].y, a1);
      }
    }
  }
  double t = block_sum<BLK>(a0 + a1, s);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// V3: three links per pass (7 reads, 1 write, 6 dots)
template <int BLK, int U>
__global__ __launch_bounds__(BLK) void v3(double *w, const double *p1, const double *p2_, const double *p3_, const double *c1, const double *c2_, const double *c3_, double h1, double h2, double h3, double *part, long n2) {
  __shared__ double s[BLK / 64];
  d2 *w2 = (d2 *)w; const d2 *P1 = (const d2 *)p1, *P2 = (const d2 *)p2_, *P3 = (const d2 *)p3_, *C1 = (const d2 *)c1, *C2 = (const d2 *)c2_, *C3 = (const d2 *)c3_;
  double a = 0, b = 0, c = 0, g12 = 0, g13 = 0, g23 = 0;
  const long tile = (long)BLK * U;
  for (long base = (long)blockIdx.x * tile; base < n2; base += (long)gridDim.x * tile) {
    d2 wv[U], p1v[U], p2v[U], p3v[U], c1v[U], c2v[U], c3v[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) { wv[u] = __builtin_nontemporal_load(&w2[i]); p1v[u] = __builtin_nontemporal_load(&P1[i]); p2v[u] = __builtin_nontemporal_load(&P2[i]); p3v[u] = __builtin_nontemporal_load(&P3[i]);
        c1v[u] = __builtin_nontemporal_load(&C1[i]); c2v[u] = __builtin_nontemporal_load(&C2[i]); c3v[u] = __builtin_nontemporal_load(&C3[i]); }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        d2 r = wv[u] - h1 * p1v[u]; r = r - h2 * p2v[u]; r = r - h3 * p3v[u];
        __builtin_nontemporal_store(r, &w2[i]);
        a = fma(r.x, c1v[u].x, a); a = fma(r.y, c1v[u].y, a);
        b = fma(r.x, c2v[u].x, b); b = fma(r.y, c2v[u].y, b);
        c = fma(r.x, c3v[u].x, c); c = fma(r.y, c3v[u].y, c);
        g12 = fma(c1v[u].x, c2v[u].x, g12); g12 = fma(c1v[u].y, c2v[u].y, g12);
        g13 = fma(c1v[u].x, c3v[u].x, g13); g13 = fma(c1v[u].y, c3v[u].y, g13);
        g23 = fma(c2v[u].x, c3v[u].x, g23); g23 = fma(c2v[u].y, c3v[u].y, g23);
      }
    }
  }
  double t0 = block_sum<BLK>(a, s), t1 = block_sum<BLK>(b, s), t2 = block_sum<BLK>(c, s), t3 = block_sum<BLK>(g12, s), t4 = block_sum<BLK>(g13, s), t5 = block_sum<BLK>(g23, s);
  if (threadIdx.x == 0) { const int G = gridDim.x; part[blockIdx.x] = t0; part[G + blockIdx.x] = t1; part[2 * G + blockIdx.x] = t2; part[3 * G + blockIdx.x] = t3; part[4 * G + blockIdx.x] = t4; part[5 * G + blockIdx.x] = t5; }
}


// V4: four links per pass (9 reads, 1 write, 4 dots + 6 Gram entries), as k_mgs_block<4,4>
template <int BLK, int U>
__global__ __launch_bounds__(BLK) void v4(double *w, const double *pbase, const double *cbase, long ld, double *part, long n2) {
  __shared__ double s[BLK / 64];
  d2 *w2 = (d2 *)w;
  double acc[10];
  for (int q = 0; q < 10; q++) acc[q] = 0.0;
  const double h[4] = {1e-3, 2e-3, 3e-3, 4e-3};
  const long tile = (long)BLK * U;
  for (long base = (long)blockIdx.x * tile; base < n2; base += (long)gridDim.x * tile) {
    d2 wv[U], pv[4][U], cv[4][U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        wv[u] = __builtin_nontemporal_load(&w2[i]);
#pragma unroll
        for (int k = 0; k < 4; k++) pv[k][u] = __builtin_nontemporal_load((const d2 *)(pbase + k * ld) + i);
#pragma unroll
        for (int k = 0; k < 4; k++) cv[k][u] = __builtin_nontemporal_load((const d2 *)(cbase + k * ld) + i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        d2 r = wv[u];
#pragma unroll
        for (int k = 0; k < 4; k++) r = r - h[k] * pv[k][u];
        __builtin_nontemporal_store(r, &w2[i]);
        int q = 4;
#pragma unroll
        for (int a = 0; a < 4; a++) {
          acc[a] = fma(r.x, cv[a][u].x, acc[a]); acc[a] = fma(r.y, cv[a][u].y, acc[a]);
#pragma unroll
          for (int b = a + 1; b < 4; b++) { acc[q] = fma(cv[a][u].x, cv[b][u].x, acc[q]); acc[q] = fma(cv[a][u].y, cv[b][u].y, acc[q]); q++; }
        }
      }
    }
  }
  for (int q = 0; q < 10; q++) { double t = block_sum<BLK>(acc[q], s); if (threadIdx.x == 0) part[q * gridDim.x + blockIdx.x] = t; }
}

// VK: K links per pass (2K+1 reads, 1 write, K dots + K(K-1)/2 Gram entries)
template <int BLK, int U, int K>
__global__ __launch_bounds__(BLK) void vk(double *w, const double *pbase, const double *cbase, long ld, double *part, long n2) {
  __shared__ double s[BLK / 64];
  constexpr int NQ = K + K * (K - 1) / 2;
  d2 *w2 = (d2 *)w;
  double acc[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) acc[q] = 0.0;
  double h[K];
#pragma unroll
  for (int k = 0; k < K; k++) h[k] = 1e-3 * (k + 1);
  const long tile = (long)BLK * U;
  for (long base = (long)blockIdx.x * tile; base < n2; base += (long)gridDim.x * tile) {
    d2 wv[U], pv[K][U], cv[K][U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        wv[u] = __builtin_nontemporal_load(&w2[i]);
#pragma unroll
        for (int k = 0; k < K; k++) pv[k][u] = __builtin_nontemporal_load((const d2 *)(pbase + k * ld) + i);
#pragma unroll
        for (int k = 0; k < K; k++) cv[k][u] = __builtin_nontemporal_load((const d2 *)(cbase + k * ld) + i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        d2 r = wv[u];
#pragma unroll
        for (int k = 0; k < K; k++) r = r - h[k] * pv[k][u];
        __builtin_nontemporal_store(r, &w2[i]);
        int q = K;
#pragma unroll
        for (int a = 0; a < K; a++) {
          acc[a] = fma(r.x, cv[a][u].x, acc[a]); acc[a] = fma(r.y, cv[a][u].y, acc[a]);
#pragma unroll
          for (int b = a + 1; b < K; b++) { acc[q] = fma(cv[a][u].x, cv[b][u].x, acc[q]); acc[q] = fma(cv[a][u].y, cv[b][u].y, acc[q]); q++; }
        }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < NQ; q++) { double t = block_sum<BLK>(acc[q], s); if (threadIdx.x == 0) part[q * gridDim.x + blockIdx.x] = t; }
}

// VD: K links per pass without Gram entries (2K+1 reads, 1 write, K dots + 1 extra dot)
template <int BLK, int U, int K>
__global__ __launch_bounds__(BLK) void vd(double *w, const double *pbase, const double *cbase, long ld, double *part, long n2) {
  __shared__ double s[BLK / 64];
  d2 *w2 = (d2 *)w;
  double acc[K];
#pragma unroll
  for (int q = 0; q < K; q++) acc[q] = 0.0;
  double h[K];
#pragma unroll
  for (int k = 0; k < K; k++) h[k] = 1e-3 * (k + 1);
  const long tile = (long)BLK * U;
  for (long base = (long)blockIdx.x * tile; base < n2; base += (long)gridDim.x * tile) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) {
        d2 r = __builtin_nontemporal_load(&w2[i]);
        d2 pv[K], cv[K];
#pragma unroll
        for (int k = 0; k < K; k++) pv[k] = __builtin_nontemporal_load((const d2 *)(pbase + k * ld) + i);
#pragma unroll
        for (int k = 0; k < K; k++) cv[k] = __builtin_nontemporal_load((const d2 *)(cbase + k * ld) + i);
#pragma unroll
        for (int k = 0; k < K; k++) r = r - h[k] * pv[k];
        __builtin_nontemporal_store(r, &w2[i]);
#pragma unroll
        for (int a = 0; a < K; a++) { acc[a] = fma(r.x, cv[a].x, acc[a]); acc[a] = fma(r.y, cv[a].y, acc[a]); }
      }
    }
  }
#pragma unroll
  for (int q = 0; q < K; q++) { double t = block_sum<BLK>(acc[q], s); if (threadIdx.x == 0) part[q * gridDim.x + blockIdx.x] = t; }
}

// plain copy for calibration
template <int BLK, int U, bool NT>
__global__ __launch_bounds__(BLK) void vcopy(double *w, const double *p, long n2) {
  d2 *w2 = (d2 *)w; const d2 *p2 = (const d2 *)p;
  const long tile = (long)BLK * U;
  for (long base = (long)blockIdx.x * tile; base < n2; base += (long)gridDim.x * tile) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) { if (NT) __builtin_nontemporal_store(__builtin_nontemporal_load(&p2[i]), &w2[i]); else w2[i] = p2[i]; }
    }
  }
}
// read-only sum of 3 streams for calibration
template <int BLK, int U>
__global__ __launch_bounds__(BLK) void vread3(const double *w, const double *p, const double *c, double *part, long n2) {
  __shared__ double s[BLK / 64];
  const d2 *w2 = (const d2 *)w; const d2 *p2 = (const d2 *)p; const d2 *c2 = (const d2 *)c;
  double a0 = 0;
  const long tile = (long)BLK * U;
  for (long base = (long)blockIdx.x * tile; base < n2; base += (long)gridDim.x * tile) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long i = base + u * BLK + threadIdx.x;
      if (i < n2) { d2 x = w2[i], y = p2[i], z = c2[i]; a0 += x.x * y.x + z.x + x.y * y.y + z.y; }
    }
  }
  double t = block_sum<BLK>(a0, s);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

int main(int argc, char **argv) {
  long n = argc > 1 ? atol(argv[1]) : 67141636L;
  const long n2 = n / 2;
  const int NV = 8;
  std::vector<double *> v(NV);
  for (int i = 0; i < NV; i++) { CHECK(hipMalloc((void **)&v[i], sizeof(double) * n)); CHECK(hipMemset(v[i], 0, sizeof(double) * n)); }
  double *part; CHECK(hipMalloc((void **)&part, sizeof(double) * 65536 * 3));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int reps = 20;
  auto run = [&](const char *name, double bytes_per_elem, auto launch) {
    for (int r = 0; r < 3; r++) launch(r);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) launch(r);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    CHECK(hipGetLastError());
    double gbps = bytes_per_elem * n * reps / (ms * 1e-3) / 1e9;
    printf("%-44s %8.1f us  %7.1f GB/s  (%.1f%% of 8 TB/s)\n", name, ms * 1e3 / reps, gbps, gbps / 80.0);
  };
#define W(r) v[0]
#define P(r) v[1 + ((r) % 3)]
#define Cc(r) v[4 + ((r) % 3)]
  char nm[128];
  // vectors with leading dimension ld for the K+K sweeps
  double *big; const long ld = (n + 31) / 32 * 32;
  CHECK(hipMalloc((void **)&big, sizeof(double) * ld * 17)); CHECK(hipMemset(big, 0, sizeof(double) * ld * 17));
  if (argc > 2 && !strcmp(argv[2], "wide32")) {
    double *big3; CHECK(hipMalloc((void **)&big3, sizeof(double) * ld * 65)); CHECK(hipMemset(big3, 0, sizeof(double) * ld * 65));
    for (int G : {128, 256, 512}) {
      snprintf(nm, sizeof nm, "VD 16+16 dots only U1 G%d (272 B)", G); run(nm, 272, [&](int r) { hipLaunchKernelGGL((vd<256, 1, 16>), dim3(G), dim3(256), 0, 0, big3, big3 + ld, big3 + 33 * ld, ld, part, n2); });
      snprintf(nm, sizeof nm, "VD 24+24 dots only U1 G%d (400 B)", G); run(nm, 400, [&](int r) { hipLaunchKernelGGL((vd<256, 1, 24>), dim3(G), dim3(256), 0, 0, big3, big3 + ld, big3 + 33 * ld, ld, part, n2); });
      snprintf(nm, sizeof nm, "VD 32+32 dots only U1 G%d (528 B)", G); run(nm, 528, [&](int r) { hipLaunchKernelGGL((vd<256, 1, 32>), dim3(G), dim3(256), 0, 0, big3, big3 + ld, big3 + 33 * ld, ld, part, n2); });
    }
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "wide16")) {
    double *big2; CHECK(hipMalloc((void **)&big2, sizeof(double) * ld * 33)); CHECK(hipMemset(big2, 0, sizeof(double) * ld * 33));
    for (int G : {96, 112, 120, 128, 192, 256}) {
      snprintf(nm, sizeof nm, "VD 8+8 dots only U1 G%d (144 B)", 2 * G); run(nm, 144, [&](int r) { hipLaunchKernelGGL((vd<256, 1, 8>), dim3(2 * G), dim3(256), 0, 0, big2, big2 + ld, big2 + 17 * ld, ld, part, n2); });
      snprintf(nm, sizeof nm, "VD 16+16 dots only U1 G%d (272 B)", G); run(nm, 272, [&](int r) { hipLaunchKernelGGL((vd<256, 1, 16>), dim3(G), dim3(256), 0, 0, big2, big2 + ld, big2 + 17 * ld, ld, part, n2); });
      snprintf(nm, sizeof nm, "VD 16+16 dots only BLK512 G%d (272 B)", G); run(nm, 272, [&](int r) { hipLaunchKernelGGL((vd<512, 1, 16>), dim3(G), dim3(512), 0, 0, big2, big2 + ld, big2 + 17 * ld, ld, part, n2); });
    }
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "wide")) {
    // bytes per link: K=4: 20, K=6: 18.7, K=8: 18
    for (int G : {128, 192, 256, 384, 512}) {
      snprintf(nm, sizeof nm, "VK 4+4 U1 G%d (80 B)", G); run(nm, 80, [&](int r) { hipLaunchKernelGGL((vk<256, 1, 4>), dim3(G), dim3(256), 0, 0, big, big + ld, big + 9 * ld, ld, part, n2); });
      snprintf(nm, sizeof nm, "VK 6+6 U1 G%d (112 B)", G); run(nm, 112, [&](int r) { hipLaunchKernelGGL((vk<256, 1, 6>), dim3(G), dim3(256), 0, 0, big, big + ld, big + 9 * ld, ld, part, n2); });
      snprintf(nm, sizeof nm, "VK 8+8 U1 G%d (144 B)", G); run(nm, 144, [&](int r) { hipLaunchKernelGGL((vk<256, 1, 8>), dim3(G), dim3(256), 0, 0, big, big + ld, big + 9 * ld, ld, part, n2); });
    }
    // launch-shape variants of the 8+8 sweep
    for (int G : {128, 256, 512}) {
      snprintf(nm, sizeof nm, "VK 8+8 BLK512 U1 G%d (144 B)", G); run(nm, 144, [&](int r) { hipLaunchKernelGGL((vk<512, 1, 8>), dim3(G), dim3(512), 0, 0, big, big + ld, big + 9 * ld, ld, part, n2); });
      snprintf(nm, sizeof nm, "VK 8+8 BLK128 U1 G%d (144 B)", 2 * G); run(nm, 144, [&](int r) { hipLaunchKernelGGL((vk<128, 1, 8>), dim3(2 * G), dim3(128), 0, 0, big, big + ld, big + 9 * ld, ld, part, n2); });
      snprintf(nm, sizeof nm, "VK 8+8 BLK256 U2 G%d (144 B)", G); run(nm, 144, [&](int r) { hipLaunchKernelGGL((vk<256, 2, 8>), dim3(G), dim3(256), 0, 0, big, big + ld, big + 9 * ld, ld, part, n2); });
    }
    for (int G : {224, 240, 256, 272, 288, 320}) {
      snprintf(nm, sizeof nm, "VK 8+8 U1 G%d (144 B)", G); run(nm, 144, [&](int r) { hipLaunchKernelGGL((vk<256, 1, 8>), dim3(G), dim3(256), 0, 0, big, big + ld, big + 9 * ld, ld, part, n2); });
    }
    return 0;
  }
  run("V1 single link NT U4 G1024 (ref)", 32, [&](int r) { hipLaunchKernelGGL((v1<256, 4, true>), dim3(1024), dim3(256), 0, 0, v[0], P(r), Cc(r), 1e-3, part, n2); });
  for (int G : {256, 384, 512, 640, 768, 1024, 1536, 2048}) {
    snprintf(nm, sizeof nm, "V4 4+4 sweep U1 G%d", G); run(nm, 80, [&](int r) { hipLaunchKernelGGL((v4<256, 1>), dim3(G), dim3(256), 0, 0, big, big + ld, big + 5 * ld, ld, part, n2); });
    snprintf(nm, sizeof nm, "V4 4+4 sweep U2 G%d", G); run(nm, 80, [&](int r) { hipLaunchKernelGGL((v4<256, 2>), dim3(G), dim3(256), 0, 0, big, big + ld, big + 5 * ld, ld, part, n2); });
    snprintf(nm, sizeof nm, "V4 4+4 sweep U3 G%d", G); run(nm, 80, [&](int r) { hipLaunchKernelGGL((v4<256, 3>), dim3(G), dim3(256), 0, 0, big, big + ld, big + 5 * ld, ld, part, n2); });
  }
  for (int G : {128, 256, 512}) {
    snprintf(nm, sizeof nm, "V4 4+4 sweep BLK512 U1 G%d", G); run(nm, 80, [&](int r) { hipLaunchKernelGGL((v4<512, 1>), dim3(G), dim3(512), 0, 0, big, big + ld, big + 5 * ld, ld, part, n2); });
    snprintf(nm, sizeof nm, "V4 4+4 sweep BLK512 U2 G%d", G); run(nm, 80, [&](int r) { hipLaunchKernelGGL((v4<512, 2>), dim3(G), dim3(512), 0, 0, big, big + ld, big + 5 * ld, ld, part, n2); });
  }
  return 0;
}
