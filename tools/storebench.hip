// Micro-benchmark: what does the ONE written vector of a one-sweep Arnoldi kernel cost, by
// cache policy of its stores?  K read streams on a rows x P grid in the sweep's access pattern
// (workgroups own a column strip and walk down a band of rows, 8 bytes per lane, non-temporal
// loads) plus one write stream whose stores carry the policy bits under test.
//   hipcc -O3 --offload-arch=gfx950 tools/storebench.hip -o tools/bin/storebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// POL: 0 no store, 1 plain, 2 nt, 3 sc1, 4 sc0 sc1, 5 sc0 sc1 nt, 6 sc1 nt, 7 sc0, 8 sc0 nt
template <int POL>
__device__ __forceinline__ void store_pol(double *p, double v) {
  if (POL == 1) { *p = v; return; }
  if (POL == 2) { __builtin_nontemporal_store(v, p); return; }
  if (POL == 3) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  if (POL == 4) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  if (POL == 5) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
  if (POL == 6) asm volatile("global_store_dwordx2 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  if (POL == 7) asm volatile("global_store_dwordx2 %0, %1, off sc0" ::"v"(p), "v"(v) : "memory");
  if (POL == 8) asm volatile("global_store_dwordx2 %0, %1, off sc0 nt" ::"v"(p), "v"(v) : "memory");
}

// BURST rows of the result are kept in registers and stored together
template <int K, int POL, int BURST, bool TILED = false>
__global__ __launch_bounds__(256) void strip(const double *__restrict__ V, long ld, int rows, int P, int W, int nstrips,
                                             int R, double *__restrict__ out, double *__restrict__ Wr) {
  const int t = threadIdx.x;
  const int strip = blockIdx.x % nstrips, band = blockIdx.x / nstrips;
  const int cs = strip * W;
  const int c = cs + t;
  const bool ok = c < P && t < W;
  const int rs = band * R, re = rs + R < rows ? rs + R : rows;
  double acc[K];
  double hold[BURST];
#pragma unroll
  for (int k = 0; k < K; k++) acc[k] = 0.0;
#pragma unroll 4
  for (int rr = rs; rr < re; rr++) {
    const long idx = (long)rr * P + c;
    double s = 0.0;
    if (ok) {
      double v[K];
#pragma unroll
      for (int k = 0; k < K; k++) v[k] = __builtin_nontemporal_load(&V[k * ld + idx]);
#pragma unroll
      for (int k = 0; k < K; k++) { acc[k] += v[k]; s += v[k]; }
    }
    if (POL > 0) {
      const int j = (rr - rs) % BURST;
#pragma unroll
      for (int q = 0; q < BURST; q++) if (q == j) hold[q] = s;
      if ((j == BURST - 1 || rr == re - 1) && ok) {
#pragma unroll
        for (int q = 0; q < BURST; q++)
          if (q <= j) store_pol<POL>(TILED ? &Wr[((long)strip * rows + (rr - j + q)) * 256 + t] : &Wr[(long)(rr - j + q) * P + c], hold[q]);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < K; k++) s += acc[k];
  if (s == 12345.678) out[blockIdx.x] = s;
}

template <int K, int POL, int BURST, bool TILED = false>
void run(const char *name, const double *V, long ld, int rows, int P, double *out, double *Wr) {
  const int W = 249, nstrips = (P + W - 1) / W;
  const int nb = 1024 / nstrips;
  const int R = (rows + nb - 1) / nb, nbands = (rows + R - 1) / R;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((strip<K, POL, BURST, TILED>), dim3(nstrips * nbands), dim3(256), 0, 0, V, ld, rows, P, W, nstrips, R, out, Wr);
  CHECK(hipDeviceSynchronize());
  const int reps = 5;
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((strip<K, POL, BURST, TILED>), dim3(nstrips * nbands), dim3(256), 0, 0, V, ld, rows, P, W, nstrips, R, out, Wr);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipGetLastError());
  const double bytes = 8.0 * (K + (POL > 0 ? 1 : 0)) * (double)rows * P;
  printf("K=%2d burst %2d  %-14s %8.1f us %7.1f GB/s\n", K, BURST, name, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e9);
}

template <int K, int BURST>
void all(const double *V, long ld, int rows, int P, double *out, double *Wr) {
  run<K, 0, BURST>("reads only", V, ld, rows, P, out, Wr);
  run<K, 1, BURST>("plain", V, ld, rows, P, out, Wr);
  run<K, 2, BURST>("nt", V, ld, rows, P, out, Wr);
  run<K, 3, BURST>("sc1", V, ld, rows, P, out, Wr);
  run<K, 4, BURST>("sc0 sc1", V, ld, rows, P, out, Wr);
  run<K, 5, BURST>("sc0 sc1 nt", V, ld, rows, P, out, Wr);
  run<K, 6, BURST>("sc1 nt", V, ld, rows, P, out, Wr);
  run<K, 7, BURST>("sc0", V, ld, rows, P, out, Wr);
  run<K, 8, BURST>("sc0 nt", V, ld, rows, P, out, Wr);
}

int main() {
  const int rows = 8194, P = 8194;
  const long n = (long)rows * P, ld = (n + 31) / 32 * 32;
  double *V, *out, *Wr;
  CHECK(hipMalloc((void **)&V, sizeof(double) * ld * 32)); CHECK(hipMemset(V, 0, sizeof(double) * ld * 32));
  CHECK(hipMalloc((void **)&out, sizeof(double) * 65536));
  CHECK(hipMalloc((void **)&Wr, sizeof(double) * ld * 2)); CHECK(hipMemset(Wr, 0, sizeof(double) * ld * 2));
  run<4, 0, 16>("reads only", V, ld, rows, P, out, Wr);
  run<4, 2, 16>("nt", V, ld, rows, P, out, Wr);
  run<4, 2, 16, true>("nt tiled", V, ld, rows, P, out, Wr);
  run<4, 1, 16, true>("plain tiled", V, ld, rows, P, out, Wr);
  run<16, 0, 16>("reads only", V, ld, rows, P, out, Wr);
  run<16, 2, 16>("nt", V, ld, rows, P, out, Wr);
  run<16, 2, 16, true>("nt tiled", V, ld, rows, P, out, Wr);
  run<16, 2, 1, true>("nt tiled", V, ld, rows, P, out, Wr);
  run<32, 0, 16>("reads only", V, ld, rows, P, out, Wr);
  run<32, 2, 16>("nt", V, ld, rows, P, out, Wr);
  run<32, 2, 16, true>("nt tiled", V, ld, rows, P, out, Wr);
  return 0;
}
