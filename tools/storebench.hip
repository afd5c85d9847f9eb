// Micro-benchmark: what does the ONE written vector of a one-sweep Arnoldi kernel cost, by
// cache policy of its stores?  K read streams on a rows x P grid in the sweep's access pattern
// (workgroups own a column strip and walk down a band of rows, 8 bytes per lane, non-temporal
// loads) plus one write stream whose stores carry the policy bits under test.
//   hipcc -O3 --offload-arch=gfx950 tools/storebench.hip -o tools/bin/storebench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
// LAYOUT of the written vector: 0 the grid's own, 1 a workgroup's rows contiguous, 2 every row of a workgroup onto the same 2 KB
// (the stores are issued, the bytes never leave the cache: separates instruction-stream cost from memory-system cost)
template <int K, int POL, int BURST, int TILED = 0, int RT = 0>
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
    // RT: the read vectors are stored strip by strip (a workgroup's rows contiguous: 256 elements each)
    // RT 2: as the sweeps would read such a layout -- 252 owned columns per strip, lanes 0, 1, 254, 255 fetch the
    // neighbouring strips' edge columns
    long idx = RT ? ((long)strip * rows + rr) * 256 + t : (long)rr * P + c;
    if (RT == 2) {
      int s2 = strip, o2 = t - 2;
      if (t < 2 && strip > 0) { s2 = strip - 1; o2 = 250 + t; }
      if (t < 2 && strip == 0) o2 = 0;
      if (t >= 254 && strip + 1 < nstrips) { s2 = strip + 1; o2 = t - 254; }
      if (t >= 254 && strip + 1 >= nstrips) o2 = 251;
      idx = ((long)s2 * rows + rr) * 256 + o2;
    }
    double s = 0.0;
    if (ok || RT == 2) {
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
          if (q <= j) store_pol<POL>(TILED == 2 ? &Wr[(long)blockIdx.x * 256 + t] : TILED == 1 ? &Wr[((long)strip * rows + (rr - j + q)) * 256 + t] : &Wr[(long)(rr - j + q) * P + c], hold[q]);
      }
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < K; k++) s += acc[k];
  if (s == 12345.678) out[blockIdx.x] = s;
}

// Latency probe: per row all K loads are issued and waited for (timed), every 16 rows 16 non-temporal stores are issued
// and waited for (timed).  out[0..3] receive sums of cycles and counts over all waves (lane 0 of each).
template <int K, int TILED>
__global__ __launch_bounds__(256) void strip_lat(const double *__restrict__ V, long ld, int rows, int P, int W, int nstrips,
                                                 int R, unsigned long long *__restrict__ out, double *__restrict__ Wr) {
  const int t = threadIdx.x;
  const int strip = blockIdx.x % nstrips, band = blockIdx.x / nstrips;
  const int c = strip * W + t;
  const bool ok = c < P && t < W;
  const int rs = band * R, re = rs + R < rows ? rs + R : rows;
  double hold[16];
  double tot = 0.0;
  unsigned long long tl = 0, ts = 0, nl = 0, ns = 0;
  for (int rr = rs; rr < re; rr++) {
    const long idx = (long)rr * P + c;
    double s = 0.0;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t0 = clock64();
    if (ok) {
      double v[K];
#pragma unroll
      for (int k = 0; k < K; k++) v[k] = __builtin_nontemporal_load(&V[k * ld + idx]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < K; k++) s += v[k];
    }
    const long long t1 = clock64();
    tl += t1 - t0; nl++;
    tot += s;
    const int j = (rr - rs) % 16;
#pragma unroll
    for (int q = 0; q < 16; q++) if (q == j) hold[q] = s;
    if (j == 15 || rr == re - 1) {
      const long long t2 = clock64();
      if (ok) {
#pragma unroll
        for (int q = 0; q < 16; q++)
          if (q <= j) __builtin_nontemporal_store(hold[q], TILED == 2 ? &Wr[(long)blockIdx.x * 256 + t] : &Wr[(long)(rr - j + q) * P + c]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long long t3 = clock64();
      ts += t3 - t2; ns++;
    }
  }
  if ((t & 63) == 0) {
    atomicAdd(&out[0], tl); atomicAdd(&out[1], nl); atomicAdd(&out[2], ts); atomicAdd(&out[3], ns);
  }
  if (tot == 12345.678) Wr[0] = tot;
}

template <int K, int TILED>
void run_lat(const char *name, const double *V, long ld, int rows, int P, double *out, double *Wr) {
  const int W = 249, nstrips = (P + W - 1) / W;
  const int nb = 1024 / nstrips;
  const int R = (rows + nb - 1) / nb, nbands = (rows + R - 1) / R;
  unsigned long long *o = (unsigned long long *)out, h[4];
  CHECK(hipMemset(o, 0, 32));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((strip_lat<K, TILED>), dim3(nstrips * nbands), dim3(256), 0, 0, V, ld, rows, P, W, nstrips, R, o, Wr);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipMemcpy(h, o, 32, hipMemcpyDeviceToHost));
  printf("K=%2d latency probe %-12s %8.1f us: a row of %d loads waited %.0f clocks, a burst of 16 stores waited %.0f clocks (shader clocks)\n", K, name,
         ms * 1e3, K, (double)h[0] / h[1], (double)h[2] / h[3]);
}


// Stores on a common clock: the last rows of the result wait in a register ring of BURST rows and are stored when the
// device-wide real-time counter (100 MHz) crosses a multiple of `period` (or the ring is full), so that all workgroups
// write at the same moments and the memory controllers see their writes as a few dense batches instead of a trickle.
template <int K, int BURST>
__global__ __launch_bounds__(256) void strip_tick(const double *__restrict__ V, long ld, int rows, int P, int W, int nstrips,
                                                  int R, double *__restrict__ out, double *__restrict__ Wr, long period,
                                                  unsigned long long *__restrict__ stats) {
  const int t = threadIdx.x;
  const int strip = blockIdx.x % nstrips, band = blockIdx.x / nstrips;
  const int c = strip * W + t;
  const bool ok = c < P && t < W;
  const int rs = band * R, re = rs + R < rows ? rs + R : rows;
  double acc[K];
  double hold[BURST];
#pragma unroll
  for (int k = 0; k < K; k++) acc[k] = 0.0;
  int p0 = rs;  // first row not stored yet
  long last = period > 0 ? (long)(wall_clock64() / period) : 0;
  unsigned long long nflush = 0, nfull = 0;
#pragma unroll 2
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
    const int j = (rr - rs) % BURST;
#pragma unroll
    for (int q = 0; q < BURST; q++) if (q == j) hold[q] = s;
    const int pending = rr - p0 + 1;
    bool flush = pending == BURST || rr == re - 1;
    if (period > 0) {
      const long now = (long)(wall_clock64() / period);
      if (now != last) { flush = true; last = now; } else if (flush) nfull++;
    }
    if (flush) {
      nflush++;
      if (ok) {
#pragma unroll
        for (int q = 0; q < BURST; q++) {
          // ring slot q holds row p0 + ((q - (p0 - rs)) mod BURST)
          const int row = p0 + (((q - (p0 - rs)) % BURST) + BURST) % BURST;
          if (row <= rr) __builtin_nontemporal_store(hold[q], &Wr[(long)row * P + c]);
        }
      }
      p0 = rr + 1;
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < K; k++) s += acc[k];
  if (s == 12345.678) out[blockIdx.x] = s;
  if (t == 0) { atomicAdd(&stats[0], nflush); atomicAdd(&stats[1], nfull); }
}

template <int K, int BURST>
void run_tick(const char *name, const double *V, long ld, int rows, int P, double *out, double *Wr, long period) {
  const int W = 249, nstrips = (P + W - 1) / W;
  const int nb = 1024 / nstrips;
  const int R = (rows + nb - 1) / nb, nbands = (rows + R - 1) / R;
  unsigned long long *st = (unsigned long long *)(out + 32768), h[2];
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((strip_tick<K, BURST>), dim3(nstrips * nbands), dim3(256), 0, 0, V, ld, rows, P, W, nstrips, R, out, Wr, period, st);
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemset(st, 0, 16));
  const int reps = 5;
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((strip_tick<K, BURST>), dim3(nstrips * nbands), dim3(256), 0, 0, V, ld, rows, P, W, nstrips, R, out, Wr, period, st);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipGetLastError());
  CHECK(hipMemcpy(h, st, 16, hipMemcpyDeviceToHost));
  const double bytes = 8.0 * (K + 1) * (double)rows * P;
  printf("K=%2d ring %2d  %-20s period %6ld x 10 ns %8.1f us %7.1f GB/s  (flushes per workgroup %.1f, of them ring full %.1f)\n", K, BURST, name, period,
         ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e9, (double)h[0] / reps / (nstrips * nbands), (double)h[1] / reps / (nstrips * nbands));
}

// Reads only, strip-major, at a chosen occupancy and a chosen number of requests in flight per wavefront: dynamic LDS
// limits the resident workgroups per compute unit (the one-sweep kernels run 8 wavefronts per compute unit), and the
// loop keeps DEPTH rows of K requests in flight (it requests row r + DEPTH before it uses row r).
template <int K, int DEPTH>
__global__ __launch_bounds__(256) void strip_occ(const double *__restrict__ V, long ld, int rows, int nstrips, int R,
                                                 double *__restrict__ out) {
  extern __shared__ double dyn[];
  const int t = threadIdx.x;
  const int strip = blockIdx.x % nstrips, band = blockIdx.x / nstrips;
  const int rs = band * R, re = rs + R < rows ? rs + R : rows;
  double acc = 0.0;
  double v[DEPTH][K];
  auto idx = [&](int r) -> long { return ((long)strip * rows + (r < rows ? r : rows - 1)) * 256 + t; };
#pragma unroll
  for (int d = 0; d < DEPTH; d++)
#pragma unroll
    for (int k = 0; k < K; k++) v[d][k] = __builtin_nontemporal_load(&V[k * ld + idx(rs + d)]);
  for (int rr = rs; rr < re; rr += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      const long in = idx(rr + d + DEPTH);
#pragma unroll
      for (int k = 0; k < K; k++) {
        acc += v[d][k];
        v[d][k] = __builtin_nontemporal_load(&V[k * ld + in]);
      }
    }
  }
  if (acc == 12345.678) { out[blockIdx.x] = acc; dyn[t] = acc; }
}

template <int K, int DEPTH>
void run_occ(const double *V, long ld, int rows, int P, double *out, int wg_per_cu) {
  const int nstrips = 33;
  const int nb = 1024 / nstrips;
  const int R = (rows + nb - 1) / nb, nbands = (rows + R - 1) / R;
  const size_t lds = wg_per_cu >= 8 ? 0 : (size_t)(160 * 1024 / wg_per_cu) - 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&strip_occ<K, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((strip_occ<K, DEPTH>), dim3(nstrips * nbands), dim3(256), lds, 0, V, ld, rows, nstrips, R, out);
  CHECK(hipDeviceSynchronize());
  const int reps = 5;
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((strip_occ<K, DEPTH>), dim3(nstrips * nbands), dim3(256), lds, 0, V, ld, rows, nstrips, R, out);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipGetLastError());
  const double bytes = 8.0 * K * (double)rows * nstrips * 256;
  printf("K=%2d rows in flight %d (%3d requests per wavefront)  workgroups per compute unit %d  %8.1f us %7.1f GB/s (bytes moved)\n", K, DEPTH, K * DEPTH,
         wg_per_cu, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e9);
}

template <int K, int POL, int BURST, int TILED = 0, int RT = 0>
void run(const char *name, const double *V, long ld, int rows, int P, double *out, double *Wr) {
  const int W = 249, nstrips = (P + W - 1) / W;
  const int nb = 1024 / nstrips;
  const int R = (rows + nb - 1) / nb, nbands = (rows + R - 1) / R;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((strip<K, POL, BURST, TILED, RT>), dim3(nstrips * nbands), dim3(256), 0, 0, V, ld, rows, P, W, nstrips, R, out, Wr);
  CHECK(hipDeviceSynchronize());
  const int reps = 5;
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((strip<K, POL, BURST, TILED, RT>), dim3(nstrips * nbands), dim3(256), 0, 0, V, ld, rows, P, W, nstrips, R, out, Wr);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipGetLastError());
  const double bytes = 8.0 * (K + (POL > 0 ? 1 : 0)) * (double)rows * P;
  printf("K=%2d burst %2d  %-32s %8.1f us %7.1f GB/s\n", K, BURST, name, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e9);
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

int main(int argc, char **argv) {
  const int rows = 8194, P = 8194;
  const long n = (long)rows * P, ld = (n + 31) / 32 * 32;
  double *V, *out, *Wr;
  CHECK(hipMalloc((void **)&V, sizeof(double) * ld * 32)); CHECK(hipMemset(V, 0, sizeof(double) * ld * 32));
  CHECK(hipMalloc((void **)&out, sizeof(double) * 65536));
  CHECK(hipMalloc((void **)&Wr, sizeof(double) * ld * 2)); CHECK(hipMemset(Wr, 0, sizeof(double) * ld * 2));
  if (argc > 1 && !strcmp(argv[1], "pmc")) {  // few distinct kernels for a counter pass
    run<16, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<16, 2, 16>("nt", V, ld, rows, P, out, Wr);
    run<16, 1, 16>("plain", V, ld, rows, P, out, Wr);
    run<16, 2, 16, 2>("nt same 2 KB", V, ld, rows, P, out, Wr);
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "memtype")) {  // the written vector, then the read vectors as well, in uncached / fine-grained memory
    double *Wu, *Wf, *Vu;
    CHECK(hipExtMallocWithFlags((void **)&Wu, sizeof(double) * ld * 2, hipDeviceMallocUncached)); CHECK(hipMemset(Wu, 0, sizeof(double) * ld * 2));
    CHECK(hipExtMallocWithFlags((void **)&Wf, sizeof(double) * ld * 2, hipDeviceMallocFinegrained)); CHECK(hipMemset(Wf, 0, sizeof(double) * ld * 2));
    CHECK(hipExtMallocWithFlags((void **)&Vu, sizeof(double) * ld * 16, hipDeviceMallocUncached)); CHECK(hipMemset(Vu, 0, sizeof(double) * ld * 16));
    run<16, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<16, 2, 16>("nt", V, ld, rows, P, out, Wr);
    run<16, 2, 16>("nt -> uncached", V, ld, rows, P, out, Wu);
    run<16, 1, 16>("plain -> uncached", V, ld, rows, P, out, Wu);
    run<16, 2, 16>("nt -> fine-grained", V, ld, rows, P, out, Wf);
    run<16, 1, 16>("plain -> fine-grained", V, ld, rows, P, out, Wf);
    run<16, 0, 16>("uncached reads only", Vu, ld, rows, P, out, Wr);
    run<16, 2, 16>("uncached reads, nt -> uncached", Vu, ld, rows, P, out, Wu);
    run<4, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<4, 2, 16>("nt", V, ld, rows, P, out, Wr);
    run<4, 2, 16>("nt -> uncached", V, ld, rows, P, out, Wu);
    run<4, 1, 16>("plain -> uncached", V, ld, rows, P, out, Wu);
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "readtiled")) {  // the read vectors strip by strip as well
    const long ldt = 34L * rows * 256;  // 33 strips of 256 stored columns
    double *Vt, *Wt;
    CHECK(hipMalloc((void **)&Vt, sizeof(double) * ldt * 32)); CHECK(hipMemset(Vt, 0, sizeof(double) * ldt * 32));
    CHECK(hipMalloc((void **)&Wt, sizeof(double) * ldt)); CHECK(hipMemset(Wt, 0, sizeof(double) * ldt));
    run<4, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<4, 0, 16, 0, 1>("reads only, strip-major", Vt, ldt, rows, P, out, Wt);
    run<4, 2, 16>("nt", V, ld, rows, P, out, Wr);
    run<4, 2, 16, 1, 1>("nt, all strip-major", Vt, ldt, rows, P, out, Wt);
    run<16, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<16, 0, 16, 0, 1>("reads only, strip-major", Vt, ldt, rows, P, out, Wt);
    run<16, 2, 16>("nt", V, ld, rows, P, out, Wr);
    run<16, 2, 16, 1, 1>("nt, all strip-major", Vt, ldt, rows, P, out, Wt);
    run<16, 2, 1, 1, 1>("nt, all strip-major", Vt, ldt, rows, P, out, Wt);
    run<32, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<32, 0, 16, 0, 1>("reads only, strip-major", Vt, ldt, rows, P, out, Wt);
    run<32, 2, 16>("nt", V, ld, rows, P, out, Wr);
    run<32, 2, 16, 1, 1>("nt, all strip-major", Vt, ldt, rows, P, out, Wt);
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "tick")) {  // stores of all workgroups on a common clock
    run<4, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<4, 2, 16>("nt", V, ld, rows, P, out, Wr);
    for (long per : {0L, 2000L, 5000L, 10000L, 20000L}) run_tick<4, 32>("nt on the clock", V, ld, rows, P, out, Wr, per);
    run<16, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<16, 2, 16>("nt", V, ld, rows, P, out, Wr);
    for (long per : {0L, 5000L, 10000L, 20000L, 40000L}) run_tick<16, 32>("nt on the clock", V, ld, rows, P, out, Wr, per);
    run<32, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<32, 2, 16>("nt", V, ld, rows, P, out, Wr);
    for (long per : {0L, 10000L, 20000L, 40000L, 80000L}) run_tick<32, 32>("nt on the clock", V, ld, rows, P, out, Wr, per);
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "readhalo")) {  // strip-major reads with the edge columns taken from the neighbouring strips
    const long ldt = 34L * rows * 256;
    double *Vt, *Wt;
    CHECK(hipMalloc((void **)&Vt, sizeof(double) * ldt * 32)); CHECK(hipMemset(Vt, 0, sizeof(double) * ldt * 32));
    CHECK(hipMalloc((void **)&Wt, sizeof(double) * ldt)); CHECK(hipMemset(Wt, 0, sizeof(double) * ldt));
    run<16, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<16, 0, 16, 0, 1>("reads only, strip-major", Vt, ldt, rows, P, out, Wt);
    run<16, 0, 16, 0, 2>("reads only, strip-major + edges", Vt, ldt, rows, P, out, Wt);
    run<16, 2, 16, 1, 2>("nt, strip-major + edges", Vt, ldt, rows, P, out, Wt);
    run<32, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<32, 0, 16, 0, 1>("reads only, strip-major", Vt, ldt, rows, P, out, Wt);
    run<32, 0, 16, 0, 2>("reads only, strip-major + edges", Vt, ldt, rows, P, out, Wt);
    run<32, 2, 16, 1, 2>("nt, strip-major + edges", Vt, ldt, rows, P, out, Wt);
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "occ")) {  // strip-major reads by occupancy and requests in flight
    const long ldt = 34L * rows * 256;
    double *Vt;
    CHECK(hipMalloc((void **)&Vt, sizeof(double) * ldt * 32)); CHECK(hipMemset(Vt, 0, sizeof(double) * ldt * 32));
    for (int w : {1, 2, 4, 8}) {
      run_occ<16, 1>(Vt, ldt, rows, P, out, w);
      run_occ<16, 2>(Vt, ldt, rows, P, out, w);
      run_occ<16, 3>(Vt, ldt, rows, P, out, w);
      run_occ<32, 1>(Vt, ldt, rows, P, out, w);
    }
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "lat")) {
    run_lat<4, 0>("grid", V, ld, rows, P, out, Wr); run_lat<4, 2>("same 2 KB", V, ld, rows, P, out, Wr);
    run_lat<16, 0>("grid", V, ld, rows, P, out, Wr); run_lat<16, 2>("same 2 KB", V, ld, rows, P, out, Wr);
    run_lat<32, 0>("grid", V, ld, rows, P, out, Wr); run_lat<32, 2>("same 2 KB", V, ld, rows, P, out, Wr);
    return 0;
  }
  if (argc > 1 && !strcmp(argv[1], "same")) {
    run<4, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<4, 2, 16>("nt", V, ld, rows, P, out, Wr);
    run<4, 2, 16, 2>("nt same 2 KB", V, ld, rows, P, out, Wr);
    run<4, 1, 16, 2>("plain same 2 KB", V, ld, rows, P, out, Wr);
    run<16, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<16, 2, 16>("nt", V, ld, rows, P, out, Wr);
    run<16, 2, 16, 2>("nt same 2 KB", V, ld, rows, P, out, Wr);
    run<16, 1, 16, 2>("plain same 2 KB", V, ld, rows, P, out, Wr);
    run<16, 1, 1, 2>("plain same 2 KB", V, ld, rows, P, out, Wr);
    run<32, 0, 16>("reads only", V, ld, rows, P, out, Wr);
    run<32, 2, 16>("nt", V, ld, rows, P, out, Wr);
    run<32, 2, 16, 2>("nt same 2 KB", V, ld, rows, P, out, Wr);
    run<32, 1, 16, 2>("plain same 2 KB", V, ld, rows, P, out, Wr);
    return 0;
  }
  run<4, 0, 16>("reads only", V, ld, rows, P, out, Wr);
  run<4, 2, 16>("nt", V, ld, rows, P, out, Wr);
  run<4, 2, 16, 1>("nt tiled", V, ld, rows, P, out, Wr);
  run<4, 1, 16, 1>("plain tiled", V, ld, rows, P, out, Wr);
  run<16, 0, 16>("reads only", V, ld, rows, P, out, Wr);
  run<16, 2, 16>("nt", V, ld, rows, P, out, Wr);
  run<16, 2, 16, 1>("nt tiled", V, ld, rows, P, out, Wr);
  run<16, 2, 1, 1>("nt tiled", V, ld, rows, P, out, Wr);
  run<32, 0, 16>("reads only", V, ld, rows, P, out, Wr);
  run<32, 2, 16>("nt", V, ld, rows, P, out, Wr);
  run<32, 2, 16, 1>("nt tiled", V, ld, rows, P, out, Wr);
  return 0;
}
