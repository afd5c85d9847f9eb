// Micro-benchmark of the access pattern of the one-sweep Arnoldi kernel: K read streams on a
// rows x P grid, workgroups own a column strip and walk down a band of rows.
//   hipcc -O3 --offload-arch=gfx950 tools/stripbench.hip -o tools/bin/stripbench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef double d2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// VEC = 1: 8 bytes per lane, strip of BLK - 2 owned columns starting one column early (unaligned);
// VEC = 2: 16 bytes per lane, strip of 2 * BLK columns starting at an even column.
// MODE 4 (VEC = 1): strip-major vectors -- strip s is a block of rows x BLK values, a workgroup reads whole
// aligned rows of BLK values and writes positions 1 .. W + 2 of them (the one-sweep kernels on a strip-major basis).
template <int K, int VEC, int BLK, bool BARRIER, int NW = 0, bool NT = false, int MODE = 0>
__global__ __launch_bounds__(BLK) void strip(const double *__restrict__ V, long ld, int rows, int P, int W, int nstrips,
                                             int R, double *__restrict__ out, double *__restrict__ Wr = nullptr) {
  __shared__ double ring[BLK * VEC];
  const int t = threadIdx.x;
  const int strip = blockIdx.x % nstrips, band = blockIdx.x / nstrips;
  const int cs = strip * W;
  const int c = VEC == 1 ? cs - 1 + t : cs + 2 * t;
  const bool ok = MODE == 4 ? true : (c >= 0 && c + VEC <= P && (VEC == 1 ? t <= W + 1 : 2 * t < W));
  const int rs = band * R, re = rs + R < rows ? rs + R : rows;
  double acc[K];
  double held = 0.0;
  double hold8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int k = 0; k < K; k++) acc[k] = 0.0;
#pragma unroll 8
  for (int rr = rs; rr < re; rr++) {
    const long idx = MODE == 4 ? ((long)strip * rows + rr) * BLK + t : (long)rr * P + c;
    double s = 0.0;
    if (ok) {
      if (VEC == 1) {
        double v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = __builtin_nontemporal_load(&V[k * ld + idx]);
#pragma unroll
        for (int k = 0; k < K; k++) { acc[k] += v[k]; s += v[k]; }
      } else {
        d2 v[K];
#pragma unroll
        for (int k = 0; k < K; k++) v[k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(&V[k * ld + idx]));
#pragma unroll
        for (int k = 0; k < K; k++) { acc[k] += v[k].x + v[k].y; s += v[k].x; }
      }
    }
    if (MODE == 3) {
      hold8[(rr - rs) & 7] = s;
      if (((rr - rs) & 7) == 7 && ok) {
#pragma unroll
        for (int k = 0; k < NW; k++)
#pragma unroll
          for (int j = 0; j < 8; j++) {
            const long idx8 = (long)(rr - 7 + j) * P + c;
            if (NT) __builtin_nontemporal_store(hold8[j] + k, &Wr[k * ld + idx8]); else Wr[k * ld + idx8] = hold8[j] + k;
          }
      }
    }
    if (MODE == 1) s = (double)rr;
    if (MODE == 2) { const double cur = s; s = held; held = cur; }
    if (NW > 0 && ok && MODE != 3 && (MODE != 2 || rr > rs) && (MODE != 4 || (t >= 1 && t <= W + 2))) {
      const long idx = MODE == 4 ? ((long)strip * rows + rr) * BLK + t : (long)(MODE == 2 ? rr - 1 : rr) * P + c;
#pragma unroll
      for (int k = 0; k < NW; k++) {
        if (VEC == 1) { if (NT) __builtin_nontemporal_store(s + k, &Wr[k * ld + idx]); else Wr[k * ld + idx] = s + k; }
        else { d2 x = {s, s + k}; if (NT) __builtin_nontemporal_store(x, reinterpret_cast<d2 *>(&Wr[k * ld + idx])); else *reinterpret_cast<d2 *>(&Wr[k * ld + idx]) = x; }
      }
    }
    if (BARRIER) {
      ring[t] = s;
      __syncthreads();
      acc[0] += ring[(t + 1) % BLK];
      __syncthreads();
    }
  }
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < K; k++) s += acc[k];
  if (s == 12345.678) out[blockIdx.x] = s;
}

template <int K, int VEC, int BLK, bool BARRIER, int NW = 0, bool NT = false, int MODE = 0>
void run(const char *name, const double *V, long ld, int rows, int P, int wgs_target, double *out, double *Wr = nullptr) {
  const int Wmax = VEC == 1 ? BLK - 2 : 2 * BLK;
  const int nstrips = (P + Wmax - 1) / Wmax;
  int W = (P + nstrips - 1) / nstrips;
  if (VEC == 2) W = (W + 1) / 2 * 2;
  if (P % 16 == 0) W = (W + 15) / 16 * 16;
  int nb = wgs_target / nstrips; if (nb < 1) nb = 1;
  const int R = (rows + nb - 1) / nb;
  const int nbands = (rows + R - 1) / R;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((strip<K, VEC, BLK, BARRIER, NW, NT, MODE>), dim3(nstrips * nbands), dim3(BLK), 0, 0, V, ld, rows, P, W, nstrips, R, out, Wr);
  CHECK(hipDeviceSynchronize());
  const int reps = 5;
  CHECK(hipEventRecord(e0));
  for (int r = 0; r < reps; r++) hipLaunchKernelGGL((strip<K, VEC, BLK, BARRIER, NW, NT, MODE>), dim3(nstrips * nbands), dim3(BLK), 0, 0, V, ld, rows, P, W, nstrips, R, out, Wr);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); CHECK(hipGetLastError());
  const double bytes = 8.0 * (K + NW) * (double)rows * P;
  printf("%-64s K=%2d wgs %5d (strips %3d x bands %3d, W %3d, R %4d) %8.1f us %7.1f GB/s\n", name, K, nstrips * nbands, nstrips,
         nbands, W, R, ms * 1e3 / reps, bytes * reps / (ms * 1e-3) / 1e9);
}

int main(int argc, char **argv) {
  const int rows = 8194, P = argc > 1 ? atoi(argv[1]) : 8194;
  const long n = (long)rows * P, ld = (n + 31) / 32 * 32;
  const int KMAX = 48;
  double *V, *out;
  CHECK(hipMalloc((void **)&V, sizeof(double) * ld * KMAX)); CHECK(hipMemset(V, 0, sizeof(double) * ld * KMAX));
  CHECK(hipMalloc((void **)&out, sizeof(double) * 65536));
  double *Wr;
  CHECK(hipMalloc((void **)&Wr, sizeof(double) * ld * 3)); CHECK(hipMemset(Wr, 0, sizeof(double) * ld * 3));
  if (argc > 2 && !strcmp(argv[2], "calib")) {
    // Known byte counts for calibrating the memory-side counters on 8-byte-per-lane accesses:
    // every launch touches `cols` columns of every row in each of its streams.
    const int Wc = 249, nstrips = 33;
    long cols = 0;
    for (int s = 0; s < nstrips; s++)
      for (int t = 0; t <= Wc + 1; t++) { const int c = s * Wc - 1 + t; if (c >= 0 && c < P) cols++; }
    const double per_stream = 8.0 * (double)rows * (double)cols;
    printf("CALIB strip<16, 1, 256, true, 0, false, 0> read_bytes %.0f write_bytes 0\n", 16 * per_stream);
    printf("CALIB strip<16, 1, 256, true, 3, false, 0> read_bytes %.0f write_bytes %.0f\n", 16 * per_stream, 3 * per_stream);
    printf("CALIB strip<16, 1, 256, true, 3, true, 0> read_bytes %.0f write_bytes %.0f\n", 16 * per_stream, 3 * per_stream);
    run<16, 1, 256, true>("8 B/lane, reads only", V, ld, rows, P, 1024, out);
    run<16, 1, 256, true, 3>("8 B/lane, 3 write streams", V, ld, rows, P, 1024, out, Wr);
    run<16, 1, 256, true, 3, true>("8 B/lane, 3 NT write streams", V, ld, rows, P, 1024, out, Wr);
    return 0;
  }
  if (argc > 2 && !strcmp(argv[2], "calib_strips")) {
    // The same for strip-major vectors (aligned rows of 256 values per strip): known bytes of the access
    // pattern of the one-sweep kernels on a strip-major basis.
    const int Wc = 249, nstrips = 33;
    const long lds = (long)nstrips * rows * 256;
    CHECK(hipFree(V)); CHECK(hipFree(Wr));
    CHECK(hipMalloc((void **)&V, sizeof(double) * lds * 16)); CHECK(hipMemset(V, 0, sizeof(double) * lds * 16));
    CHECK(hipMalloc((void **)&Wr, sizeof(double) * lds * 3)); CHECK(hipMemset(Wr, 0, sizeof(double) * lds * 3));
    const double rd = 8.0 * (double)rows * nstrips * 256, wr = 8.0 * (double)rows * nstrips * (Wc + 2);
    printf("CALIB strip<16, 1, 256, true, 0, false, 4> read_bytes %.0f write_bytes 0\n", 16 * rd);
    printf("CALIB strip<16, 1, 256, true, 3, true, 4> read_bytes %.0f write_bytes %.0f\n", 16 * rd, 3 * wr);
    run<16, 1, 256, true, 0, false, 4>("8 B/lane, strip-major, reads only", V, lds, rows, P, 1024, out);
    run<16, 1, 256, true, 3, true, 4>("8 B/lane, strip-major, 3 NT write streams", V, lds, rows, P, 1024, out, Wr);
    return 0;
  }
  for (int wgs : {1024}) {
    run<16, 1, 256, true>("8 B/lane, reads only", V, ld, rows, P, wgs, out);
    run<16, 1, 256, true, 1>("8 B/lane, 1 write stream", V, ld, rows, P, wgs, out, Wr);
    run<16, 1, 256, true, 1, true>("8 B/lane, 1 NT write stream", V, ld, rows, P, wgs, out, Wr);
    run<16, 1, 256, true, 3, true>("8 B/lane, 3 NT write streams", V, ld, rows, P, wgs, out, Wr);
    run<16, 2, 256, true>("16 B/lane, reads only", V, ld, rows, P, wgs, out);
    run<16, 2, 256, true, 1>("16 B/lane, 1 write stream", V, ld, rows, P, wgs, out, Wr);
    run<16, 2, 256, true, 1, true>("16 B/lane, 1 NT write stream", V, ld, rows, P, wgs, out, Wr);
    run<16, 2, 256, true, 3, true>("16 B/lane, 3 NT write streams", V, ld, rows, P, wgs, out, Wr);
    run<40, 1, 256, true, 3, true>("8 B/lane, 3 NT write streams", V, ld, rows, P, wgs, out, Wr);
    run<40, 2, 256, true, 3, true>("16 B/lane, 3 NT write streams", V, ld, rows, P, wgs, out, Wr);
    run<16, 1, 256, true, 1, true, 1>("8 B/lane, 1 NT write stream, value independent of the loads", V, ld, rows, P, wgs, out, Wr);
    run<16, 1, 256, true, 1, true, 2>("8 B/lane, 1 NT write stream, stored one row late", V, ld, rows, P, wgs, out, Wr);
    run<16, 1, 256, true, 3, true, 1>("8 B/lane, 3 NT write streams, value independent of the loads", V, ld, rows, P, wgs, out, Wr);
    run<16, 1, 256, true, 1, true, 3>("8 B/lane, 1 NT write stream, 8 rows stored together", V, ld, rows, P, wgs, out, Wr);
    run<16, 1, 256, true, 3, true, 3>("8 B/lane, 3 NT write streams, 8 rows stored together", V, ld, rows, P, wgs, out, Wr);
    run<40, 1, 256, true, 3, true, 3>("8 B/lane, 3 NT write streams, 8 rows stored together", V, ld, rows, P, wgs, out, Wr);
    run<48, 1, 256, true>("8 B/lane, reads only", V, ld, rows, P, wgs, out);
    run<48, 2, 256, true>("16 B/lane, reads only", V, ld, rows, P, wgs, out);
  }
  return 0;
}
