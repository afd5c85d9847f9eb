// Latency of the line solver's dependent chains on one wavefront per SIMD:
//   ieee : prev = (s - prev * a) / d            (compiler's IEEE fp64 division)
//   recip: prev = corrected (s - prev * a) * y  (reciprocal + two exact residual corrections)
//   back : prev = s - prev * a
// Operands live in registers; the loop-carried value is the only dependence.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(64) void chain(double *out, const double *in, int iters) {
  const int t = blockIdx.x * 64 + threadIdx.x;
  double s[8], a[8], d[8], y[8];
  for (int k = 0; k < 8; k++) { s[k] = in[(t * 8 + k) % 1024] + 1.0; a[k] = 0.25 + 0.01 * k; d[k] = 2.0 + 0.125 * k; y[k] = 1.0 / d[k]; }
  double prev = in[t % 1024];
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      if (MODE == 0) {
        prev = (s[k] - prev * a[k]) / d[k];
      } else if (MODE == 1) {
        const double n = s[k] - prev * a[k];
        const double q0 = n * y[k];
        const double q1 = fma(fma(-d[k], q0, n), y[k], q0);
        prev = fma(fma(-d[k], q1, n), y[k], q1);
      } else {
        prev = s[k] - prev * a[k];
      }
    }
  }
  out[t] = prev;
}

int main() {
  double *in, *out; CHECK(hipMalloc((void **)&in, 8192)); CHECK(hipMalloc((void **)&out, 8 * 64 * 4096)); CHECK(hipMemset(in, 0, 8192));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 20000;  // x 8 chain steps
  const char *names[3] = {"ieee division", "reciprocal + 2 corrections", "back substitution"};
  for (int waves : {256, 1024, 2048, 4096}) {
    for (int mode = 0; mode < 3; mode++) {
      for (int rep = 0; rep < 2; rep++) {
        CHECK(hipEventRecord(e0));
        if (mode == 0) hipLaunchKernelGGL(chain<0>, dim3(waves), dim3(64), 0, 0, out, in, iters);
        if (mode == 1) hipLaunchKernelGGL(chain<1>, dim3(waves), dim3(64), 0, 0, out, in, iters);
        if (mode == 2) hipLaunchKernelGGL(chain<2>, dim3(waves), dim3(64), 0, 0, out, in, iters);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
      }
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      printf("%5d wavefronts  %-28s %8.3f ms  %7.1f ns per chain step\n", waves, names[mode], ms, ms * 1e6 / (iters * 8.0));
    }
  }
  return 0;
}
