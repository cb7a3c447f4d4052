// FP64 matrix-core throughput on gfx950: v_mfma_f64_16x16x4_f64, NACC independent accumulator chains per
// wavefront, WPS wavefronts per SIMD on every CU.  Prints TFLOP/s (2048 flop per instruction).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef double v4f64 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k(double* out, const double* in, int iters) {
  const double a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
  v4f64 acc[NACC];
  for (int q = 0; q < NACC; ++q) acc[q] = v4f64{0.0, 0.0, 0.0, 0.0};
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NACC; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
  }
  double s = 0;
  for (int q = 0; q < NACC; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
void run(int wps, double* out, const double* in) {
  #ifdef QT_UBENCH_WARM  // >= 40 ms of the same kernel first, then kernels of >= 20 ms
  const int iters = 4096 * 24,
#else
  const int iters = 4096,
#endif
  blocks = 256 * wps;  // 256 threads = 4 waves = one per SIMD; wps blocks per CU
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
  CHK(hipDeviceSynchronize());
#ifdef QT_UBENCH_WARM
  for (float warm = 0.f; warm < 60.f;) {  // pre-roll: the chip needs ~40 ms of work to reach its running clocks
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float w; CHK(hipEventElapsedTime(&w, e0, e1));
    warm += w;
  }
#endif
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 2048.0 * NACC * iters * (double)blocks * 4;
  printf("chains/wave %d  waves/SIMD %d : %8.3f ms  %7.2f TFLOP/s\n", NACC, wps, ms, flop / (ms * 1e-3) / 1e12);
}
int main() {
  double *out, *in;
  CHK(hipMalloc(&out, 256 * 8 * 256 * sizeof(double))); CHK(hipMalloc(&in, 128 * sizeof(double)));
  double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1e-3 * (i % 7) - 2e-3;
  CHK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  run<1>(1, out, in); run<2>(1, out, in); run<4>(1, out, in);
  run<1>(2, out, in); run<2>(2, out, in); run<1>(4, out, in); run<4>(4, out, in);
  return 0;
}
