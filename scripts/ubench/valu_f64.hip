// FP64 vector FMA throughput on gfx950: NCH independent v_fma_f64 chains per lane, WPS wavefronts per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
template <int NCH>
__global__ void __launch_bounds__(256) k(double* out, const double* in, int iters) {
  const double a = in[threadIdx.x & 63], b = in[64 + (threadIdx.x & 63)];
  double c[NCH];
  for (int q = 0; q < NCH; ++q) c[q] = a + q;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int q = 0; q < NCH; ++q) c[q] = fma(a, c[q], b);
  }
  double s = 0;
  for (int q = 0; q < NCH; ++q) s += c[q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NCH>
void run(int wps, double* out, const double* in) {
  #ifdef QT_UBENCH_WARM  // >= 40 ms of the same kernel first, then kernels of >= 20 ms
  const int iters = 16384 * 24,
#else
  const int iters = 16384,
#endif
  blocks = 256 * wps;
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
  CHK(hipDeviceSynchronize());
#ifdef QT_UBENCH_WARM
  for (float warm = 0.f; warm < 60.f;) {  // pre-roll: the chip needs ~40 ms of work to reach its running clocks
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float w; CHK(hipEventElapsedTime(&w, e0, e1));
    warm += w;
  }
#endif
  CHK(hipEventRecord(e0));
  hipLaunchKernelGGL(k<NCH>, dim3(blocks), dim3(256), 0, 0, out, in, iters);
  CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  const double flop = 128.0 * NCH * iters * (double)blocks * 4;
  const double instr = (double)NCH * iters * blocks * 4;
  printf("chains/lane %d  waves/SIMD %d : %8.3f ms  %7.2f TFLOP/s  %.2f ns per wave-instruction per SIMD\n", NCH, wps, ms,
         flop / (ms * 1e-3) / 1e12, ms * 1e6 / (instr / 1024));
}
int main() {
  double *out, *in;
  CHK(hipMalloc(&out, 256 * 8 * 256 * sizeof(double))); CHK(hipMalloc(&in, 128 * sizeof(double)));
  double h[128]; for (int i = 0; i < 128; ++i) h[i] = 1e-3 * (i % 7) + 0.5;
  CHK(hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice));
  run<1>(1, out, in); run<4>(1, out, in); run<8>(1, out, in);
  run<4>(2, out, in); run<4>(4, out, in); run<8>(4, out, in); run<4>(8, out, in);
  return 0;
}
