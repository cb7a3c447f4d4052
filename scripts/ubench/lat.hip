// Micro-benchmarks of single-wave instruction latencies on gfx950 (calibration for qt_small.h).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define N 2048
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

template <int MODE>
__global__ void __launch_bounds__(64) k(double* out, const double* in, unsigned long long* cyc, int* idx) {
  __shared__ double lds[1024];
  const int t = threadIdx.x;
  double a = in[t], b = in[t + 64], c0 = in[t + 128], c1 = c0 + 1, c2 = c0 + 2, c3 = c0 + 3;
  for (int i = t; i < 1024; i += 64) lds[i] = in[i % 256];
  int p = idx[t];
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {  // dependent FMA chain
#pragma unroll 16
    for (int i = 0; i < N; ++i) c0 = fma(a, c0, b);
  } else if (MODE == 1) {  // 4 independent FMA chains
#pragma unroll 4
    for (int i = 0; i < N / 4; ++i) { c0 = fma(a, c0, b); c1 = fma(a, c1, b); c2 = fma(a, c2, b); c3 = fma(a, c3, b); }
  } else if (MODE == 2) {  // dependent LDS read chain (pointer chasing)
    for (int i = 0; i < N; ++i) p = ((int*)lds)[(p & 1023)];
    c0 += p;
  } else if (MODE == 3) {  // dependent global (L2) load chain
    for (int i = 0; i < N; ++i) p = idx[p & 4095];
    c0 += p;
  } else if (MODE == 4) {  // dependent DPP
    for (int i = 0; i < N; ++i) { int v = __builtin_amdgcn_update_dpp(0, p, 0xB1, 0xf, 0xf, false); p = v + 1; }
    c0 += p;
  } else if (MODE == 5) {  // dependent bpermute
    for (int i = 0; i < N; ++i) { p = __builtin_amdgcn_ds_bpermute((t ^ 1) * 4, p) + 1; }
    c0 += p;
  } else if (MODE == 6) {  // dependent rsq
    for (int i = 0; i < N; ++i) c0 = __builtin_amdgcn_rsq(c0) + b;
  } else if (MODE == 7) {  // dependent sqrt
    for (int i = 0; i < N; ++i) c0 = sqrt(c0) + b;
  } else if (MODE == 8) {  // dependent division
    for (int i = 0; i < N; ++i) c0 = a / c0 + b;
  } else if (MODE == 9) {  // dependent log
    for (int i = 0; i < N; ++i) c0 = log(c0) + b;
  } else if (MODE == 10) {  // dependent f32 fma
    float x = (float)c0, y = (float)a, z = (float)b;
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = fmaf(y, x, z);
    c0 = x;
  } else if (MODE == 11) {  // readlane-based wave reduction pattern
    for (int i = 0; i < N / 8; ++i) {
      long long bb = __builtin_bit_cast(long long, c0);
      int lo = __builtin_amdgcn_readlane((int)bb, 0), hi = __builtin_amdgcn_readlane((int)(bb >> 32), 0);
      c0 = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo) + b;
    }
  } else if (MODE == 12) {  // independent LDS reads (throughput, conflict-free b64)
    double s = 0;
#pragma unroll 8
    for (int i = 0; i < N; ++i) s += lds[(t + i * 64) & 1023];
    c0 += s;
  } else if (MODE == 13) {  // LDS write + fence + read round trip
    for (int i = 0; i < N / 4; ++i) { lds[t] = c0; __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); c0 = lds[t ^ 1] + b; }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * 64 + t] = c0 + c1 + c2 + c3;
  if (t == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, int grid, int n_ops, double* out, double* in, unsigned long long* cyc, int* idx) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, in, cyc, idx);
  CHK(hipEventRecord(e0));
  for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(64), 0, 0, out, in, cyc, idx);
  CHK(hipEventRecord(e1));
  CHK(hipEventSynchronize(e1));
  float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long c; CHK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
  printf("%-34s grid=%5d  memtime ticks/op=%7.2f  wall/launch=%8.2f us  (ns/op=%6.2f)\n", name, grid, (double)c / n_ops, ms * 100, ms * 1e5 / n_ops);
}

int main() {
  double *out, *in; unsigned long long* cyc; int* idx;
  CHK(hipMalloc(&out, 1 << 20)); CHK(hipMalloc(&in, 1 << 16)); CHK(hipMalloc(&cyc, 1 << 16)); CHK(hipMalloc(&idx, 1 << 16));
  double h[4096]; int hi[4096];
  for (int i = 0; i < 4096; ++i) { h[i] = 1.0 + 1e-3 * (i % 97); hi[i] = (i * 37 + 11) & 4095; }
  CHK(hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice)); CHK(hipMemcpy(idx, hi, sizeof hi, hipMemcpyHostToDevice));
  for (int grid : {1, 1024}) {
    run<0>("dependent v_fma_f64", grid, N, out, in, cyc, idx);
    run<1>("4 independent v_fma_f64 chains", grid, N, out, in, cyc, idx);
    run<10>("dependent v_fma_f32", grid, N, out, in, cyc, idx);
    run<2>("dependent ds_read_b32", grid, N, out, in, cyc, idx);
    run<12>("independent ds_read_b64 + add", grid, N, out, in, cyc, idx);
    run<13>("LDS write/fence/read round trip", grid, N / 4, out, in, cyc, idx);
    run<3>("dependent global load (L2)", grid, N, out, in, cyc, idx);
    run<4>("dependent DPP mov + add", grid, N, out, in, cyc, idx);
    run<5>("dependent ds_bpermute + add", grid, N, out, in, cyc, idx);
    run<11>("readlane x2 + f64 add", grid, N / 8, out, in, cyc, idx);
    run<6>("dependent v_rsq_f64 + add", grid, N, out, in, cyc, idx);
    run<7>("dependent sqrt(f64) + add", grid, N, out, in, cyc, idx);
    run<8>("dependent f64 division + add", grid, N, out, in, cyc, idx);
    run<9>("dependent log(f64) + add", grid, N, out, in, cyc, idx);
  }
  return 0;
}
