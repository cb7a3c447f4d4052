// CPTP / TP / CP projection of 16 x 16 Choi matrices (two qubits; reference quantpy/tomography/process.py:231-278) with
// ONE WAVEFRONT per process and the matrix in the registers of the FP64 matrix-core tile (round 3).
//
// Layout: lane (kq = lane / 16, r16 = lane % 16) holds elements (i = kq + 4 r, j = r16), r = 0..3 -- what
// v_mfma_f64_16x16x4_f64 leaves in accumulator element r.  Two facts make the CP step's sign iteration
// (qt_signclip_wg.h explains the iteration) a register-to-register affair with no LDS traffic and no barrier:
//   * the B operand of k-step r is B[4 r + kq][r16]: accumulator element r of the same lane, so a product's result is the
//     next product's right-hand operand as it stands;
//   * the A operand of k-step r is A[r16][4 r + kq] = conj(A[4 r + kq][r16]) for a HERMITIAN A: the conjugate of the same
//     accumulator element.  Every left-hand operand of the iteration (X, and the input matrix in the final A S) is
//     Hermitian -- X is replaced by its Hermitian part every fourth step anyway (qt_signclip_wg.h: the anti-Hermitian
//     rounding grows with the lifting), through the wavefront's 4 KB of LDS, which is also where the Hermitian completion
//     of the input takes its transposed elements from.
// A step is 2 x 12 matrix instructions (three real products per complex one) in six independent chains of two, against the
// workgroup form's two barrier-separated phases on two of four wavefronts (k_cptp_project<16>, ProcWG<16>: 4.4 k clocks per
// step at four workgroups per CU); the Dykstra bookkeeping, the partial trace of the TP step (16 lanes hold the diagonal of
// each 4 x 4 block: four shuffles) and the positive-definite test (right-looking elimination, column k by shuffles) run on
// the same registers.  Same arithmetic as ProcWG<16> up to the order of sums; identical Dykstra iteration counts on
// every fixture.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_ops.h"    // v4f64
#include "qt_small.h"  // cd, gsum, wave_sync, readlane_f64

namespace qt {

struct ProcWave16 {
  static constexpr int DC = 16, DQ = 4, PT = 17;  // PT: row pitch of the transpose scratch, in complex elements
  static constexpr int kLdsComplexPerWave = DC * PT;

  struct Lane {
    int kq, r16;
    cd* T;  // this wavefront's transpose scratch
    __device__ __forceinline__ int i(int r) const { return kq + 4 * r; }
    __device__ __forceinline__ int j() const { return r16; }
  };

  __device__ __forceinline__ static cd shfl(cd v, int src) { return cd{__shfl(v.re, src), __shfl(v.im, src)}; }

  // t[r] = element (j, i) of the matrix whose element (i, j) is v[r]
  __device__ __forceinline__ static void transpose(const Lane& L, const cd (&v)[4], cd (&t)[4]) {
    wave_sync();  // (earlier reads of the scratch are done)
#pragma unroll
    for (int r = 0; r < 4; ++r) L.T[L.i(r) * PT + L.r16] = v[r];
    wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) t[r] = L.T[L.r16 * PT + L.i(r)];
  }
  __device__ __forceinline__ static void hermitian_part(const Lane& L, cd (&v)[4]) {
    cd t[4];
    transpose(L, v, t);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = cd{0.5 * (v[r].re + t[r].re), 0.5 * (v[r].im - t[r].im)};
  }

  // c = a b for a HERMITIAN a (its elements serve, conjugated, as the A operand); three real products, six chains
  __device__ __forceinline__ static void mul_h(const cd (&a)[4], const cd (&b)[4], cd (&c)[4]) {
    const v4f64 z = {0.0, 0.0, 0.0, 0.0};
    v4f64 p1[2] = {z, z}, p2[2] = {z, z}, p3[2] = {z, z};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double ar = a[r].re, ai = -a[r].im;  // conj(a[4 r + kq][r16]) = a[r16][4 r + kq]
      p1[r & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar, b[r].re, p1[r & 1], 0, 0, 0);
      p2[r & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, b[r].im, p2[r & 1], 0, 0, 0);
      p3[r & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar + ai, b[r].re + b[r].im, p3[r & 1], 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double q1 = p1[0][r] + p1[1][r], q2 = p2[0][r] + p2[1][r], q3 = p3[0][r] + p3[1][r];
      c[r] = cd{q1 - q2, q3 - q1 - q2};
    }
  }

  // TP projection (process.py:259-265): C[(a,o),(b,o)] += (delta_ab - sum_o' C[(a,o'),(b,o')]) / d.
  // Element (i, j) = ((a, o), (b, o2)) with a = r, o = kq, b = r16 / 4, o2 = r16 % 4: the diagonal of block (a, b) sits in
  // register a of the four lanes (kq = o, r16 = 4 b + o).
  __device__ __forceinline__ static void tp_project(const Lane& L, cd (&x)[4]) {
    const int b = L.r16 >> 2, o2 = L.r16 & 3;
    const bool on = L.kq == o2;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      double sr = 0.0, si = 0.0;
#pragma unroll
      for (int o = 0; o < DQ; ++o) {  // the same order in every lane: identical bits
        const cd v = shfl(x[r], 16 * o + 4 * b + o);
        sr += v.re;
        si += v.im;
      }
      if (on) {
        x[r].re += ((r == b ? 1.0 : 0.0) - sr) / DQ;
        x[r].im += (0.0 - si) / DQ;
      }
    }
  }

  // Hermitian completion from the lower triangle (LAPACK zheevd, uplo = 'L': what numpy.linalg.eigh reads)
  __device__ __forceinline__ static void complete_lower(const Lane& L, cd (&a)[4]) {
    cd t[4];
    transpose(L, a, t);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (L.i(r) < L.r16) a[r] = cd{t[r].re, -t[r].im};
      else if (L.i(r) == L.r16) a[r].im = 0.0;
    }
  }

  // Cholesky test: every pivot above eps <=> the clip at eps is the identity (to within eps).  Right-looking elimination
  // on the full Hermitian matrix: the trailing block lives in registers and, for the column / row reads of the next round,
  // in the wavefront's LDS scratch (a rolled loop: unrolled, with the columns travelling by shuffles, the kernel needed 248
  // registers).  TWO columns per round trip: column and row k + 1 are read as they are and take column k's update locally --
  // the same operations in the same order as one column at a time, so the same bits, in 8 LDS round trips instead of 16.
  __device__ __forceinline__ static bool is_pd(const Lane& L, const cd (&a)[4], double eps) {
    cd w[4];
    wave_sync();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      w[r] = a[r];
      L.T[L.i(r) * PT + L.r16] = a[r];
    }
    for (int k = 0; k < DC; k += 2) {
      wave_sync();  // the stores of the round before are in the scratch
      const double piv0 = L.T[k * PT + k].re;
      const cd akk1 = L.T[k * PT + k + 1], ak1k = L.T[(k + 1) * PT + k];  // a[k][k+1], a[k+1][k]
      const double d1 = L.T[(k + 1) * PT + k + 1].re;
      const cd r0 = L.T[k * PT + L.r16], r1 = L.T[(k + 1) * PT + L.r16];  // a[k][j], a[k+1][j]
      cd c0[4], c1[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        c0[r] = L.T[L.i(r) * PT + k];  // a[i][k]
        c1[r] = L.T[L.i(r) * PT + k + 1];
      }
      if (!(piv0 > eps)) return false;  // uniform: every lane reads the same pivots
      const double inv0 = 1.0 / piv0;
      // column k out of row k + 1, column k + 1 and the second pivot -- what one column at a time would have stored
      const double piv1 = d1 - (ak1k.re * akk1.re - ak1k.im * akk1.im) * inv0;
      if (!(piv1 > eps)) return false;
      const double inv1 = 1.0 / piv1;
      const cd r1u{r1.re - (ak1k.re * r0.re - ak1k.im * r0.im) * inv0, r1.im - (ak1k.re * r0.im + ak1k.im * r0.re) * inv0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (L.i(r) > k + 1 && L.r16 > k + 1) {
          const cd c1u{c1[r].re - (c0[r].re * akk1.re - c0[r].im * akk1.im) * inv0,
                       c1[r].im - (c0[r].re * akk1.im + c0[r].im * akk1.re) * inv0};
          w[r].re -= (c0[r].re * r0.re - c0[r].im * r0.im) * inv0;  // a_ik a_kj / a_kk  (a_kj = conj(a_jk))
          w[r].im -= (c0[r].re * r0.im + c0[r].im * r0.re) * inv0;
          w[r].re -= (c1u.re * r1u.re - c1u.im * r1u.im) * inv1;
          w[r].im -= (c1u.re * r1u.im + c1u.im * r1u.re) * inv1;
        }
      }
      wave_sync();  // every lane has read its columns and rows
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (L.i(r) > k + 1 && L.r16 > k + 1) L.T[L.i(r) * PT + L.r16] = w[r];
    }
    return true;
  }

  // U max(lambda, eps) U^dagger of the Hermitian matrix a (sign-function iteration, qt_signclip_wg.h); returns its steps
  __device__ __forceinline__ static int clip(const Lane& L, cd (&a)[4], double eps) {
    double n2 = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) n2 += a[r].re * a[r].re + a[r].im * a[r].im;
    const double nrm2 = gsum<64>(n2);
    if (!(nrm2 > 0.0)) {  // the zero matrix (or NaN input): every eigenvalue is clipped to eps
#pragma unroll
      for (int r = 0; r < 4; ++r) a[r] = cd{nrm2 == 0.0 ? (L.i(r) == L.r16 ? eps : 0.0) : nrm2, 0.0};
      return 0;
    }
    const double scale = 1.0 / sqrt(nrm2);
    cd x[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) x[r] = cd{a[r].re * scale, a[r].im * scale};
    bool lifting = true;
    int ns_left = 12, steps = 0;
    QT_STAMP(26);  // (profile build, scripts/phase_timing_cptp.py)
    for (int k = 0; k < 64; ++k) {  // every exit condition is wave-uniform (gsum returns identical bits)
      QT_STAMP_VAL(25, (long long)(k + 1));
      cd y[4];
      mul_h(x, x, y);
      double rs = 0.0;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double dr = (L.i(r) == L.r16 ? 1.0 : 0.0) - y[r].re;
        rs += dr * dr + y[r].im * y[r].im;
      }
      const double res = gsum<64>(rs);
      if (lifting && (res < 0.5 || k >= 40)) lifting = false;
      const bool last = !lifting && (res < 1e-14 || --ns_left <= 0);
      const double alpha = lifting ? 2.0 : 1.5, beta = lifting ? -1.0 : -0.5;
      cd xy[4];
      mul_h(x, y, xy);
#pragma unroll
      for (int r = 0; r < 4; ++r) x[r] = cd{fma(beta, xy[r].re, alpha * x[r].re), fma(beta, xy[r].im, alpha * x[r].im)};
      ++steps;
      if (last || !(res == res)) break;
      if ((k & 3) == 3) hermitian_part(L, x);
    }
    QT_STAMP(27);
    hermitian_part(L, x);  // S = sign(A); R = (A + A S) / 2 + eps (I - S) / 2
    cd as[4];
    mul_h(a, x, as);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      a[r] = cd{0.5 * (a[r].re + as[r].re) + 0.5 * eps * ((L.i(r) == L.r16 ? 1.0 : 0.0) - x[r].re),
                0.5 * (a[r].im + as[r].im) - 0.5 * eps * x[r].im};
    hermitian_part(L, a);
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (L.i(r) == L.r16) a[r].im = 0.0;
    return steps;
  }

  // CP projection (process.py:270-277); returns the clip's steps (0: positive definite, returned as it came)
  __device__ __forceinline__ static int cp_project(const Lane& L, cd (&a)[4], double eps) {
    complete_lower(L, a);
    const bool pd = is_pd(L, a, eps);
    QT_STAMP(2);
    if (pd) return 0;
    return clip(L, a, eps);
  }

  // Dykstra alternation (process.py:237-257); returns the iteration count
  __device__ __forceinline__ static int dykstra(const Lane& L, cd (&x)[4], int n_iter, double tol) {
    cd p[4], q[4], y[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) p[r] = q[r] = y[r] = cd{0.0, 0.0};
    int it = 0;
    for (; it < n_iter; ++it) {
      cd t[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) t[r] = cd{x[r].re + p[r].re, x[r].im + p[r].im};
      tp_project(L, t);
      double six[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const cd yd{t[r].re - y[r].re, t[r].im - y[r].im};
        y[r].re += yd.re;
        y[r].im += yd.im;
        six[0] += yd.re * q[r].re + yd.im * q[r].im;  // sum conj(y_diff) q
        six[1] += yd.re * q[r].im - yd.im * q[r].re;
        t[r] = cd{y[r].re + q[r].re, y[r].im + q[r].im};
      }
      cp_project(L, t, 1e-12);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const cd xd{t[r].re - x[r].re, t[r].im - x[r].im};
        x[r].re += xd.re;
        x[r].im += xd.im;
        const cd pd{x[r].re - y[r].re, x[r].im - y[r].im}, qd{y[r].re - x[r].re, y[r].im - x[r].im};
        six[2] += xd.re * p[r].re + xd.im * p[r].im;  // sum conj(x_diff) p
        six[3] += xd.re * p[r].im - xd.im * p[r].re;
        six[4] += pd.re * pd.re + pd.im * pd.im;
        six[5] += qd.re * qd.re + qd.im * qd.im;
        p[r].re += pd.re;
        p[r].im += pd.im;
        q[r].re += qd.re;
        q[r].im += qd.im;
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) six[u] = gsum<64>(six[u]);
      const double crit = 2.0 * (hypot(six[0], six[1]) + hypot(six[2], six[3])) + six[4] + six[5];
      if (crit < tol) {
        ++it;
        break;
      }
    }
    return it;
  }
};

// mode 0: Dykstra CPTP, 1: TP only, 2: CP only; in / out [B][16][16] complex, row-major; one wavefront per process
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) k_cptp_wave16(const double* __restrict__ in, int B, int mode, int n_iter, double tol,
                                                     double* __restrict__ out, int32_t* __restrict__ iters,
                                                     int32_t* __restrict__ status) {
  __shared__ cd scratch[4 * ProcWave16::kLdsComplexPerWave];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + w;
  if (b >= B) return;  // (wave-uniform; no workgroup barrier below)
  const ProcWave16::Lane L{lane >> 4, lane & 15, scratch + w * ProcWave16::kLdsComplexPerWave};
  const cd* src = reinterpret_cast<const cd*>(in) + (size_t)b * 256;
  cd x[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] = src[L.i(r) * 16 + L.r16];
  int it = 0;
  QT_STAMP(0);
  if (mode == 0) it = ProcWave16::dykstra(L, x, n_iter, tol);
  else if (mode == 1) ProcWave16::tp_project(L, x);
  else it = ProcWave16::cp_project(L, x, 1e-12);
  QT_STAMP(1);
  cd* dst = reinterpret_cast<cd*>(out) + (size_t)b * 256;
#pragma unroll
  for (int r = 0; r < 4; ++r) dst[L.i(r) * 16 + L.r16] = x[r];
  if (lane == 0) {
    if (iters) iters[b] = it;
    if (status) status[b] = (x[0].re == x[0].re) ? 0 : 4;
  }
}

}  // namespace qt
