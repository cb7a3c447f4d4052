// Workgroup eigenvalue clip  U max(lambda, eps) U^dagger  WITHOUT an eigensolver, for the d x d Hermitian matrices of
// the n = 4, 5 state kernels (d = 16, 32; one element per thread, thread t = i * d + j) -- a7, state.py:267-273.
//
// Why: the clip only needs the projector onto the negative eigenspace,
//     U max(L, eps) U^dagger = (A + A S) / 2 + eps (I - S) / 2,     S = sign(A) = U sign(L) U^dagger,
// and sign(A) is the limit of a polynomial matrix iteration that is nothing but d x d complex products -- which run on
// the FP64 matrix cores (v_mfma_f64_16x16x4_f64), while the cyclic Jacobi of qt_jacobi_wg.h is ~190 rounds of one
// barrier + one LDS burst each (370 k of the ~620 k clocks of a 5-qubit MLE trial).
//
// Iteration on X_0 = A / ||A||_F (spectrum in [-1, 1]); every step is  Y = X^2,  X <- X (alpha I + beta Y):
//   * lifting, (alpha, beta) = (2, -1): p(x) = 2 x - x^3 has slope 2 at 0, maps [0, 1.089] into itself and keeps
//     the sign, so an eigenvalue x grows by 2x per step until it sits in ~[0.88, 1.09];
//   * Newton-Schulz, (3/2, -1/2): quadratic convergence to +-1 once every |1 - x^2| < 1.
// The switch is decided by res = ||I - Y||_F^2 (a by-product of Y): res < 1/2 bounds every |1 - x_i^2| by 0.71.
// An eigenvalue that is still unlifted after 40 doublings is below 1e-12 ||A||_F; what it then contributes to the
// result is wrong by at most its own size, so the cap costs nothing measurable and there is no failure mode that
// needs a fallback.  Y is re-symmetrised when it is read (one extra LDS read), which keeps rounding from feeding
// a non-Hermitian component, and X is replaced by its Hermitian part every fourth step (round 3): X W is Hermitian only
// up to rounding, and the rows of its anti-Hermitian part that belong to a still-unlifted eigen-direction are multiplied
// by the lifting slope with everything else -- 1e-16 became 1e-4 over 40 doublings on an exactly rank-deficient matrix
// (the linear-inversion estimate of a pure state from exact frequencies), and the clip ended 1e-9 from an eigh-based one
// on such inputs while agreeing to 1e-15 on separated spectra (NumPy model of this loop: 8e-10 without, 5e-16 with the
// symmetrisation, every step or every eighth alike; qt_process64.h has the 64 x 64 measurements on the GPU).
// Agreement with LAPACK-eigh clipping: ~1e-15 (tests/test_gpu_large.py).
//
// Complex product C = A B on the matrix cores, THREE real products per tile (round 3): one wavefront per 16 x 16 tile of
// C accumulates P1 = Ar Br, P2 = Ai Bi and P3 = (Ar + Ai)(Br + Bi) over K in steps of 4 (operand layout as k_gemm,
// qt_ops.h) -- three independent MFMA chains -- and forms Cr = P1 - P2, Ci = P3 - P1 - P2 in its accumulators.  The iteration
// is bound by the throughput of the FP64 matrix pipe (one v_mfma_f64_16x16x4_f64 per ~88 clocks per SIMD):
// 96 instead of 128 instructions per 32 x 32 product, on four wavefronts (one per SIMD) instead of eight, and the lane that
// holds a result holds both of its parts (16-byte stores, the residual and the epilogue need no second wavefront).  The
// imaginary part carries a rounding error of order eps (|Ar| + |Ai|)(|Br| + |Bi|) instead of eps (|Ar||Bi| + |Ai||Br|):
// normwise the same, and the clip agrees with an eigh-based one as before (tests/test_gpu_large.py, test_gpu_fullsize.py).
// Images are complex interleaved with row pitch d + 1 (conflict-free 16-byte operand reads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_small.h"

namespace qt {

typedef double sc_v4f64 __attribute__((ext_vector_type(4)));

template <int d, int NT, bool THREE_MULT = (d >= 32)>
struct SignClipWG {
  static_assert(NT == d * d && d % 16 == 0 && NT % 64 == 0, "one thread per matrix element, 16 x 16 MFMA tiles");
  static constexpr int P = d + 1;            // row pitch of the images, in complex elements
  static constexpr int TPR = d / 16;         // tiles per row
  static constexpr int NTILE = TPR * TPR;
  static constexpr int NW = NT / 64;
  static_assert(2 * NTILE <= NW, "one wavefront per (tile, part)");
  // three images of d * P complex each (offsets in doubles from the 16-byte aligned LDS base, all even) + red [32]
  struct Lds {
    int img0, img1, img2, red;
  };

  // d = 32: three real products per tile on one wavefront (above).  d = 16 keeps four products on two wavefronts (real |
  // imaginary part): its kernels sit at the 128-register step of four workgroups per CU and the third accumulator tile pushed
  // them over it (k_mle_large_start<4>: 119 VGPRs + 24 AGPRs, three workgroups per CU), and a d = 16 step is not bound by the
  // matrix pipe's throughput in the first place (profiles/round3_phase_timing_cptp.txt).
  static constexpr bool kThreeMult = THREE_MULT;  // (the process kernels of qt_process.h ask for it at d = 16: registers to spare)
  // Which wavefronts carry the tile products: NTILE consecutive ones starting at duty0().  Where that is fewer than a
  // quarter of the workgroup (d = 16: one of four), workgroups that share a CU should not all use the same SIMD's matrix
  // pipe: wavefront w of a workgroup sits on SIMD w mod 4, and the workgroups resident on one CU are those whose indices
  // differ by multiples of the number of CUs (256 on MI355X: consecutive workgroups go to different XCDs / CUs), so the
  // starting wavefront rotates with blockIdx / 256.  A heuristic about placement: only the balance depends on it.
  static constexpr int kDuty = kThreeMult ? NTILE : 2 * NTILE;  // product wavefronts per workgroup
  __device__ __forceinline__ static int duty0() {
    if constexpr (NW >= 2 * kDuty && NW <= 4) return (int)((blockIdx.x >> 8) % (NW / kDuty)) * kDuty;
    else return 0;
  }
  // Independent accumulator chains per product tile.  A dependent v_mfma_f64_16x16x4_f64 costs ~170 clocks of latency
  // against ~16 of issue (profiles/round3_ubench_mfma_f64_warm.txt: one chain per wavefront runs the matrix pipe at 27 of
  // its 46 TFLOP/s), and since round 3 a step of the iteration IS two such chains plus two barriers: the d / 4 k-steps of
  // a tile are dealt round-robin to NACC accumulators and summed at the end.  Registers: 8 per chain.
  static constexpr int NACC = 2;
  __device__ __forceinline__ static sc_v4f64 sum_chains(const sc_v4f64 (&a)[NACC]) {
    sc_v4f64 s = a[0];
#pragma unroll
    for (int q = 1; q < NACC; ++q) s += a[q];
    return s;
  }

  // One tile's three real products over the whole K range: lane (r16, kq) feeds A[row0 + r16][k0 + kq] and
  // B[k0 + kq][col0 + r16] (B through `bload`, which may symmetrise on the fly) and ends with rows kq + 4 r of column r16.
  struct Tile3 {
    sc_v4f64 p1, p2, p3;
    __device__ __forceinline__ double re(int r) const { return p1[r] - p2[r]; }
    __device__ __forceinline__ double im(int r) const { return p3[r] - p1[r] - p2[r]; }
  };
  template <class BLoad>
  __device__ __forceinline__ static Tile3 tile_product(const cd* ap, BLoad bload) {
    Tile3 t{sc_v4f64{0.0, 0.0, 0.0, 0.0}, sc_v4f64{0.0, 0.0, 0.0, 0.0}, sc_v4f64{0.0, 0.0, 0.0, 0.0}};
#pragma unroll
    for (int k0 = 0; k0 < d; k0 += 4) {
      const cd a = ap[k0], b = bload(k0);
      t.p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.re, t.p1, 0, 0, 0);
      t.p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, b.im, t.p2, 0, 0, 0);
      t.p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re + a.im, b.re + b.im, t.p3, 0, 0, 0);
    }
    return t;
  }

  __device__ static double wsum(double* red, double v) {  // identical bits in every thread
    v = gsum<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += red[w];
    return s;
  }

  // X <- (X + X^dagger) / 2, one element per thread (e = i P + j, et = j P + i); starts behind a barrier, ends with one
  __device__ __forceinline__ static void symmetrise(cd* X, int e, int et) {
    const cd x = X[e], xt = X[et];
    __syncthreads();
    X[e] = cd{0.5 * (x.re + xt.re), 0.5 * (x.im - xt.im)};
    __syncthreads();
  }

  // C = A * B.  Ends with a barrier: C is visible to every thread, A and B may be overwritten.
  __device__ static void matmul(const cd* A, const cd* B, cd* C) {
    if constexpr (kThreeMult) {
      const int wave = (int)(threadIdx.x >> 6) - duty0(), lane = threadIdx.x & 63;
      if (wave >= 0 && wave < NTILE) {  // wave-uniform
        const int row0 = (wave / TPR) * 16, col0 = (wave % TPR) * 16;
        const int r16 = lane & 15, kq = lane >> 4;
        const cd* bp = B + kq * P + col0 + r16;
        const Tile3 t = tile_product(A + (row0 + r16) * P + kq, [&](int k0) { return bp[k0 * P]; });
  #pragma unroll
        for (int r = 0; r < 4; ++r) C[(row0 + kq + 4 * r) * P + col0 + r16] = cd{t.re(r), t.im(r)};
      }
      __syncthreads();
    } else {
      const int wave = (int)(threadIdx.x >> 6) - duty0(), lane = threadIdx.x & 63;
      if (wave >= 0 && wave < 2 * NTILE) {  // wave-uniform
        const int tile = wave >> 1, part = wave & 1;
        const int row0 = (tile / TPR) * 16, col0 = (tile % TPR) * 16;
        const int r16 = lane & 15, kq = lane >> 4;
        // (one accumulator: a second chain -- one per real product -- was measured: no change at d = 32, and the 16 x 16
        //  kernels came out 20 % slower end to end, k_mle_large_start<4> 0.100 -> 0.123 ms per 1024)
        sc_v4f64 acc = {0.0, 0.0, 0.0, 0.0};
        const cd* ap = A + (row0 + r16) * P + kq;
        const cd* bp = B + kq * P + col0 + r16;
  #pragma unroll
        for (int k0 = 0; k0 < d; k0 += 4) {
          const cd a = ap[k0], b = bp[k0 * P];
          if (part == 0) {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.re, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.im, b.im, acc, 0, 0, 0);
          } else {
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.im, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, b.re, acc, 0, 0, 0);
          }
        }
        double* cdst = reinterpret_cast<double*>(C) + part;
  #pragma unroll
        for (int r = 0; r < 4; ++r) cdst[((row0 + kq + 4 * r) * P + col0 + r16) * 2] = acc[r];
      }
      __syncthreads();
    }
  }

  // In: this thread's element of a Hermitian matrix.  Out: its element of U max(lambda, eps) U^dagger (/ trace).
  __device__ __forceinline__ static cd clip(const int t, cd a, const double eps, double* sm, const Lds o, const bool normalise) {
    const int i = t / d, j = t % d, e = i * P + j, et = j * P + i;
    const double dlt = (i == j) ? 1.0 : 0.0;
    double* red = sm + o.red;
    // (images are picked by OFFSET from the LDS base: an array of pointers indexed at run time makes the compiler
    //  lose the address space and emit flat_* for every access in the loop -- DESIGN.md section 4.5)
    int xo = o.img0, wo = o.img2;
    const int yo = o.img1;
    if (i == j) a.im = 0.0;
    const double nrm2 = wsum(red, a.re * a.re + a.im * a.im);
    if (!(nrm2 > 0.0)) {  // the zero matrix (or NaN input): every eigenvalue is clipped to eps
      const double v = normalise ? dlt / d : dlt * eps;
      return cd{nrm2 == 0.0 ? v : nrm2, 0.0};
    }
    const double scale = 1.0 / sqrt(nrm2);
    cd x{a.re * scale, a.im * scale};
    reinterpret_cast<cd*>(sm + xo)[e] = x;
    __syncthreads();
    bool lifting = true;
    int ns_left = 12;
    // Round 3: a step is TWO barrier-separated phases on the wavefronts that own the product tiles (it was five, with
    // every thread of the workgroup reading Y, reducing res and writing W in between): the residual comes out of the
    // accumulators of Y = X^2, and X (alpha I + beta Y) = alpha X + beta X Y is finished in the accumulators of the
    // second product, whose B operand is the Hermitian part of Y taken on the fly.  Measured (profiles/round3_phase_timing_*):
    // that alone moved nothing at d = 32 -- 127 k clocks per clip as before -- because the ~50 products of a clip are bound by
    // the matrix pipe's throughput, not by barriers; what moved it is fewer MFMAs per product (tile_product above).
    if constexpr (kThreeMult) {
      const int wave = (int)(threadIdx.x >> 6) - duty0(), lane = threadIdx.x & 63;
      const bool mm = wave >= 0 && wave < NTILE;  // wave-uniform: one wavefront per tile (d = 32: one per SIMD)
      const int row0 = (wave / TPR) * 16, col0 = (wave % TPR) * 16;
      const int r16 = lane & 15, kq = lane >> 4;
      QT_STAMP(26);
      for (int k = 0; k < 64; ++k) {  // every exit condition is workgroup-uniform (identical bits in every thread)
        QT_STAMP_VAL(25, (long long)(k + 1));  // (profile build: steps taken)
        const cd* X = reinterpret_cast<const cd*>(sm + xo);
        cd* Y = reinterpret_cast<cd*>(sm + yo);
        cd* Xn = reinterpret_cast<cd*>(sm + wo);
        if (mm) {  // phase 1: Y = X X, and this tile's share of res = ||I - Y||_F^2
          const cd* bp = X + kq * P + col0 + r16;
          const Tile3 t = tile_product(X + (row0 + r16) * P + kq, [&](int k0) { return bp[k0 * P]; });
          double rp = 0.0;
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + kq + 4 * r, col = col0 + r16;
            const double yr = t.re(r), yi = t.im(r);
            Y[row * P + col] = cd{yr, yi};
            const double dv = (row == col) ? 1.0 - yr : yr;
            rp = fma(dv, dv, fma(yi, yi, rp));
          }
          rp = gsum<64>(rp);
          if (lane == 0) red[wave] = rp;
        }
        __syncthreads();
        double res = 0.0;
  #pragma unroll
        for (int w = 0; w < NTILE; ++w) res += red[w];
        if (lifting && (res < 0.5 || k >= 40)) lifting = false;
        const bool last = !lifting && (res < 1e-14 || --ns_left <= 0);  // one more quadratic step squares the error
        const double alpha = lifting ? 2.0 : 1.5, beta = lifting ? -1.0 : -0.5;
        if (mm) {  // phase 2: X_next = alpha X + beta X Yh, Yh = (Y + Y^dagger) / 2 read on the fly
          const cd* bp = Y + kq * P + col0 + r16;    // Y[k][col]
          const cd* bt = Y + (col0 + r16) * P + kq;  // Y[col][k]
          const Tile3 t = tile_product(X + (row0 + r16) * P + kq, [&](int k0) {
            const cd b0 = bp[k0 * P], b1 = bt[k0];
            return cd{0.5 * (b0.re + b1.re), 0.5 * (b0.im - b1.im)};
          });
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int at = (row0 + kq + 4 * r) * P + col0 + r16;
            const cd x = X[at];
            Xn[at] = cd{fma(beta, t.re(r), alpha * x.re), fma(beta, t.im(r), alpha * x.im)};
          }
        }
        __syncthreads();
        const int tmp = xo;  // the new X sits in the third image; the old X image is the next step's target
        xo = wo;
        wo = tmp;
        if (last || !(res == res)) break;
        if ((k & 3) == 3) symmetrise(reinterpret_cast<cd*>(sm + xo), e, et);  // (uniform)
      }
    } else {
      const int wave = (int)(threadIdx.x >> 6) - duty0(), lane = threadIdx.x & 63;
      const bool mm = wave >= 0 && wave < 2 * NTILE;  // wave-uniform
      const int tile = wave >> 1, part = wave & 1;
      const int row0 = (tile / TPR) * 16, col0 = (tile % TPR) * 16;
      const int r16 = lane & 15, kq = lane >> 4;
      QT_STAMP(26);
      for (int k = 0; k < 64; ++k) {  // every exit condition is workgroup-uniform (identical bits in every thread)
        QT_STAMP_VAL(25, (long long)(k + 1));  // (profile build: steps taken)
        const cd* X = reinterpret_cast<const cd*>(sm + xo);
        cd* Y = reinterpret_cast<cd*>(sm + yo);
        cd* Xn = reinterpret_cast<cd*>(sm + wo);
        if (mm) {  // phase 1: Y = X X, and this tile's share of res = ||I - Y||_F^2
          sc_v4f64 accs[NACC];
  #pragma unroll
          for (int q = 0; q < NACC; ++q) accs[q] = sc_v4f64{0.0, 0.0, 0.0, 0.0};
          const cd* ap = X + (row0 + r16) * P + kq;
          const cd* bp = X + kq * P + col0 + r16;
  #pragma unroll
          for (int k0 = 0; k0 < d; k0 += 4) {
            const cd av = ap[k0], bv = bp[k0 * P];
            sc_v4f64& acc = accs[(k0 / 4) % NACC];
            if (part == 0) {
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.re, bv.re, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av.im, bv.im, acc, 0, 0, 0);
            } else {
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.re, bv.im, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.im, bv.re, acc, 0, 0, 0);
            }
          }
          const sc_v4f64 acc = sum_chains(accs);
          double* ydst = reinterpret_cast<double*>(Y) + part;
          double rp = 0.0;
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = row0 + kq + 4 * r, col = col0 + r16;
            ydst[(row * P + col) * 2] = acc[r];
            const double dv = (part == 0 && row == col) ? 1.0 - acc[r] : acc[r];
            rp = fma(dv, dv, rp);
          }
          rp = gsum<64>(rp);
          if (lane == 0) red[wave] = rp;
        }
        __syncthreads();
        double res = 0.0;
  #pragma unroll
        for (int w = 0; w < 2 * NTILE; ++w) res += red[w];
        if (lifting && (res < 0.5 || k >= 40)) lifting = false;
        const bool last = !lifting && (res < 1e-14 || --ns_left <= 0);  // one more quadratic step squares the error
        const double alpha = lifting ? 2.0 : 1.5, beta = lifting ? -1.0 : -0.5;
        if (mm) {  // phase 2: X_next = alpha X + beta X Yh, Yh = (Y + Y^dagger) / 2 read on the fly
          sc_v4f64 accs[NACC];
  #pragma unroll
          for (int q = 0; q < NACC; ++q) accs[q] = sc_v4f64{0.0, 0.0, 0.0, 0.0};
          const cd* ap = X + (row0 + r16) * P + kq;
          const cd* bp = Y + kq * P + col0 + r16;       // Y[k][col]
          const cd* bt = Y + (col0 + r16) * P + kq;     // Y[col][k]
  #pragma unroll
          for (int k0 = 0; k0 < d; k0 += 4) {
            const cd av = ap[k0], b0 = bp[k0 * P], b1 = bt[k0];
            const double br = 0.5 * (b0.re + b1.re), bi = 0.5 * (b0.im - b1.im);
            sc_v4f64& acc = accs[(k0 / 4) % NACC];
            if (part == 0) {
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.re, br, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-av.im, bi, acc, 0, 0, 0);
            } else {
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.re, bi, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av.im, br, acc, 0, 0, 0);
            }
          }
          const sc_v4f64 acc = sum_chains(accs);
          const double* xsrc = reinterpret_cast<const double*>(X) + part;
          double* xdst = reinterpret_cast<double*>(Xn) + part;
  #pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int at = ((row0 + kq + 4 * r) * P + col0 + r16) * 2;
            xdst[at] = fma(beta, acc[r], alpha * xsrc[at]);
          }
        }
        __syncthreads();
        const int tmp = xo;  // the new X sits in the third image; the old X image is the next step's target
        xo = wo;
        wo = tmp;
        if (last || !(res == res)) break;
        if ((k & 3) == 3) symmetrise(reinterpret_cast<cd*>(sm + xo), e, et);  // (uniform)
      }
    }
    QT_STAMP(27);
    // S = sign(A) sits in the image at xo.  R = (A + A S) / 2 + eps (I - S) / 2
    cd* S = reinterpret_cast<cd*>(sm + xo);
    cd* Y = reinterpret_cast<cd*>(sm + yo);
    cd* W = reinterpret_cast<cd*>(sm + wo);
    W[e] = a;
    __syncthreads();
    matmul(W, S, Y);
    const cd as = Y[e], s = S[e];
    cd r{0.5 * (a.re + as.re) + 0.5 * eps * (dlt - s.re), 0.5 * (a.im + as.im) - 0.5 * eps * s.im};
    W[e] = r;  // (all reads of W by the product are behind the barrier that ended it)
    __syncthreads();
    const cd rt = W[et];
    r = cd{0.5 * (r.re + rt.re), 0.5 * (r.im - rt.im)};
    if (i == j) r.im = 0.0;
    if (!normalise) {
      __syncthreads();
      return r;
    }
    const double tr = wsum(red, i == j ? r.re : 0.0);  // (barriers inside: W is free again)
    return cd{r.re / tr, r.im / tr};
  }
};

}  // namespace qt
