// Process tomography of THREE qubits (reference quantpy/tomography/process.py:142-289, size-generic there):
// 64 input states x 216 POVM rows, a 64 x 64 Choi matrix.
//
// Linear inversion without the 13824 x 4096 design matrix.  The reference builds rows
//     vec(rho_s (x) E_m^T)            (process.py:203-208; 906 MB complex at n = 3)
// and applies inv(L^T L) L^T to the frequencies (routines.py:69-71, plain transposes).  A row is the Kronecker
// product of vec(rho_s) and vec(E_m^T) up to a fixed permutation of the columns, L = (V_S (x) V_P) Pi^T, and the
// left inverse of a Kronecker product is the Kronecker product of the left inverses:
//     L^+ = Pi (V_S^+ (x) V_P^+),      V_S = [vec rho_s]  64 x 64,   V_P = [vec E_m]  216 x 64   (both complex).
// So   X = V_S^+ . F . V_P^+^T   with F[s][m] the frequencies, and  Choi[(a, b)][(c, e)] = X[(a, c)][(e, b)]:
// two small products per process (1.1 M multiply-adds instead of 56 M) and 0.3 MB of operands instead of 1.8 GB.
// qt_process_setup factors the design matrix this way for n = 3; k_lifp_kron_finish is the second product + the
// index shuffle (the first one, over the whole batch, is k_gemm).
//
// CPTP projection (process.py:231-278) for the 64 x 64 Choi matrix: one 512-thread workgroup per process, EIGHT
// elements per thread -- those a wavefront's lane receives from v_mfma_f64_16x16x4_f64 for its two 16 x 16 tiles
// (wavefront w <-> tiles 2w, 2w + 1: rows 16 (w / 2) + kq + 4 r, columns 32 (w % 2) + 16 t + r16) -- so the products
// of the CP step's sign iteration (qt_signclip_wg.h explains the iteration) leave their result where the element-wise
// updates of the Dykstra loop want it.  LDS: two 64 x 65 complex images (the third image of SignClipWG is the
// accumulator registers here) + a column buffer for the Cholesky test; Dykstra's p, q, y wait in global memory.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_process.h"
#include "qt_signclip_wg.h"

namespace qt {

struct Proc64 {
  // 512 threads: a wavefront owns TWO neighbouring 16 x 16 tiles of the matrix (tiles 2w and 2w + 1 of the 4 x 4 grid:
  // same rows, adjacent columns), eight elements per thread.  With 1024 threads x 4 elements the 128-register cap of
  // a 1024-thread workgroup left the Dykstra body + the sign iteration 75-98 registers short (280-370 B of scratch);
  // eight waves have 256 registers each, the two tiles share every A-operand read, and a wavefront carries four
  // independent MFMA chains.  The matrix pipe sees the same 256 MFMAs per SIMD and product either way.
  static constexpr int DC = 64, DQ = 8, NT = 512, NW = NT / 64, EPT = 8, P = DC + 1;
  // LDS layout, offsets in doubles (all even: cd accesses are 16-byte aligned)
  static constexpr int oImg0 = 0, oImg1 = oImg0 + 2 * DC * P, oCol = oImg1 + 2 * DC * P, oTp = oCol + 2 * 2 * DC,
                       oRed = oTp + 2 * DC, oRed6 = oRed + 32, oPart = oRed6 + 6 * NW + ((6 * NW) & 1),
                       kDoubles = oPart + 2 * 4 * 64 * 2;  // oPart: the halves of the two split tiles (below), [tile][r][lane] complex
  static constexpr size_t kLdsBytes = (size_t)kDoubles * sizeof(double);
  static constexpr int kWsComplex = 5 * EPT * NT;  // global workspace per process: p, q, y, x, the clip's parked input

  struct Map {  // element r of this thread: row i(r), column j(r); r = 4 * tile + accumulator slot
    int row0, col0, r16, kq, tid;
    __device__ explicit Map(int tid_) : tid(tid_) {
      const int w = tid >> 6, lane = tid & 63;
      row0 = 16 * (w >> 1), col0 = 32 * (w & 1), r16 = lane & 15, kq = lane >> 4;
    }
    __device__ __forceinline__ int i(int r) const { return row0 + kq + 4 * (r & 3); }
    __device__ __forceinline__ int j(int r) const { return col0 + 16 * (r >> 2) + r16; }
  };

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

  // Every product of the CP step is Hermitian (Y = X X, Z = Y Y, X W, the final A S), so only the 10 tiles on and above the
  // diagonal of the 4 x 4 tile grid are computed and the 6 below are written as their mirror images (clip() below has why
  // the new X may be mirrored too, and what was wrong with round 2's mirror).  Which wavefront computes what is decoupled
  // from which thread owns which element.  Ten tiles on four matrix pipes: every wavefront takes ONE tile over the whole
  // K range, and the two tiles left over -- (0, 2) and (1, 3) -- are split in K between wavefronts 0 | 1 and 2 | 3, one half per
  // SIMD (wavefronts w and w + 4 share a SIMD): 2.5 tiles per SIMD and product where rounds 2-3a had 3 + 3 + 2 + 2.  The first
  // halves travel through 8 KB of LDS (`oPart`) to wavefronts 1 and 3, which add them and store the tile; the barrier that
  // publishes them is the one that ends the operand reads anyway (product() below).
  struct Work {
    int row0, col0, r16, kq, wave;
    int hrow0, hcol0, kh0;  // the split tile and this wavefront's half of its K range (wavefronts 0..3)
    bool half, owner;
    __device__ explicit Work(int tid) {
      wave = tid >> 6;
      const int lane = tid & 63;
      row0 = 16 * ((0x3A50 >> (2 * wave)) & 3);  // tile rows    0 0 1 1 2 2 3 0
      col0 = 16 * ((0xFE94 >> (2 * wave)) & 3);  // tile columns 0 1 1 2 2 3 3 3
      half = wave < 4;
      owner = half && (wave & 1);  // (wavefronts 1 and 3: their own tiles are off the diagonal -- no symmetrisation round trip)
      hrow0 = 16 * (wave >> 1), hcol0 = 16 * (2 + (wave >> 1));  // (0, 2) for wavefronts 0 | 1, (1, 3) for 2 | 3
      kh0 = 32 * (wave & 1);
      r16 = lane & 15, kq = lane >> 4;
    }
  };

  // This wavefront's share of A B into registers: c[0..3] its tile, c[4..7] its half of a split tile (wavefronts 0..3); no
  // barrier inside.  THREE real products per complex tile (round 3, as SignClipWG::tile_product at d = 32): P1 = Ar Br,
  // P2 = Ai Bi, P3 = (Ar + Ai)(Br + Bi), C = (P1 - P2) + i (P3 - P1 - P2) -- the step is bound by the throughput of the FP64
  // matrix pipe, and the three (six) accumulator chains are independent.
  __device__ __forceinline__ static void tiles(const cd* A, const cd* B, const Work& wk, cd (&c)[EPT]) {
    const sc_v4f64 z = {0.0, 0.0, 0.0, 0.0};
    sc_v4f64 p1a = z, p2a = z, p3a = z, p1b = z, p2b = z, p3b = z;
    const cd* ap = A + (wk.row0 + wk.r16) * P + wk.kq;
    const cd* bp = B + wk.kq * P + wk.col0 + wk.r16;
    if (wk.half) {  // wave-uniform
      const cd* hp = A + (wk.hrow0 + wk.r16) * P + wk.kq + wk.kh0;
      const cd* gp = B + (wk.kq + wk.kh0) * P + wk.hcol0 + wk.r16;
#pragma unroll 4
      for (int k0 = 0; k0 < DC / 2; k0 += 4) {  // the split tile's half beside the first half of the own tile: six chains
        const cd a = ap[k0], b = bp[k0 * P], ha = hp[k0], hb = gp[k0 * P];
        p1a = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.re, p1a, 0, 0, 0);
        p1b = __builtin_amdgcn_mfma_f64_16x16x4f64(ha.re, hb.re, p1b, 0, 0, 0);
        p2a = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, b.im, p2a, 0, 0, 0);
        p2b = __builtin_amdgcn_mfma_f64_16x16x4f64(ha.im, hb.im, p2b, 0, 0, 0);
        p3a = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re + a.im, b.re + b.im, p3a, 0, 0, 0);
        p3b = __builtin_amdgcn_mfma_f64_16x16x4f64(ha.re + ha.im, hb.re + hb.im, p3b, 0, 0, 0);
      }
#pragma unroll 4
      for (int k0 = DC / 2; k0 < DC; k0 += 4) {
        const cd a = ap[k0], b = bp[k0 * P];
        p1a = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.re, p1a, 0, 0, 0);
        p2a = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, b.im, p2a, 0, 0, 0);
        p3a = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re + a.im, b.re + b.im, p3a, 0, 0, 0);
      }
    } else {
#pragma unroll 4
      for (int k0 = 0; k0 < DC; k0 += 4) {
        const cd a = ap[k0], b = bp[k0 * P];
        p1a = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.re, p1a, 0, 0, 0);
        p2a = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, b.im, p2a, 0, 0, 0);
        p3a = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re + a.im, b.re + b.im, p3a, 0, 0, 0);
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      c[r] = cd{p1a[r] - p2a[r], p3a[r] - p1a[r] - p2a[r]};
      c[4 + r] = cd{p1b[r] - p2b[r], p3b[r] - p1b[r] - p2b[r]};
    }
  }

  // One tile into image D together with its mirror image; a tile ON the diagonal is replaced by its Hermitian part --
  // inside the wavefront that computed it: raw store, transposed read, second store; LDS serves one wavefront's accesses in
  // program order -- so the image is exactly Hermitian when the barrier after it opens, and c[] holds what was stored.
  // Returns the tile's share of sum |delta_ij - D_ij|^2 over the WHOLE matrix (a mirrored element counts twice).
  __device__ __forceinline__ static double store_hermitian_tile(cd* D, int row0, int col0, int r16, int kq, cd* c) {
    const bool diag = col0 == row0;  // wave-uniform
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = row0 + kq + 4 * r, j = col0 + r16;
      D[i * P + j] = c[r];
      if (!diag) D[j * P + i] = cd{c[r].re, -c[r].im};
    }
    if (diag) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const cd ct = D[(col0 + r16) * P + row0 + kq + 4 * r];
        c[r] = cd{0.5 * (c[r].re + ct.re), 0.5 * (c[r].im - ct.im)};
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) D[(row0 + kq + 4 * r) * P + col0 + r16] = c[r];
    }
    double rp = 0.0;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double dr = ((row0 + kq + 4 * r == col0 + r16) ? 1.0 : 0.0) - c[r].re, di = c[r].im;
      rp += (diag ? 1.0 : 2.0) * (dr * dr + di * di);
    }
    return rp;
  }

  // D <- fin(A B) for a product that is Hermitian up to rounding; D may be A or B.  `fin(c, row0, col0)` turns the raw tile
  // c[0..3] at (row0, col0) into what is stored (it may read the OLD contents of any image at the tile's own positions).
  // Sequence: every wavefront computes its share; wavefronts 0, 2 leave their halves of the split tiles in `part`;
  // BARRIER (the halves are visible, and every operand read of A and B is done -- so D may alias them); stores.  The
  // caller's next barrier publishes D.  Returns this wavefront's share of ||I - D||_F^2.
  template <class Fin>
  __device__ __forceinline__ static double product(const cd* A, const cd* B, cd* D, const Work& wk, double* sm, Fin fin) {
    cd c[EPT];
    tiles(A, B, wk, c);
    cd* part = reinterpret_cast<cd*>(sm + oPart) + (wk.wave >> 1) * 4 * 64 + (wk.kq * 16 + wk.r16);
    if (wk.half && !wk.owner) {
#pragma unroll
      for (int r = 0; r < 4; ++r) part[r * 64] = c[4 + r];
    }
    __syncthreads();
    fin(c, wk.row0, wk.col0);
    double rs = store_hermitian_tile(D, wk.row0, wk.col0, wk.r16, wk.kq, c);
    if (wk.owner) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const cd o = part[r * 64];
        c[4 + r] = cd{c[4 + r].re + o.re, c[4 + r].im + o.im};
      }
      fin(c + 4, wk.hrow0, wk.hcol0);
      rs += store_hermitian_tile(D, wk.hrow0, wk.hcol0, wk.r16, wk.kq, c + 4);
    }
    return rs;
  }

  // Hermitian completion from the lower triangle (LAPACK zheevd, uplo = 'L': what numpy.linalg.eigh reads)
  __device__ static void complete_lower(cd (&a)[EPT], const Map& m, double* sm) {
    cd* X = reinterpret_cast<cd*>(sm + oImg0);
#pragma unroll
    for (int r = 0; r < EPT; ++r) X[m.i(r) * P + m.j(r)] = a[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int i = m.i(r), j = m.j(r);
      if (i < j) {
        const cd t = X[j * P + i];
        a[r] = cd{t.re, -t.im};
      } else if (i == j) {
        a[r].im = 0.0;
      }
    }
    __syncthreads();
  }

  // Does the (Hermitian) matrix have a Cholesky factorisation with every pivot above eps?  Then no eigenvalue is
  // clipped by more than eps and the CP projection is the identity (the common case once Dykstra's iterates
  // settle).  Right-looking elimination on the register-resident elements; only column k travels through LDS.
  __device__ static bool is_pd(const cd (&a)[EPT], double eps, const Map& m, double* sm) {
    cd w[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) w[r] = a[r];
    bool pd = true;
    for (int k = 0; k < DC; ++k) {
      cd* col = reinterpret_cast<cd*>(sm + oCol) + (k & 1) * DC;
#pragma unroll
      for (int r = 0; r < EPT; ++r)
        if (m.j(r) == k) col[m.i(r)] = w[r];
      __syncthreads();
      const double piv = col[k].re;
      if (!(piv > eps)) {  // uniform: every thread reads the same pivot
        pd = false;
        break;
      }
      const double inv = 1.0 / piv;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (m.j(4 * t) > k) {
          const cd cj = col[m.j(4 * t)];
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * t + rr;
            if (m.i(r) > k) {
              const cd ci = col[m.i(r)];
              w[r].re -= (ci.re * cj.re + ci.im * cj.im) * inv;  // a_ik conj(a_jk) / a_kk
              w[r].im -= (ci.im * cj.re - ci.re * cj.im) * inv;
            }
          }
        }
      }
    }
    __syncthreads();  // the column buffers are free again
    return pd;
  }

  // U max(lambda, eps) U^dagger of the Hermitian matrix whose elements a[] this thread holds (sign-function
  // iteration of qt_signclip_wg.h; X in image 0, Y / W / A in image 1, products land in registers).
  // `park` = EPT x NT complex of global memory for this process (element r of thread t at [r * NT + t]): the input
  // waits there while the iteration runs.
  __device__ static int clip(cd (&a)[EPT], double eps, const Map& m, double* sm, cd* park) {  // returns its product pairs
    cd* X = reinterpret_cast<cd*>(sm + oImg0);
    cd* Y = reinterpret_cast<cd*>(sm + oImg1);
    double* red = sm + oRed;
    int e[EPT], et[EPT];
    double n2 = 0.0;
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      e[r] = m.i(r) * P + m.j(r);
      et[r] = m.j(r) * P + m.i(r);
      if (e[r] == et[r]) a[r].im = 0.0;
      n2 += a[r].re * a[r].re + a[r].im * a[r].im;
    }
    const double nrm2 = wsum(red, n2);
    if (!(nrm2 > 0.0)) {  // the zero matrix (or NaN input): every eigenvalue is clipped to eps
#pragma unroll
      for (int r = 0; r < EPT; ++r) a[r] = cd{nrm2 == 0.0 ? (e[r] == et[r] ? eps : 0.0) : nrm2, 0.0};
      return 0;
    }
    const double scale = 1.0 / sqrt(nrm2);
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      park[r * NT + threadIdx.x] = a[r];
      X[e[r]] = cd{a[r].re * scale, a[r].im * scale};
    }
    __syncthreads();
    bool lifting = true;
    int ns_left = 12, steps = 0;
    const Work wk(m.tid);  // (derived from the caller's re-derived thread index: see dykstra)
    const int wave = m.tid >> 6;
    // A step (round 3: EVERY product Hermitian, three barriers where there were six, a lifting polynomial of degree 5).
    //   phase 1  Y = X X on the upper tiles, mirrored (store_hermitian); res = ||I - Y||_F^2 out of the accumulators.
    //   lifting  x <- x (3 - 3.25 x^2 + 1.25 x^4):  Z = Y Y, W' = -3.25 Y + 1.25 Z written over Y by the wavefronts that own
    //            the tiles, X <- 3 X + X W'.
    //   Newton-Schulz  X <- 1.5 X - 0.5 X Y.
    //   Both finish in the accumulators of the product's UPPER tiles and are stored like Y: mirrored, diagonal tiles
    //   replaced by their Hermitian part.
    // Why the new X may be mirrored after all (round 2 computed it in full: mirroring had cost four digits): the product X W
    // is Hermitian only up to rounding, and its anti-Hermitian part N is not harmless noise -- the rows of N that belong to a
    // still-unlifted eigen-direction are multiplied by the lifting slope with everything else, step after step (1e-16 becomes
    // 1e-8 over the 22 steps that lift an eigenvalue of 1e-10 ||A||).  With X computed in full N stayed in X, and the mirrored
    // Y turned it into a HERMITIAN perturbation of the iteration: 1e-9 on a geometric spectrum down to 1e-10 ||A||, 6e-6 on
    // a matrix of exact rank 32 (GPU and the NumPy model, scripts/sign_schedule_model.py).  Round 2's mirrored X kept the
    // diagonal tiles as computed, i.e. kept N inside them (2e-10 on the same spectrum).  Taking the Hermitian part of
    // EVERY iterate -- upper tiles mirrored, diagonal tiles symmetrised in the wavefront that computed them -- removes N
    // where it arises: 1e-16 on all of these, and a step is 3 + 3 (+ 3) tile products per SIMD instead of 3 + 4 (+ 3).
    // The quintic: the odd one with p(1) = 1, p'(1) = -1/2 (lifted eigenvalues settle on 1 geometrically, so res ends up
    // measuring the stragglers only) and slope 3 at 0; slope 3.2 is where the invariant interval [0, 1.26] is lost.  It
    // grows an unlifted eigenvalue by 3.0 for 9 tile products where 1.9 x - 0.9 x^3 (round 2) took 6 for 1.9.
    auto keep = [](cd*, int, int) {};
    for (int k = 0; k < 64; ++k) {  // every exit condition is workgroup-uniform (identical bits in every thread)
      double rs = product(X, X, Y, wk, sm, keep);  // (image 1 is idle: the last product ended behind a barrier)
      rs = gsum<64>(rs);
      if ((m.tid & 63) == 0) red[wave] = rs;
      __syncthreads();
      double res = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) res += red[w];
      if (lifting && (res < 0.5 || k >= 24)) lifting = false;  // (3^24 = 2.8e11: what is still unlifted then is below 1e-11 ||A||)
      const bool last = !lifting && (res < 1e-14 || --ns_left <= 0);
      double alpha = 1.5, beta = -0.5;
      if (lifting) {  // uniform.  W' = -3.25 Y + 1.25 Y Y over Y (its own tile re-read: cheaper than registers kept alive)
        (void)product(Y, Y, Y, wk, sm, [&](cd* c, int row0, int col0) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const cd y = Y[(row0 + wk.kq + 4 * r) * P + col0 + wk.r16];
            c[r] = cd{fma(1.25, c[r].re, -3.25 * y.re), fma(1.25, c[r].im, -3.25 * y.im)};
          }
        });
        __syncthreads();
        alpha = 3.0, beta = 1.0;
      }
      (void)product(X, Y, X, wk, sm, [&](cd* c, int row0, int col0) {  // X <- alpha X + beta X Y (or X W')
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const cd x = X[(row0 + wk.kq + 4 * r) * P + col0 + wk.r16];
          c[r] = cd{fma(beta, c[r].re, alpha * x.re), fma(beta, c[r].im, alpha * x.im)};
        }
      });
      __syncthreads();
      ++steps;
      if (last || !(res == res)) break;
    }
    // S = sign(A) sits in X.  R = (A + A S) / 2 + eps (I - S) / 2
    cd s[EPT];
    asm volatile("" : "+v"(park) : : "memory");  // a real reload: without it the compiler forwards the stored registers
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      a[r] = park[r * NT + threadIdx.x];
      s[r] = X[e[r]];
      Y[e[r]] = a[r];
    }
    __syncthreads();
    (void)product(Y, X, Y, wk, sm, keep);  // A S over A, Hermitian as well (A and its sign commute)
    __syncthreads();
    cd out[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const cd as = Y[e[r]];
      out[r] = cd{0.5 * (a[r].re + as.re) + 0.5 * eps * ((e[r] == et[r] ? 1.0 : 0.0) - s[r].re),
                  0.5 * (a[r].im + as.im) - 0.5 * eps * s[r].im};
    }
    __syncthreads();  // everybody has its elements of A S
#pragma unroll
    for (int r = 0; r < EPT; ++r) Y[e[r]] = out[r];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const cd rt = Y[et[r]];
      a[r] = cd{0.5 * (out[r].re + rt.re), 0.5 * (out[r].im - rt.im)};
      if (e[r] == et[r]) a[r].im = 0.0;
    }
    __syncthreads();
    return steps;
  }

  // CP projection (process.py:270-277): eigh on the lower triangle, clip at eps, rebuild.  Returns the number of
  // sign-iteration steps it took (0: the matrix was positive definite and is returned as it came).
  __device__ static int cp_project(cd (&a)[EPT], double eps, const Map& m, double* sm, cd* park) {
    complete_lower(a, m, sm);
    if (is_pd(a, eps, m, sm)) return 0;
    return clip(a, eps, m, sm, park);
  }

  // TP projection (process.py:259-265): C[(a,o),(b,o)] += (delta_ab - sum_o' C[(a,o'),(b,o')]) / d
  __device__ static void tp_project(cd (&a)[EPT], const Map& m, double* sm) {
    cd* X = reinterpret_cast<cd*>(sm + oImg0);
    cd* rr = reinterpret_cast<cd*>(sm + oTp);
#pragma unroll
    for (int r = 0; r < EPT; ++r) X[m.i(r) * P + m.j(r)] = a[r];
    __syncthreads();
    if (threadIdx.x < DQ * DQ) {
      const int ia = threadIdx.x / DQ, ib = threadIdx.x % DQ;
      double sr = 0.0, si = 0.0;
      for (int o = 0; o < DQ; ++o) {
        const cd t = X[(ia * DQ + o) * P + (ib * DQ + o)];
        sr += t.re;
        si += t.im;
      }
      rr[threadIdx.x] = cd{sr, si};
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
      const int i = m.i(r), j = m.j(r), ia = i / DQ, o = i % DQ, ib = j / DQ, o2 = j % DQ;
      if (o == o2) {
        const cd t = rr[ia * DQ + ib];
        a[r].re += ((ia == ib ? 1.0 : 0.0) - t.re) / DQ;
        a[r].im += (0.0 - t.im) / DQ;
      }
    }
    __syncthreads();
  }

  // Dykstra alternation (process.py:237-257); x = this thread's elements; returns the iteration count.
  // p, q, y -- and x across the CP step -- live in `ws` (kWsComplex complex numbers of global memory per process,
  // element r of thread t at [r * NT + t]: coalesced, L2-resident, touched a few times per iteration), so the CP
  // step's product tiles, iterate and Hermitian part are all that is live while it runs.
  __device__ static int dykstra(cd (&x)[EPT], int n_iter, double tol, const Map&, double* sm, cd* ws) {
    cd* base = ws + threadIdx.x;  // one live address; the blocks are constant offsets from it
    constexpr int pw = 0, qw = EPT * NT, yw = 2 * EPT * NT, xw = 3 * EPT * NT, parkw = 4 * EPT * NT;
    int it = 0;
    double* const sm0 = sm;
    cd* const ws0 = ws;
    for (; it < n_iter; ++it) {
      // Addresses are re-derived inside every iteration: left alone, the compiler hoists the ~40 global and ~50 LDS
      // addresses of the body out of this loop and keeps them alive through the CP step (114 registers spilled).
      int fresh = 0;
      asm volatile("" : "+v"(fresh));
      sm = sm0 + fresh;
      ws = ws0 + fresh;
      base = ws + threadIdx.x;
      const Map m(threadIdx.x + fresh);  // (shadows the argument: the element indices are re-derived too)
      const bool first = it == 0;  // p = q = y = 0 without reading the workspace
      cd t[EPT];
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        const cd p = first ? cd{0.0, 0.0} : base[pw + r * NT];
        t[r] = cd{x[r].re + p.re, x[r].im + p.im};
      }
      tp_project(t, m, sm);
      double s0 = 0.0, s1 = 0.0;
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        cd y = first ? cd{0.0, 0.0} : base[yw + r * NT];
        const cd q = first ? cd{0.0, 0.0} : base[qw + r * NT];
        const cd yd{t[r].re - y.re, t[r].im - y.im};
        y.re += yd.re;
        y.im += yd.im;
        base[yw + r * NT] = y;
        s0 += yd.re * q.re + yd.im * q.im;  // sum conj(y_diff) q
        s1 += yd.re * q.im - yd.im * q.re;
        t[r] = cd{y.re + q.re, y.im + q.im};
        base[xw + r * NT] = x[r];
      }
      cp_project(t, 1e-12, m, sm, ws + parkw);
      double six[6] = {s0, s1, 0.0, 0.0, 0.0, 0.0};
      asm volatile("" : "+v"(base) : : "memory");  // x is really reloaded (see clip)
#pragma unroll
      for (int r = 0; r < EPT; ++r) {
        x[r] = base[xw + r * NT];
        cd p = first ? cd{0.0, 0.0} : base[pw + r * NT];
        cd q = first ? cd{0.0, 0.0} : base[qw + r * NT];
        const cd y = base[yw + r * NT];
        const cd xd{t[r].re - x[r].re, t[r].im - x[r].im};
        x[r].re += xd.re;
        x[r].im += xd.im;
        const cd pd{x[r].re - y.re, x[r].im - y.im}, qd{y.re - x[r].re, y.im - x[r].im};
        six[2] += xd.re * p.re + xd.im * p.im;  // sum conj(x_diff) p
        six[3] += xd.re * p.im - xd.im * p.re;
        six[4] += pd.re * pd.re + pd.im * pd.im;
        six[5] += qd.re * qd.re + qd.im * qd.im;
        p.re += pd.re;
        p.im += pd.im;
        q.re += qd.re;
        q.im += qd.im;
        base[pw + r * NT] = p;
        base[qw + r * NT] = q;
      }
      block_sums<NT, 6>(six, sm + oRed6);
      const double crit = 2.0 * (hypot(six[0], six[1]) + hypot(six[2], six[3])) + six[4] + six[5];
      __syncthreads();  // the reduction scratch is read; the next round may write it
      if (crit < tol) {
        ++it;
        break;
      }
    }
    return it;
  }
};

// mode 0: Dykstra CPTP, 1: TP only, 2: CP only  (process.py:231-278); in / out [B][64][64] complex, row-major;
// ws: Proc64::kWsComplex complex numbers per process (modes 0 and 2)
__global__ void __launch_bounds__(Proc64::NT) k_cptp_project64(const double* __restrict__ in, int B, int mode, int n_iter,
                                                                double tol, double* __restrict__ out,
                                                                int32_t* __restrict__ iters, int32_t* __restrict__ status,
                                                                double* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) double sm64[];
  const int b = blockIdx.x;
  if (b >= B) return;
  const Proc64::Map m(threadIdx.x);
  const cd* src = reinterpret_cast<const cd*>(in) + (size_t)b * Proc64::DC * Proc64::DC;
  cd* wsb = reinterpret_cast<cd*>(ws) + (size_t)b * Proc64::kWsComplex;
  cd x[Proc64::EPT];
#pragma unroll
  for (int r = 0; r < Proc64::EPT; ++r) x[r] = src[m.i(r) * Proc64::DC + m.j(r)];
  int it = 0;
  if (mode == 0) it = Proc64::dykstra(x, n_iter, tol, m, sm64, wsb);
  else if (mode == 1) Proc64::tp_project(x, m, sm64);
  else it = Proc64::cp_project(x, 1e-12, m, sm64, wsb + 4 * Proc64::EPT * Proc64::NT);  // (mode 2 reports the clip's steps)
  cd* dst = reinterpret_cast<cd*>(out) + (size_t)b * Proc64::DC * Proc64::DC;
#pragma unroll
  for (int r = 0; r < Proc64::EPT; ++r) dst[m.i(r) * Proc64::DC + m.j(r)] = x[r];
  if (threadIdx.x == 0) {
    if (iters) iters[b] = it;
    if (status) status[b] = (x[0].re == x[0].re) ? 0 : 4;
  }
}

// Second half of the factored linear inversion: X = V_S^+ . T_b (64 x 64 complex each), then
// Choi[(a d + b)][(c d + e)] = X[(a d + c)][(e d + b)].  T = F . V_P^+^T comes from k_gemm over the whole batch:
// T[b][s][beta] complex, beta = e d + b the index of E_m[e][b] in `emats`.  One 256-thread workgroup per process;
// T_b is staged in LDS, a thread owns one row alpha of X and 16 of its columns.
__global__ void __launch_bounds__(256) k_lifp_kron_finish(const double* __restrict__ T, const double* __restrict__ vs_pinv,
                                                          int B, double* __restrict__ choi, int32_t* __restrict__ status,
                                                          int32_t* __restrict__ iters) {
  constexpr int DC = 64, d = 8;
  __shared__ cd tb[DC * DC];
  const int b = blockIdx.x;
  if (b >= B) return;
  const cd* t = reinterpret_cast<const cd*>(T) + (size_t)b * DC * DC;
  for (int k = threadIdx.x; k < DC * DC; k += 256) tb[k] = t[k];
  __syncthreads();
  const int alpha = threadIdx.x >> 2, b0 = (threadIdx.x & 3) * 16;
  const cd* vrow = reinterpret_cast<const cd*>(vs_pinv) + (size_t)alpha * DC;
  cd acc[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) acc[k] = cd{0.0, 0.0};
  for (int s0 = 0; s0 < DC; s0 += 8) {  // eight rows of V_S^+ requested at once: the loop was one L2 round trip per s
    cd v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = vrow[s0 + u];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const cd w = tb[(s0 + u) * DC + b0 + k];
        acc[k].re += v[u].re * w.re - v[u].im * w.im;
        acc[k].im += v[u].re * w.im + v[u].im * w.re;
      }
    }
  }
  const int a = alpha / d, c = alpha % d;
  cd* out = reinterpret_cast<cd*>(choi) + (size_t)b * DC * DC;
  bool nan = false;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int beta = b0 + k, e = beta / d, bb = beta % d;
    out[(a * d + bb) * DC + (c * d + e)] = acc[k];
    nan = nan || !(acc[k].re == acc[k].re);
  }
  if (status) {  // an input state without counts makes its frequencies NaN (process.py:285 divides by the sum)
    const bool any = __syncthreads_or(nan);
    if (threadIdx.x == 0) status[b] = any ? 4 : 0;
  }
  if (iters && threadIdx.x == 0) iters[b] = 0;
}

// The whole factored linear inversion of one process on the FP64 matrix cores (round 3; process.py:284-289 with the
// factored operator above): FOUR 256-thread workgroups per process, workgroup j the 16 columns
//     beta(j, t) = (t % 8) d + 2 j + t / 8,  t = 0..15      (beta = e d + b: every e, b in {2 j, 2 j + 1})
// of T and X -- exactly the rows (a, b in {2 j, 2 j + 1}) of the Choi matrix, so nothing is exchanged between them.
//   stage 1  T[s][beta] = (sum_m n[s][m] V_P^+[beta][m]) / N_s   real x complex: wavefront w the 16 input states
//            16 w .. 16 w + 15, two accumulator tiles (re, im) x two chains, K = M in steps of 4.  The counts are the A
//            operand as they come (int64 -> double in the load path), the normalisation by N_s = sum_m n[s][m]
//            (process.py:285) is applied to the 8 results per lane instead of the M operands: 0 / 0 = NaN as there.
//   stage 2  X[alpha][beta] = sum_s V_S^+[alpha][s] T[s][beta]   complex x complex (three real products, three chains):
//            T through a 17 KB LDS tile, V_S^+ straight from L2 (requested when stage 1's count buffers are free).
//   store    Choi[(a d + b)][(c d + e)] = X[(a d + c)][(e d + b)]: a lane's 16 bytes sit in 128-byte runs (e = t % 8).
// `vp_perm` [4][M][32] is V_P^+ with the columns of each workgroup side by side (re x 16 | im x 16): k_vp_perm
// (qt_process.h), once per set-up.  Measured at B = 64: k_lifp_freq + k_gemm + k_lifp_kron_finish took 14.7 + 32.9 + 40.9 us (the first product
// one wavefront per tile with a load round trip per k-step, the second a scalar FMA loop); see profiles/README.md.
__global__ void __launch_bounds__(256) k_lifp64(const int64_t* __restrict__ counts, int B, int M,
                                                const double* __restrict__ vp_perm, const double* __restrict__ vs_pinv,
                                                double* __restrict__ choi, int32_t* __restrict__ status,
                                                int32_t* __restrict__ iters) {
  constexpr int DC = 64, PT = 17, CH = 9;  // k-steps per request; TWO requests are in flight while a third is consumed
  __shared__ cd tl[DC * PT];
  const int b = blockIdx.x >> 2, j = blockIdx.x & 3;
  if (b >= B) return;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
  const cd* vrow = reinterpret_cast<const cd*>(vs_pinv) + (size_t)(16 * w + r16) * DC + kq;
  const int64_t* crow = counts + ((size_t)b * DC + 16 * w + r16) * M + kq;
  const double* bp = vp_perm + ((size_t)j * M + kq) * 32 + r16;
  const int KS = M >> 2;  // (the host sends M % 4 == 0 here)
  sc_v4f64 tre[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}}, tim[2] = {{0.0, 0.0, 0.0, 0.0}, {0.0, 0.0, 0.0, 0.0}};
  long long a0[CH], a1[CH], a2[CH];  // (raw counts: converted when they are consumed, so that a request does not wait for its data)
  double p0[CH], q0[CH], p1[CH], q1[CH], p2[CH], q2[CH];
  double rs = 0.0;
  auto request = [&](int k0, long long (&a)[CH], double (&p)[CH], double (&q)[CH]) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      a[u] = crow[4 * (k0 + u)];
      p[u] = bp[(size_t)(k0 + u) * 128];
      q[u] = bp[(size_t)(k0 + u) * 128 + 16];
    }
  };
  auto consume = [&](const long long (&a)[CH], const double (&p)[CH], const double (&q)[CH]) {
#pragma unroll
    for (int u = 0; u < CH; ++u) {
      const double n = (double)a[u];
      rs += n;
      tre[u & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(n, p[u], tre[u & 1], 0, 0, 0);
      tim[u & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(n, q[u], tim[u & 1], 0, 0, 0);
    }
  };
  const int nfull = KS / CH;  // (every condition below is uniform over the workgroup)
  // (round 3b: with ONE request ahead a workgroup alone on its CU -- B = 64 -- waited for nine load round trips in sequence,
  //  20 us per launch; two ahead through three register buffers)
  if (nfull > 0) request(0, a0, p0, q0);
  if (nfull > 1) request(CH, a1, p1, q1);
  for (int c = 0; c < nfull; c += 3) {
    if (c + 2 < nfull) request((c + 2) * CH, a2, p2, q2);
    consume(a0, p0, q0);
    if (c + 3 < nfull) request((c + 3) * CH, a0, p0, q0);
    if (c + 1 < nfull) consume(a1, p1, q1);
    if (c + 4 < nfull) request((c + 4) * CH, a1, p1, q1);
    if (c + 2 < nfull) consume(a2, p2, q2);
  }
  for (int k = nfull * CH; k < KS; ++k) {  // M / 4 not a multiple of CH: the last k-steps one at a time
    const double n = (double)crow[4 * k];
    rs += n;
    tre[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(n, bp[(size_t)k * 128], tre[0], 0, 0, 0);
    tim[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(n, bp[(size_t)k * 128 + 16], tim[0], 0, 0, 0);
  }
  cd va[DC / 4];  // (the second product's A operand, requested now that the count buffers are free: 64 registers)
#pragma unroll
  for (int k = 0; k < DC / 4; ++k) va[k] = vrow[4 * k];
  rs += __shfl_xor(rs, 16);
  rs += __shfl_xor(rs, 32);  // N_s of state 16 w + r16 (a sum of integers: exact in any order)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = kq + 4 * r;
    const double tot = __shfl(rs, row);
    tl[(16 * w + row) * PT + r16] = cd{(tre[0][r] + tre[1][r]) / tot, (tim[0][r] + tim[1][r]) / tot};
  }
  __syncthreads();
  // (three real products per complex one, as in the CP step: 48 instead of 64 matrix instructions per wavefront)
  const sc_v4f64 z4 = {0.0, 0.0, 0.0, 0.0};
  sc_v4f64 x1 = z4, x2 = z4, x3 = z4;
#pragma unroll
  for (int k = 0; k < DC / 4; ++k) {
    const cd a = va[k], t = tl[(4 * k + kq) * PT + r16];
    x1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, t.re, x1, 0, 0, 0);
    x2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, t.im, x2, 0, 0, 0);
    x3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re + a.im, t.re + t.im, x3, 0, 0, 0);
  }
  cd* out = reinterpret_cast<cd*>(choi) + (size_t)b * DC * DC;
  const int e = r16 & 7, bb = 2 * j + (r16 >> 3);
  bool nan = false;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int alpha = 16 * w + kq + 4 * r, a = alpha >> 3, c = alpha & 7;
    const cd x{x1[r] - x2[r], x3[r] - x1[r] - x2[r]};
    out[(a * 8 + bb) * DC + c * 8 + e] = x;
    nan = nan || !(x.re == x.re);
  }
  // an input state without counts makes its frequencies NaN (process.py:285 divides by the sum): its row of T is NaN and
  // with it every element of X, in every one of the four workgroups -- workgroup 0 reports
  const bool any = __syncthreads_or(nan);
  if (j == 0 && threadIdx.x == 0) {
    if (status) status[b] = any ? 4 : 0;
    if (iters) iters[b] = 0;
  }
}

// 'pgdb' at n = 3 (process.py:291-314, the arithmetic as k_pgdb_batch restates it for n <= 2) through the FACTORED
// design matrix: with X[(a, c)][(e, b)] = Choi[(a, b)][(c, e)] the model and the gradient are four small products,
//     p[s][m] = Re sum_{alpha, beta} V_S[s][alpha] V_P[m][beta] X[alpha][beta]   =  Re (V_S (X V_P^T))[s][m]
//     g_X     = -conj(V_S^T (W V_P)),   W[s][m] = n[s][m] / p[s][m]
// (1.8 M complex multiply-adds each way per process instead of 56 M against the 906 MB dense operator).  An iteration is
// three launches over the batch -- k_pgdb64_grad (p, W, g, the trial point c - g / mu), k_cptp_project64 (the Dykstra
// projection, unchanged), k_pgdb64_step (direction, q = L Dir, backtracking line search, stopping rule) -- with the
// per-process loop state in global memory; processes that have stopped return at once.  One 256-thread workgroup per
// process; Y = X V_P^T (64 x M complex) goes through a global workspace, everything else through one 64 x 65 LDS image.
struct Pgdb64 {
  static constexpr int DC = 64, d = 8, NT = 256, P = DC + 1, NE = DC * DC;
  static constexpr size_t kLdsBytes = ((size_t)2 * DC * P + 64) * sizeof(double);
  // per-process workspace, in doubles: Y[64][M] complex | p[R] | q[R] | w[R] | g[64][64] complex (Choi layout)
  __host__ __device__ static size_t ws_doubles(int M) { return (size_t)2 * DC * M + 3 * (size_t)DC * M + 2 * NE; }

  __device__ static double bsum(double* red, double v) {  // identical bits in every thread
    v = gsum<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
  }

  // X image (pitch P) of a Choi matrix held row-major in global memory
  __device__ static void load_x(const cd* choi, cd* X) {
    for (int k = threadIdx.x; k < NE; k += NT) {
      const int row = k >> 6, col = k & 63, a = row >> 3, b = row & 7, c = col >> 3, e = col & 7;
      X[(a * d + c) * P + e * d + b] = choi[k];
    }
  }

  // out[s][m] = Re (V_S (X V_P^T))[s][m]; X in LDS, Y in global memory; ends with a barrier
  __device__ static void forward(const cd* X, const cd* __restrict__ vs, const cd* __restrict__ vp, int M, cd* Y, double* out) {
    const int row = threadIdx.x >> 2, mc = (M + 3) / 4, m0 = (threadIdx.x & 3) * mc, m1 = m0 + mc < M ? m0 + mc : M;
    __syncthreads();
    for (int m = m0; m < m1; ++m) {
      const cd* v = vp + (size_t)m * DC;
      double re = 0.0, im = 0.0;
#pragma unroll 8
      for (int beta = 0; beta < DC; ++beta) {
        const cd x = X[row * P + beta], u = v[beta];
        re += x.re * u.re - x.im * u.im;
        im += x.re * u.im + x.im * u.re;
      }
      Y[(size_t)row * M + m] = cd{re, im};
    }
    __syncthreads();
    const cd* vrow = vs + (size_t)row * DC;
    for (int m = m0; m < m1; ++m) {
      double re = 0.0;
#pragma unroll 8
      for (int alpha = 0; alpha < DC; ++alpha) {
        const cd u = vrow[alpha], y = Y[(size_t)alpha * M + m];
        re += u.re * y.re - u.im * y.im;
      }
      out[row * M + m] = re;
    }
    __syncthreads();
  }

  // g (Choi layout, row-major) = -conj(V_S^T (W V_P)) shuffled back; Z = W V_P staged in the LDS image
  __device__ static void adjoint(const double* __restrict__ w, const cd* __restrict__ vs, const cd* __restrict__ vp, int M, cd* Z,
                                 cd* g) {
    const int row = threadIdx.x >> 2, b0 = (threadIdx.x & 3) * 16;
    cd acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = cd{0.0, 0.0};
    for (int m = 0; m < M; ++m) {
      const double wm = w[row * M + m];
      const cd* v = vp + (size_t)m * DC + b0;
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        acc[k].re += wm * v[k].re;
        acc[k].im += wm * v[k].im;
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) Z[row * P + b0 + k] = acc[k];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = cd{0.0, 0.0};
    for (int s = 0; s < DC; ++s) {
      const cd u = vs[(size_t)s * DC + row];  // V_S[s][alpha], alpha = row
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const cd z = Z[s * P + b0 + k];
        acc[k].re += u.re * z.re - u.im * z.im;
        acc[k].im += u.re * z.im + u.im * z.re;
      }
    }
    const int a = row / d, c = row % d;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int beta = b0 + k, e = beta / d, bb = beta % d;
      g[(a * d + bb) * DC + (c * d + e)] = cd{-acc[k].re, acc[k].im};
    }
    __syncthreads();
  }
};

// The model p = Re(V_S (X V_P^T)) and the NLL -sum n log|p + 1e-12| of ONE Choi matrix on the matrix cores, tiled over the
// POVM rows (round 3): workgroup (c, tile) takes the 16 columns m0 .. m0 + 15 of p -- both products of a column tile are
// independent of the other tiles -- and leaves its share of the NLL in fpart[c][tile]; the caller sums the tiles in order.
// Pgdb64::forward runs the same two products as scalar loops in ONE workgroup: 0.7 ms of the 0.9 ms of a chain step.
//   stage 1  Y[alpha][m] = sum_beta X[alpha][beta] V_P[m][beta]      wavefront w: rows 16 w .. 16 w + 15; three real products
//   stage 2  p[s][m]     = Re sum_alpha V_S[s][alpha] Y[alpha][m]    wavefront w: states 16 w ..; the B operand of k-step k
//            is accumulator element k % 4 of wavefront k / 4 at the SAME lane: Y passes through LDS as [wave][r][lane]
struct Fwd64 {
  static constexpr int DC = 64, P = DC + 1;
  static constexpr int oX = 0, oY = 2 * DC * P, oRed = oY + 2 * 4 * 4 * 64, kDoubles = oRed + 8;
  static constexpr size_t kLdsBytes = (size_t)kDoubles * sizeof(double);
  __host__ __device__ static int tiles(int M) { return (M + 15) / 16; }
};

__global__ void __launch_bounds__(256) k_fwd64_nll(const int64_t* __restrict__ counts, int C, int M,
                                                   const double* __restrict__ vs, const double* __restrict__ vp,
                                                   const double* __restrict__ choi_a, const double* __restrict__ choi_b,
                                                   int use_b, double* __restrict__ fpart) {
  extern __shared__ __attribute__((aligned(16))) double smf[];
  const int nt = Fwd64::tiles(M), c = blockIdx.x / nt, tile = blockIdx.x % nt;
  if (c >= C) return;
  constexpr int DC = Fwd64::DC, P = Fwd64::P;
  cd* X = reinterpret_cast<cd*>(smf + Fwd64::oX);
  cd* yl = reinterpret_cast<cd*>(smf + Fwd64::oY);
  double* red = smf + Fwd64::oRed;
  const cd* src = reinterpret_cast<const cd*>(use_b ? choi_b : choi_a) + (size_t)c * DC * DC;
  Pgdb64::load_x(src, X);
  __syncthreads();
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4;
  const int m0 = tile * 16, mrow = (m0 + r16 < M) ? m0 + r16 : M - 1;  // (columns past M: a valid row, ignored below)
  const sc_v4f64 z = {0.0, 0.0, 0.0, 0.0};
  {
    sc_v4f64 p1 = z, p2 = z, p3 = z;
    const cd* ap = X + (16 * w + r16) * P + kq;
    const cd* bp = reinterpret_cast<const cd*>(vp) + (size_t)mrow * DC + kq;
#pragma unroll
    for (int k = 0; k < DC / 4; ++k) {
      const cd a = ap[4 * k], b = bp[4 * k];
      p1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, b.re, p1, 0, 0, 0);
      p2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.im, b.im, p2, 0, 0, 0);
      p3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re + a.im, b.re + b.im, p3, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) yl[(w * 4 + r) * 64 + lane] = cd{p1[r] - p2[r], p3[r] - p1[r] - p2[r]};
  }
  __syncthreads();
  sc_v4f64 q0 = z, q1 = z;
  {
    const cd* ap = reinterpret_cast<const cd*>(vs) + (size_t)(16 * w + r16) * DC + kq;
#pragma unroll
    for (int k = 0; k < DC / 4; ++k) {
      const cd a = ap[4 * k], y = yl[k * 64 + lane];  // (k = 4 wave + r: element r of wavefront `wave`)
      q0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.re, y.re, q0, 0, 0, 0);
      q1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.im, y.im, q1, 0, 0, 0);
    }
  }
  const int64_t* cnt = counts + (size_t)c * DC * M;
  double part = 0.0;
  if (m0 + r16 < M) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int st = 16 * w + kq + 4 * r;
      part += (double)cnt[(size_t)st * M + m0 + r16] * log(fabs((q0[r] + q1[r]) + 1e-12));
    }
  }
  part = gsum<64>(part);
  if (lane == 0) red[w] = part;
  __syncthreads();
  if (threadIdx.x == 0) fpart[(size_t)c * nt + tile] = (red[0] + red[1]) + (red[2] + red[3]);
}

// the accept test of a chain step on the tiles' partial sums (k_mhmc64_accept's second half); t = -1: the starting point
__global__ void __launch_bounds__(256) k_mhmc64_decide(int C, int nt, int T_steps, int t, const double* __restrict__ fpart,
                                                       const double* __restrict__ proposal,
                                                       const double* __restrict__ uniforms, double* __restrict__ x,
                                                       double* __restrict__ fcur, double* __restrict__ chain_out,
                                                       int32_t* __restrict__ accepted) {
  const int c = blockIdx.x;
  if (c >= C) return;
  double sum = 0.0;
  for (int k = 0; k < nt; ++k) sum += fpart[(size_t)c * nt + k];  // the same order in every thread
  const double fn = -sum;
  if (t < 0) {
    if (threadIdx.x == 0) fcur[c] = fn;
    return;
  }
  using S = Pgdb64;
  cd* cur = reinterpret_cast<cd*>(x) + (size_t)c * S::NE;
  const cd* prop = reinterpret_cast<const cd*>(proposal) + (size_t)c * S::NE;
  const double f = fcur[c];
  const bool acc = uniforms[(size_t)c * T_steps + t] <= exp(f - fn);
  cd* out = reinterpret_cast<cd*>(chain_out) + ((size_t)c * T_steps + t) * S::NE;
  for (int k = threadIdx.x; k < S::NE; k += 256) {
    const cd v = acc ? prop[k] : cur[k];
    out[k] = v;
    if (acc) cur[k] = v;
  }
  __syncthreads();  // (every thread has read fcur[c])
  if (threadIdx.x == 0) {
    if (acc) fcur[c] = fn;
    accepted[(size_t)c * T_steps + t] = acc ? 1 : 0;
  }
}

// state[b] = {iteration, stopped, NaN seen, -}
__global__ void __launch_bounds__(256) k_pgdb64_init(int B, double* __restrict__ choi, int32_t* __restrict__ state,
                                                     int32_t* __restrict__ iters, int32_t* __restrict__ status,
                                                     int32_t* __restrict__ n_active) {
  const int b = blockIdx.x;
  if (b >= B) return;
  cd* c = reinterpret_cast<cd*>(choi) + (size_t)b * Pgdb64::NE;
  for (int k = threadIdx.x; k < Pgdb64::NE; k += 256) c[k] = cd{(k >> 6) == (k & 63) ? 1.0 / Pgdb64::DC : 0.0, 0.0};
  if (threadIdx.x < 4) state[4 * b + threadIdx.x] = 0;
  if (threadIdx.x == 0) {
    if (iters) iters[b] = 0;
    if (status) status[b] = 0;
    if (b == 0) *n_active = B;
  }
}

__global__ void __launch_bounds__(Pgdb64::NT) k_pgdb64_grad(const int64_t* __restrict__ counts, int B, int M,
                                                           const double* __restrict__ vs, const double* __restrict__ vp,
                                                           const double* __restrict__ choi, const int32_t* __restrict__ state,
                                                           double* __restrict__ ws, double* __restrict__ trial) {
  extern __shared__ __attribute__((aligned(16))) double smp[];
  const int b = blockIdx.x;
  if (b >= B || state[4 * b + 1]) return;
  using S = Pgdb64;
  const int R = S::DC * M;
  cd* X = reinterpret_cast<cd*>(smp);
  double* wsb = ws + (size_t)b * S::ws_doubles(M);
  cd* Y = reinterpret_cast<cd*>(wsb);
  double* p = wsb + (size_t)2 * S::DC * M;
  double* w = p + 2 * (size_t)R;
  cd* g = reinterpret_cast<cd*>(w + R);
  const cd* cur = reinterpret_cast<const cd*>(choi) + (size_t)b * S::NE;
  const cd *VS = reinterpret_cast<const cd*>(vs), *VP = reinterpret_cast<const cd*>(vp);
  S::load_x(cur, X);
  S::forward(X, VS, VP, M, Y, p);
  for (int r = threadIdx.x; r < R; r += S::NT) w[r] = (double)counts[(size_t)b * R + r] / p[r];
  __syncthreads();
  S::adjoint(w, VS, VP, M, X, g);
  const double mu = 1.5 / S::DC;
  cd* t = reinterpret_cast<cd*>(trial) + (size_t)b * S::NE;
  for (int k = threadIdx.x; k < S::NE; k += S::NT) t[k] = cd{cur[k].re - g[k].re / mu, cur[k].im - g[k].im / mu};
}

__global__ void __launch_bounds__(Pgdb64::NT) k_pgdb64_step(const int64_t* __restrict__ counts, int B, int M,
                                                           const double* __restrict__ vs, const double* __restrict__ vp,
                                                           const double* __restrict__ proj, int n_iter, double tol,
                                                           int stop_rule, double* __restrict__ choi, int32_t* __restrict__ state,
                                                           double* __restrict__ ws, int32_t* __restrict__ iters,
                                                           int32_t* __restrict__ status, int32_t* __restrict__ n_active) {
  extern __shared__ __attribute__((aligned(16))) double smp[];
  const int b = blockIdx.x;
  if (b >= B || state[4 * b + 1]) return;
  using S = Pgdb64;
  const int R = S::DC * M;
  cd* X = reinterpret_cast<cd*>(smp);
  double* red = smp + 2 * S::DC * S::P;
  double* wsb = ws + (size_t)b * S::ws_doubles(M);
  cd* Y = reinterpret_cast<cd*>(wsb);
  const double* p = wsb + (size_t)2 * S::DC * M;
  double* q = wsb + (size_t)2 * S::DC * M + R;
  const cd* g = reinterpret_cast<const cd*>(wsb + (size_t)2 * S::DC * M + 3 * (size_t)R);
  cd* cur = reinterpret_cast<cd*>(choi) + (size_t)b * S::NE;
  const cd* pr = reinterpret_cast<const cd*>(proj) + (size_t)b * S::NE;
  const cd *VS = reinterpret_cast<const cd*>(vs), *VP = reinterpret_cast<const cd*>(vp);
  cd dir[S::NE / S::NT];
  double dgp = 0.0;
#pragma unroll
  for (int u = 0; u < S::NE / S::NT; ++u) {
    const int k = threadIdx.x + u * S::NT;
    dir[u] = cd{pr[k].re - cur[k].re, pr[k].im - cur[k].im};
    dgp += dir[u].re * g[k].re - dir[u].im * g[k].im;  // numpy.dot(Dir, grad): no conjugation, real part
    const int row = k >> 6, col = k & 63, a = row >> 3, bb = row & 7, c = col >> 3, e = col & 7;
    X[(a * S::d + c) * S::P + e * S::d + bb] = dir[u];
  }
  S::forward(X, VS, VP, M, Y, q);
  const double dg = S::bsum(red, dgp);
  const int64_t* cnt = counts + (size_t)b * R;
  auto nll_at = [&](double alpha) {
    double part = 0.0;
    for (int r = threadIdx.x; r < R; r += S::NT) part += (double)cnt[r] * log(fabs(p[r] + alpha * q[r] + 1e-12));
    return -S::bsum(red, part);
  };
  const double gamma = 0.3;
  const double f0 = nll_at(0.0);
  double alpha = 1.0, f1 = nll_at(1.0);
  for (int hh = 0; hh < 1100 && (f1 - f0 > gamma * alpha * dg); ++hh) {
    alpha *= 0.5;
    f1 = nll_at(alpha);
  }
  const bool bad = !(f0 == f0) || !(f1 == f1);
  const int it = state[4 * b];
  bool take, stop;
  int final_it;
  if (stop_rule == 0) {  // the reference leaves when a step LOWERS the NLL by more than tol, without taking it
    stop = f0 - f1 > tol;
    take = !stop;
    final_it = it;
  } else {
    take = true;
    stop = !(f0 - f1 > tol);
    final_it = it + 1;
  }
  if (!stop && it + 1 >= n_iter) stop = true, final_it = n_iter;
  if (take) {
#pragma unroll
    for (int u = 0; u < S::NE / S::NT; ++u) {
      const int k = threadIdx.x + u * S::NT;
      cur[k] = cd{cur[k].re + alpha * dir[u].re, cur[k].im + alpha * dir[u].im};
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int was_bad = state[4 * b + 2] | (bad ? 1 : 0);
    state[4 * b + 2] = was_bad;
    state[4 * b] = it + 1;
    if (stop) {
      state[4 * b + 1] = 1;
      if (iters) iters[b] = final_it;
      if (status) status[b] = was_bad ? 4 : 0;
      atomicSub(n_active, 1);
    }
  }
}

// The Metropolis-Hastings chain of MHMCProcessInterval at n = 3 (interval.py:808-836; the arithmetic as k_mhmc_process
// restates it for n <= 2): step t proposes x' = P_CPTP(x + step * delta_t) and accepts iff u_t <= exp(nll(x) - nll(x')),
// nll = -sum n log|L x + 1e-12| with L applied through its factors.  Four launches per step -- k_mhmc64_propose,
// k_cptp_project64, k_fwd64_nll (the model and the NLL, tiled over the POVM rows on the matrix cores), k_mhmc64_decide --
// with the chain's point and its NLL in global memory; nothing returns to the host between steps.  t = -1 evaluates the NLL
// of the starting point.
__global__ void __launch_bounds__(256) k_mhmc64_propose(int C, int T_steps, int t, double step, const double* __restrict__ x,
                                                        const double* __restrict__ deltas, double* __restrict__ trial) {
  const int c = blockIdx.x;
  if (c >= C) return;
  const cd* cur = reinterpret_cast<const cd*>(x) + (size_t)c * Pgdb64::NE;
  cd* out = reinterpret_cast<cd*>(trial) + (size_t)c * Pgdb64::NE;
  const double* dl = deltas + ((size_t)c * T_steps + t) * Pgdb64::NE;
  for (int k = threadIdx.x; k < Pgdb64::NE; k += 256) {
    const int row = k >> 6, col = k & 63;
    out[k] = cd{cur[k].re + step * dl[col * Pgdb64::DC + row], cur[k].im};  // delta is indexed like the column-stacked vector
  }
}

}  // namespace qt
