// Process-tomography kernels (reference quantpy/tomography/process.py).
//
//   design matrix   rows vec(rho_in (x) E_m^T), column stacking          process.py:203-208
//   linear inversion Choi = vec2mat(A^+ f), f per-input-state frequencies  process.py:284-286
//   CPTP projection  Dykstra alternation of the TP and CP projections      process.py:231-278
//
// One workgroup reconstructs one process: thread (r, c) owns element C[r][c] of the DC x DC Choi
// matrix (DC = 4^n: 4 or 16; DC = 64 has its own layout, qt_process64.h) through the whole Dykstra loop; the TP step is a partial-trace
// reduction in LDS, the CP step a parallel-order Jacobi eigensolver in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_jacobi_wg.h"
#include "qt_signclip_wg.h"
#include "qt_ops.h"    // v4f64
#include "qt_small.h"  // gsum

namespace qt {

struct ProcessState {
  void* lifp = nullptr;      // [D*M][D^2] complex design matrix
  void* pinvT = nullptr;     // [D*M][D^2] complex: transpose of its left inverse
  void* pinvR = nullptr;     // the same with each row's D^2 entries in ROW-major Choi order (k_lifp_gemm's operand: its product
                             // columns are then the doubles of choi[b] in order); built for n = 2 only
  void* emats = nullptr;     // [M][d][d] complex POVM elements
  void* in_states = nullptr; // [D][d][d] complex
  void* aug = nullptr;       // [D^2][2 D^2] complex Gauss-Jordan workspace
  void* pinv = nullptr;      // [D^2][D*M] complex
  // n = 3 (qt_process64.h): the design matrix stays Kronecker-factored -- left inverses of its two factors
  void* vs_pinv = nullptr;   // [D][D] complex: left inverse of V_S = [vec rho_s]
  void* vp_pinv = nullptr;   // [D][M] complex: left inverse of V_P = [vec E_m] (index e d + b)
  void* vp_pinvT = nullptr;  // [M][D] complex: its transpose, the right-hand operand of T = F V_P^+^T
  void* vp_perm = nullptr;   // [groups][M][32] real: the same, 16 columns (re | im) per group in the order k_lifp64 (n = 3:
                             // 4 groups) / k_lifp16 (n = 2: 1 group) store them; M % 4 == 0 only
  bool factored = false;
  size_t cap_rows = 0;
  void release() {
    factored = false;
    for (void** p : {&lifp, &pinvT, &pinvR, &emats, &in_states, &aug, &pinv, &vs_pinv, &vp_pinv, &vp_pinvT, &vp_perm}) {
      if (*p) (void)hipFree(*p);
      *p = nullptr;
    }
    cap_rows = 0;
  }
};

// lifp[(s*M + m)][col*d^2 + row] = rho_s[a][c] * E_m[e][b],  row = a*d + b, col = c*d + e
__global__ void k_lifp_rows(int d, int M, const double* __restrict__ in_states, const double* __restrict__ emats,
                            double* __restrict__ lifp) {
  const int D = d * d;
  const size_t ncol = (size_t)D * D;
  const size_t total = (size_t)D * M * ncol;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t v = t % ncol, rr = t / ncol;
    const int s = (int)(rr / M), m = (int)(rr % M);
    const int row = (int)(v % D), col = (int)(v / D);
    const int a = row / d, b = row % d, c = col / d, e = col % d;
    const double* rho = in_states + ((size_t)s * D + a * d + c) * 2;
    const double* em = emats + ((size_t)m * D + e * d + b) * 2;
    lifp[2 * t] = rho[0] * em[0] - rho[1] * em[1];
    lifp[2 * t + 1] = rho[0] * em[1] + rho[1] * em[0];
  }
}

template <int NT>
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = gsum<64>(v);  // DPP: 8 vector instructions instead of 12 ds_bpermute round trips
  constexpr int NW = (NT + 63) / 64;
  if (NW == 1) return v;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
  __syncthreads();
  double s = 0.0;
#pragma unroll
  for (int w = 0; w < NW; ++w) s += scratch[w];
  return s;
}

// K sums at once (same order of additions per sum as block_sum): one barrier pair for all of them
template <int NT, int K>
__device__ __forceinline__ void block_sums(double (&v)[K], double* scratch /* [K][NT / 64] */) {
  constexpr int NW = (NT + 63) / 64;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = gsum<64>(v[k]);
  if (NW == 1) return;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) scratch[k * NW + (threadIdx.x >> 6)] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < K; ++k) {
    double s = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) s += scratch[k * NW + w];
    v[k] = s;
  }
}

template <int DC>
struct ProcWG {
  static constexpr int NE = DC * DC;               // matrix elements = active threads
  static constexpr int NT = NE < 64 ? 64 : NE;     // workgroup size
  static constexpr int DQ = (DC == 4) ? 2 : (DC == 16 ? 4 : 8);  // d = sqrt(DC)

  // Row pitch DC + 1 for the DC x DC images: with pitch DC (128 bytes at DC = 16) the column-wise
  // reads of the Jacobi rounds and of the rebuild fall on two banks (8-way conflicts).
  static constexpr int LDC = DC + 1;
  static constexpr int NEP = DC * LDC;
  // The CP step's eigensolver: DC = 16 uses the workgroup Jacobi of qt_jacobi_wg.h (one thread per element, one
  // barrier per round) in its own LDS block `jac`; DC = 4 (16 active threads of one wavefront) keeps the small
  // round-robin version below with the v* / r* / o* arrays.
  static constexpr bool kWgJacobi = DC == 16;
  static constexpr bool kSignClipCP = true;  // false: the workgroup Jacobi eigensolver (the version before round 2)
  using JW = JacobiWG<kWgJacobi ? DC : 16, kWgJacobi ? NE : 256, false>;
  static constexpr int oJimg0 = 0, oJimg1 = oJimg0 + 2 * NE, oJrot = oJimg1 + 2 * NE, oJv = oJrot + 6 * DC,
                       oJlam = oJv + 2 * DC * (DC + 1), oJred = oJlam + DC, kJacDoubles = oJred + 32;
  static constexpr int NEV = kWgJacobi ? 2 : NEP;  // eigenvector images of the small version
  struct Sh {
    double are[NEP], aim[NEP];   // work matrix
    double tre[NEP], tim[NEP];   // column-rotated matrix
    double vre[2][NEV], vim[2][NEV];  // eigenvectors, double buffered (DC = 4)
    double rc[DC], ore[DC], oim[DC];
    double red[16];
    double red6[6 * ((NT + 63) / 64)];
    double rre[DQ * DQ], rim[DQ * DQ];  // reduced (input-space) matrix of the TP step
    alignas(16) double jac[kWgJacobi ? kJacDoubles : 2];
  };

  __device__ __forceinline__ static int partner(int i, int r) {
    if (i == DC - 1) return r;
    if (i == r) return DC - 1;
    int p = (2 * r - i) % (DC - 1);
    if (p < 0) p += DC - 1;
    return p;
  }

  // CP projection (process.py:270-277): eigh on the lower triangle, clip at eps, rebuild.
  // (re, im) = this thread's element (i, j); returns the projected element.
  // warm (DC = 16): in / out -- set when this call's eigensolve may start from the eigenvectors a previous call left
  // in `jac`, and set by a call that ran the eigensolver (see JacobiWG::clip)
  __device__ static void cp_project(Sh& sh, bool act, int i, int j, double& re, double& im, double eps,
                                    bool* warm = nullptr) {
    const int e = i * LDC + j;
    // Hermitian completion from the lower triangle, like LAPACK's zheevd with uplo = 'L'
    if (act) {
      sh.are[e] = re;
      sh.aim[e] = im;
    }
    __syncthreads();
    double ar = 0.0, ai = 0.0, vr = 0.0, vi = 0.0;
    if (act) {
      if (i > j) {
        ar = re;
        ai = im;
      } else if (i == j) {
        ar = re;
      } else {
        ar = sh.are[j * LDC + i];
        ai = -sh.aim[j * LDC + i];
      }
      vr = (i == j) ? 1.0 : 0.0;
    }
    __syncthreads();
    // Positive-definite short cut: if the (completed) matrix has a Cholesky factorisation, every eigenvalue
    // is positive and the clip at eps = 1e-12 moves it by less than eps -- the projection is the
    // identity.  In the Dykstra loop this is the common case once the iterates have settled (C3: the
    // last two of five iterations), and 16 elimination steps replace ~90 Jacobi rounds.
    {
      if (act) {
        sh.tre[e] = ar;
        sh.tim[e] = ai;
      }
      __syncthreads();
      double wr = ar, wi = ai;
      int pd = 1;
      // the trailing block alternates between the t* and the a* images (the a* pair is free again: the completed
      // element is in registers), so a step needs one barrier; unrolled: the image of a step is a constant
#pragma unroll
      for (int k = 0; k < DC; ++k) {
        const double* cre = (k & 1) ? sh.are : sh.tre;
        const double* cim = (k & 1) ? sh.aim : sh.tim;
        double* nre = (k & 1) ? sh.tre : sh.are;
        double* nim = (k & 1) ? sh.tim : sh.aim;
        const double piv = cre[k * LDC + k];
        if (!(piv > eps)) {
          pd = 0;  // uniform: every thread reads the same pivot
          break;
        }
        if (act && i > k && j > k) {
          const double ikr = cre[i * LDC + k], iki = cim[i * LDC + k];
          const double jkr = cre[j * LDC + k], jki = cim[j * LDC + k];
          const double inv = 1.0 / piv;
          wr -= (ikr * jkr + iki * jki) * inv;  // a_ik conj(a_jk) / a_kk
          wi -= (iki * jkr - ikr * jki) * inv;
          nre[e] = wr;
          nim[e] = wi;
        }
        __syncthreads();
      }
      QT_STAMP(2);
      if (pd) {
        re = ar;
        im = ai;
        return;
      }
    }
    if constexpr (kWgJacobi && kSignClipCP) {
      // eigenvalue clip through the matrix sign function on the FP64 matrix cores (qt_signclip_wg.h): three 16 x 17
      // complex images + reduction scratch inside the `jac` block
      // (default at d = 16: four real products on two wavefronts.  Three products on ONE wavefront, rotated over the SIMDs of
      //  co-resident workgroups -- SignClipWG<DC, NE, true> -- measured slower here: CP step 43 -> 49 us per 1024, 4.4 -> 5.1 k
      //  clocks per iteration step: a d = 16 step is bound by the latency of its one wavefront's chain, not by the matrix pipe)
      using SC = SignClipWG<DC, NE>;
      static_assert(3 * 2 * DC * SC::P + 32 <= kJacDoubles, "the sign-clip images fit the Jacobi block");
      const typename SC::Lds o{0, 2 * DC * SC::P, 4 * DC * SC::P, 6 * DC * SC::P};
      const cd out = SC::clip(threadIdx.x, cd{ar, ai}, eps, sh.jac, o, false);
      __syncthreads();
      re = out.re;
      im = out.im;
      return;
    } else if constexpr (kWgJacobi) {
      const typename JW::Lds o{oJimg0, oJimg1, oJrot, oJv, oJlam, oJred};
      const cd out = JW::clip(threadIdx.x, cd{ar, ai}, eps, sh.jac, o, false, warm && *warm);
      if (warm) *warm = true;
      __syncthreads();  // the caller goes on to overwrite the a* / t* images; nothing of `jac` is read after this
      re = out.re;
      im = out.im;
      return;
    } else {  // DC = 4: the small round-robin version
    int cur = 0;
    if (act) {
      sh.are[e] = ar;
      sh.aim[e] = ai;
      sh.vre[0][e] = vr;
      sh.vim[0][e] = vi;
    }
    __syncthreads();
    for (int sweep = 0; sweep < 20; ++sweep) {
      const double n2 = act ? ar * ar + ai * ai : 0.0;
      const double off = block_sum<NT>((act && i != j) ? n2 : 0.0, sh.red);
      const double nrm = block_sum<NT>(n2, sh.red);
      if (!(off > 1e-26 * nrm)) break;  // off/norm <= 1e-13: the clip moves by no more than that (qt_large.h)
      for (int r = 0; r < DC - 1; ++r) {
        if (threadIdx.x < DC / 2) {
          const int t = threadIdx.x;
          int pa, pb;
          if (t == 0) {
            pa = DC - 1;
            pb = r;
          } else {
            pa = (r + t) % (DC - 1);
            pb = (r - t + (DC - 1)) % (DC - 1);
          }
          const int p = pa < pb ? pa : pb, q = pa < pb ? pb : pa;
          const double app = sh.are[p * LDC + p], aqq = sh.are[q * LDC + q];
          const double xr = sh.are[p * LDC + q], xi = sh.aim[p * LDC + q];
          const double ab2 = xr * xr + xi * xi;
          double cs = 1.0, wre = 0.0, wim = 0.0;  // w = s e^{i phi} = c u a_pq  (see qt_small.h)
          if (ab2 > 1e-300) {
            const double dl = 0.5 * (aqq - app);
            const double u = copysign(1.0, dl) / (fabs(dl) + sqrt(dl * dl + ab2));
            cs = 1.0 / sqrt(1.0 + u * u * ab2);
            wre = cs * u * xr;
            wim = cs * u * xi;
          }
          sh.rc[p] = cs;
          sh.rc[q] = cs;
          sh.ore[p] = -wre;
          sh.oim[p] = wim;
          sh.ore[q] = wre;
          sh.oim[q] = wim;
        }
        __syncthreads();
        double t_r = 0.0, t_i = 0.0;
        int pi = 0;
        double ci = 1.0, oir = 0.0, oii = 0.0;
        if (act) {
          const int pj = partner(j, r);
          pi = partner(i, r);
          const double cj = sh.rc[j], ojr = sh.ore[j], oji = sh.oim[j];
          ci = sh.rc[i];
          oir = sh.ore[i];
          oii = sh.oim[i];
          const double br = sh.are[i * LDC + pj], bi = sh.aim[i * LDC + pj];
          t_r = ar * cj + (br * ojr - bi * oji);
          t_i = ai * cj + (br * oji + bi * ojr);
          const double wr = sh.vre[cur][i * LDC + pj], wi = sh.vim[cur][i * LDC + pj];
          const double nvr = vr * cj + (wr * ojr - wi * oji);
          const double nvi = vi * cj + (wr * oji + wi * ojr);
          vr = nvr;
          vi = nvi;
          sh.tre[e] = t_r;
          sh.tim[e] = t_i;
          sh.vre[cur ^ 1][e] = vr;
          sh.vim[cur ^ 1][e] = vi;
        }
        cur ^= 1;
        __syncthreads();
        if (act) {
          const double ur = sh.tre[pi * LDC + j], ui = sh.tim[pi * LDC + j];
          ar = ci * t_r + (oir * ur + oii * ui);  // ci T_ij + conj(o_i) T[pi][j]
          ai = ci * t_i + (oir * ui - oii * ur);
          if (j == pi) {
            ar = 0.0;
            ai = 0.0;
          }
          if (i == j) ai = 0.0;
          sh.are[e] = ar;
          sh.aim[e] = ai;
        }
        __syncthreads();
      }
    }
    if (act) {
      double rr = 0.0, ri = 0.0;
      for (int k = 0; k < DC; ++k) {
        const double lam = sh.are[k * LDC + k];
        const double lc = lam > eps ? lam : eps;
        const double xr = sh.vre[cur][i * LDC + k], xi = sh.vim[cur][i * LDC + k];
        const double yr = sh.vre[cur][j * LDC + k], yi = sh.vim[cur][j * LDC + k];
        rr += lc * (xr * yr + xi * yi);  // V_ik conj(V_jk)
        ri += lc * (xi * yr - xr * yi);
      }
      re = rr;
      im = ri;
    }
    __syncthreads();
    }  // DC = 4
  }

  // TP projection (process.py:259-265) of the matrix whose element (i, j) this thread holds:
  // C[(a,o),(b,o)] += (delta_ab - sum_o' C[(a,o'),(b,o')]) / d ; other entries unchanged.
  __device__ static void tp_project(Sh& sh, bool act, int i, int j, double& re, double& im) {
    const int e = i * LDC + j;
    if (act) {
      sh.tre[e] = re;
      sh.tim[e] = im;
    }
    __syncthreads();
    if (threadIdx.x < DQ * DQ) {
      const int a = threadIdx.x / DQ, b = threadIdx.x % DQ;
      double sr = 0.0, si = 0.0;
      for (int o = 0; o < DQ; ++o) {
        sr += sh.tre[(a * DQ + o) * LDC + (b * DQ + o)];
        si += sh.tim[(a * DQ + o) * LDC + (b * DQ + o)];
      }
      sh.rre[threadIdx.x] = sr;
      sh.rim[threadIdx.x] = si;
    }
    __syncthreads();
    if (act) {
      const int a = i / DQ, o = i % DQ, b = j / DQ, o2 = j % DQ;
      if (o == o2) {
        re += ((a == b ? 1.0 : 0.0) - sh.rre[a * DQ + b]) / DQ;
        im += (0.0 - sh.rim[a * DQ + b]) / DQ;
      }
    }
    __syncthreads();
  }

  // Dykstra alternation (process.py:237-257).  x = this thread's element; returns iterations.
  __device__ static int dykstra(Sh& sh, bool act, int i, int j, double& xr, double& xi, int n_iter, double tol) {
    double pr = 0.0, pim = 0.0, qr = 0.0, qi = 0.0, yr = 0.0, yi = 0.0;
    int it = 0;
    // successive CP steps see nearly the same matrix: from the second eigensolve on, start from the eigenvectors
    // of the previous one (2-3 sweeps instead of ~7); every Dykstra run starts cold, so rounding in V cannot pile up
    bool warm = false;
    for (; it < n_iter; ++it) {
      double tr_ = xr + pr, ti_ = xi + pim;
      tp_project(sh, act, i, j, tr_, ti_);
      const double ydr = tr_ - yr, ydi = ti_ - yi;
      yr += ydr;
      yi += ydi;
      double cr = yr + qr, ci = yi + qi;
      cp_project(sh, act, i, j, cr, ci, 1e-12, &warm);
      const double xdr = cr - xr, xdi = ci - xi;
      xr += xdr;
      xi += xdi;
      // 2 (|sum conj(y_diff) q| + |sum conj(x_diff) p|) + |p_diff|^2 + |q_diff|^2: the six sums share one barrier pair
      const double pdr = xr - yr, pdi = xi - yi;
      const double qdr = yr - xr, qdi = yi - xi;
      double six[6] = {act ? ydr * qr + ydi * qi : 0.0,   act ? ydr * qi - ydi * qr : 0.0,
                       act ? xdr * pr + xdi * pim : 0.0,  act ? xdr * pim - xdi * pr : 0.0,
                       act ? pdr * pdr + pdi * pdi : 0.0, act ? qdr * qdr + qdi * qdi : 0.0};
      block_sums<NT, 6>(six, sh.red6);
      double crit = 2.0 * (hypot(six[0], six[1]) + hypot(six[2], six[3]));
      pr += pdr;
      pim += pdi;
      qr += qdr;
      qi += qdi;
      crit += six[4] + six[5];
      if (crit < tol) {
        ++it;
        break;
      }
    }
    return it;
  }
};

// counts[B][D][M] -> choi[B][DC][DC]; DC = D = 4^n.  pinvT = [D*M][DC*DC] complex (row r of the
// design matrix along the slow axis so that threads read consecutive Choi-vector entries).
// (second launch bound: at least 4 waves per SIMD, i.e. <= 128 VGPRs -- the ILP scheduling strategy the
//  library is built with would otherwise spend 254 registers here and halve the occupancy)
template <int DC>
__global__ void __launch_bounds__(ProcWG<DC>::NT, 4) k_lifp_batch(const int64_t* __restrict__ counts, int B, int M,
                                                              const double* __restrict__ pinvT, int cptp,
                                                              double* __restrict__ choi, int32_t* __restrict__ iters,
                                                              int32_t* __restrict__ status) {
  using W = ProcWG<DC>;
  __shared__ typename W::Sh sh;
  extern __shared__ double sfreq[];  // [DC * M]
  const int b = blockIdx.x;
  if (b >= B) return;
  const int tid = threadIdx.x;
  const bool act = tid < W::NE;
  const int64_t* cb = counts + (size_t)b * DC * M;
  for (int s = 0; s < DC; ++s) {  // per-input-state frequencies (process.py:285)
    double part = 0.0;
    for (int m = tid; m < M; m += W::NT) part += (double)cb[s * M + m];
    const double tot = block_sum<W::NT>(part, sh.red);
    for (int m = tid; m < M; m += W::NT) sfreq[s * M + m] = (double)cb[s * M + m] / tot;
  }
  __syncthreads();
  // thread tid = Choi-vector entry v = col * DC + row  ->  element (row, col)
  double xr = 0.0, xi = 0.0;
  const int row = tid % DC, col = tid / DC;
  if (act) {
    const double* p = pinvT + (size_t)tid * 2;
    const int R = DC * M;
#pragma unroll 4
    for (int r = 0; r < R; ++r) {
      const double f = sfreq[r];
      xr += p[(size_t)r * W::NE * 2] * f;
      xi += p[(size_t)r * W::NE * 2 + 1] * f;
    }
  }
  // re-distribute so that thread (i, j) = (tid / DC, tid % DC) holds C[i][j]: C[row][col] sits in
  // thread col*DC+row, i.e. the transposed position
  __syncthreads();
  if (act) {
    sh.tre[row * W::LDC + col] = xr;
    sh.tim[row * W::LDC + col] = xi;
  }
  __syncthreads();
  const int i = tid / DC, j = tid % DC;
  if (act) {
    xr = sh.tre[i * W::LDC + j];
    xi = sh.tim[i * W::LDC + j];
  }
  __syncthreads();
  int it = 0;
  if (cptp) it = W::dykstra(sh, act, i, j, xr, xi, 1000, 1e-12);
  if (act) {
    double* out = choi + ((size_t)b * W::NE + tid) * 2;
    out[0] = xr;
    out[1] = xi;
  }
  if (tid == 0) {
    if (iters) iters[b] = it;
    if (status) status[b] = (xr == xr) ? 0 : 4;
  }
}

// 'pgdb' (process.py:291-308): projected gradient descent with backtracking on the Choi vector,
//   c_0 = vec(I / D)                       (fully_mixed(2n), process.py:292)
//   p = A c,  g = -A^H (n / p)             (n = raw counts; :296-297)
//   Dir = P_CPTP(c - g / mu) - c,  mu = 1.5 / 4^n                      (:298, :293)
//   alpha = 1; while nll(c + alpha Dir) - nll(c) > gamma alpha <Dir, g>: alpha /= 2   (gamma = 0.3; :299-301)
//   nll(c) = -sum n log(A c + 1e-12)                                   (:310-314)
// kept with the reference's arithmetic: <Dir, g> is numpy.dot -- no conjugation, i.e.
// sum (Dr Gr - Di Gi) for the Hermitian pair -- and log of a complex number enters through its
// real part log|z|.  stop_rule 0 is the reference's loop exit (:303-305): it leaves when a step
// LOWERS the NLL by more than tol and returns the point BEFORE that step; stop_rule 1 is the
// evident intent (accept the step, stop when the decrease falls below tol).
// One workgroup per process; thread (i, j) owns Choi element C[i][j]; vectors are column-stacked
// (v = col * DC + row, routines.py:59-61).  A = lifp [R][NE] complex, R = D * M.
template <int DC>
__global__ void __launch_bounds__(ProcWG<DC>::NT, 2) k_pgdb_batch(const int64_t* __restrict__ counts, int B, int M,
                                                              const double* __restrict__ lifp, int n_iter, double tol,
                                                              int stop_rule, double* __restrict__ choi,
                                                              int32_t* __restrict__ iters, int32_t* __restrict__ status) {
  using W = ProcWG<DC>;
  constexpr int NE = W::NE, NT = W::NT, NW = NT / 64;
  __shared__ typename W::Sh sh;
  __shared__ double cre[NE], cim[NE], dre[NE], dim_[NE];  // current point / direction, index v
  extern __shared__ double dynsh[];                       // cnt[R] | p[R] | q[R] | w[R]
  const int R = DC * M;
  double* cnt = dynsh;
  double* pr = cnt + R;
  double* qr = pr + R;
  double* wr = qr + R;
  const int b = blockIdx.x;
  if (b >= B) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool act = tid < NE;
  const int i = tid / DC, j = tid % DC;
  const int v_own = j * DC + i;  // where this thread's element sits in the column-stacked vector
  for (int r = tid; r < R; r += NT) cnt[r] = (double)counts[(size_t)b * R + r];
  double xr = (act && i == j) ? 1.0 / DC : 0.0, xi = 0.0;
  const double mu = 1.5 / DC, gamma = 0.3;
  // Re(A x) for a Hermitian-vector x held in LDS (xre, xim): one wave per row, lanes over columns
  auto row_products = [&](const double* xre, const double* xim, double* out) {
    for (int r = wave; r < R; r += NW) {
      const double* a = lifp + (size_t)r * NE * 2;
      double acc = 0.0;
      for (int v = lane; v < NE; v += 64) acc += a[2 * v] * xre[v] - a[2 * v + 1] * xim[v];
      for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
      if (lane == 0) out[r] = acc;
    }
  };
  auto nll_at = [&](double alpha) {
    double part = 0.0;
    for (int r = tid; r < R; r += NT) part += cnt[r] * log(fabs(pr[r] + alpha * qr[r] + 1e-12));
    return -block_sum<NT>(part, sh.red);
  };
  int it = 0;
  int bad = 0;
  for (; it < n_iter; ++it) {
    __syncthreads();
    if (act) {
      cre[v_own] = xr;
      cim[v_own] = xi;
    }
    __syncthreads();
    row_products(cre, cim, pr);
    __syncthreads();
    for (int r = tid; r < R; r += NT) wr[r] = cnt[r] / pr[r];
    __syncthreads();
    // g_v = -sum_r conj(A[r][v]) w_r   (thread v; consecutive threads read consecutive columns)
    double gr = 0.0, gi = 0.0;
    if (act) {
      for (int r = 0; r < R; ++r) {
        const double w = wr[r];
        gr -= lifp[((size_t)r * NE + tid) * 2] * w;
        gi += lifp[((size_t)r * NE + tid) * 2 + 1] * w;
      }
      dre[tid] = gr;  // parked in the direction buffer, index v = tid
      dim_[tid] = gi;
    }
    __syncthreads();
    double gmr = 0.0, gmi = 0.0;  // gradient element of this thread's (i, j)
    if (act) {
      gmr = dre[v_own];
      gmi = dim_[v_own];
    }
    __syncthreads();
    double tr_ = xr - gmr / mu, ti_ = xi - gmi / mu;
    W::dykstra(sh, act, i, j, tr_, ti_, 1000, 1e-12);
    const double ddr = tr_ - xr, ddi = ti_ - xi;
    if (act) {
      dre[v_own] = ddr;
      dim_[v_own] = ddi;
    }
    __syncthreads();
    row_products(dre, dim_, qr);
    const double dg = block_sum<NT>(act ? ddr * gmr - ddi * gmi : 0.0, sh.red);  // numpy.dot(D, grad)
    const double f0 = nll_at(0.0);
    double alpha = 1.0, f1 = nll_at(1.0);
    for (int h = 0; h < 1100 && (f1 - f0 > gamma * alpha * dg); ++h) {
      alpha *= 0.5;
      f1 = nll_at(alpha);
    }
    if (!(f0 == f0) || !(f1 == f1)) bad = 1;
    if (stop_rule == 0) {
      if (f0 - f1 > tol) break;  // the reference leaves here, WITHOUT taking the step
      xr += alpha * ddr;
      xi += alpha * ddi;
    } else {
      xr += alpha * ddr;
      xi += alpha * ddi;
      if (!(f0 - f1 > tol)) {
        ++it;
        break;
      }
    }
  }
  if (act) {
    double* out = choi + ((size_t)b * NE + tid) * 2;
    out[0] = xr;
    out[1] = xi;
  }
  if (tid == 0) {
    if (iters) iters[b] = it;
    if (status) status[b] = bad ? 4 : 0;
  }
}

// Metropolis-Hastings chain on the Choi vector (MHMCProcessInterval, interval.py:808-836; mhmc.py:80-119
// with update_rule = `_cptp_update_rule`, process.py:279-281): step t proposes
//   x' = P_CPTP(x + step * delta_t)      (delta real, added entry by entry to the column-stacked vector)
// and accepts iff u_t <= exp(nll(x) - nll(x')), nll = -sum n log(A x + 1e-12) (process.py:310-314; the
// complex logarithm enters through its real part, as NumPy orders complex numbers by it).
// One workgroup per chain; chain_out[c][t] = Choi matrix (row-major) after step t.
template <int DC>
__global__ void __launch_bounds__(ProcWG<DC>::NT, 2) k_mhmc_process(const int64_t* __restrict__ counts, int C, int M,
                                                                const double* __restrict__ lifp,
                                                                const double* __restrict__ choi_init,
                                                                const double* __restrict__ deltas,
                                                                const double* __restrict__ uniforms, int T_steps,
                                                                double step, double* __restrict__ chain_out,
                                                                int32_t* __restrict__ accepted) {
  using W = ProcWG<DC>;
  constexpr int NE = W::NE, NT = W::NT, NW = NT / 64;
  __shared__ typename W::Sh sh;
  __shared__ double cre[NE], cim[NE];
  extern __shared__ double dynsh[];  // cnt[R] | p[R]
  const int R = DC * M;
  double* cnt = dynsh;
  double* pr = cnt + R;
  const int b = blockIdx.x;
  if (b >= C) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool act = tid < NE;
  const int i = tid / DC, j = tid % DC;
  const int v_own = j * DC + i;
  for (int r = tid; r < R; r += NT) cnt[r] = (double)counts[(size_t)b * R + r];
  double xr = 0.0, xi = 0.0;
  if (act) {
    xr = choi_init[((size_t)b * NE + tid) * 2];
    xi = choi_init[((size_t)b * NE + tid) * 2 + 1];
  }
  auto nll_of = [&](double er, double ei) {  // nll of the matrix whose element (i, j) this thread passes in
    __syncthreads();
    if (act) {
      cre[v_own] = er;
      cim[v_own] = ei;
    }
    __syncthreads();
    for (int r = wave; r < R; r += NW) {
      const double* a = lifp + (size_t)r * NE * 2;
      double acc = 0.0;
      for (int v = lane; v < NE; v += 64) acc += a[2 * v] * cre[v] - a[2 * v + 1] * cim[v];
      for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
      if (lane == 0) pr[r] = acc;
    }
    __syncthreads();
    double part = 0.0;
    for (int r = tid; r < R; r += NT) part += cnt[r] * log(fabs(pr[r] + 1e-12));
    return -block_sum<NT>(part, sh.red);
  };
  double f = nll_of(xr, xi);
  for (int t = 0; t < T_steps; ++t) {
    double pr_ = xr, pi_ = xi;
    if (act) pr_ += step * deltas[((size_t)b * T_steps + t) * NE + v_own];
    W::dykstra(sh, act, i, j, pr_, pi_, 1000, 1e-12);
    const double fn = nll_of(pr_, pi_);
    const bool acc = uniforms[(size_t)b * T_steps + t] <= exp(f - fn);
    if (acc) {
      xr = pr_;
      xi = pi_;
      f = fn;
    }
    if (act) {
      double* out = chain_out + (((size_t)b * T_steps + t) * NE + tid) * 2;
      out[0] = xr;
      out[1] = xi;
    }
    if (tid == 0) accepted[(size_t)b * T_steps + t] = acc ? 1 : 0;
  }
}

// ---- 'lifp' over a batch as a GEMM on the FP64 matrix cores (n = 2) -------------------------------------------
// For many processes at once, Choi-vector = pinv . freq is [B x R] . [R x 2 NE] (freq real, pinv complex with
// re / im interleaved: exactly the layout of pinvT).  k_lifp_batch reads the whole 2.4 MB operand from L2 for
// every process (24 TB/s at B = 1024: the L2 is the limit); here a workgroup keeps a 16-column slice of it in
// LDS ([R][16] doubles, 74 KB at R = 576) and its wavefronts take 16 processes at a time through
// v_mfma_f64_16x16x4_f64, streaming their frequency rows as the A operand (64 k-values requested ahead).
//
// pinvR[r][er DC + ec] = pinvT[r][ec DC + er] (complex): the column-stacked Choi vector of routines.py:59-61 re-ordered
// to the row-major matrix, once per set-up, so that the GEMM below writes choi[b] as contiguous 128-byte runs AND stages
// contiguous operand slices.  Measured at B = 1024 (k_lifp_freq + k_lifp_gemm per call): 34-36 us with the column-stacked
// operand (16-byte result pieces 256 bytes apart), 40 us when the slices were gathered on the fly from pinvT instead
// (eight times the L2 lines per slice), 32 us with this copy.  Also tried on top and dropped: the frequencies formed
// inside the GEMM's A-operand path from the raw counts and per-state inverse totals (no k_lifp_freq launch, no
// frequency matrix): 34 us -- the int64 -> double conversions and the scaling sit in the MFMA loop at 256 VGPRs.
__global__ void k_choi_order_rows(const double* __restrict__ pinvT, size_t rows, int DC, double* __restrict__ pinvR) {
  const size_t ne = (size_t)DC * DC, total = rows * ne;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < total; t += (size_t)gridDim.x * blockDim.x) {
    const size_t r = t / ne, v = t % ne;
    const size_t er = v / DC, ec = v % DC;
    const double2 x = *reinterpret_cast<const double2*>(pinvT + (r * ne + ec * DC + er) * 2);
    *reinterpret_cast<double2*>(pinvR + t * 2) = x;
  }
}

// Step 1: counts [B][DC][M] -> frequencies [B][R = DC M], normalised per input state (process.py:285);
// 16 lanes per (process, input state) row.
__global__ void __launch_bounds__(256) k_lifp_freq(const int64_t* __restrict__ counts, int rows, int M, int DC, int Rp,
                                                   double* __restrict__ F) {
  // 16 lanes per row (M = 36 at C3: three loads per lane, all in flight), 16 rows per workgroup.
  // F is [B][Rp]: a process's R = DC M frequencies, then zeros up to the pitch Rp (a multiple of 64), so that
  // k_lifp_gemm never reads one process's numbers into another's product (a NaN frequency -- an input state without
  // counts -- must stay that process's own); 192 zeros follow the last process for its look-ahead loads.
  const int row = blockIdx.x * 16 + (threadIdx.x >> 4), l = threadIdx.x & 15;
  const int nproc = rows / DC, R = DC * M;
  if (blockIdx.x == 0 && threadIdx.x < 192) F[(size_t)nproc * Rp + threadIdx.x] = 0.0;
  const bool live = row < rows;
  const int64_t* c = counts + (size_t)(live ? row : 0) * M;
  double part = 0.0;
  for (int m = l; m < M; m += 16) part += (double)c[m];
  const double tot = gsum<16>(part);  // sums of integers: exact in any order
  if (!live) return;
  const int b = row / DC, st = row - b * DC;
  double* f = F + (size_t)b * Rp;
  for (int m = l; m < M; m += 16) f[st * M + m] = (double)c[m] / tot;
  if (st == DC - 1)
    for (int k = R + l; k < Rp; k += 16) f[k] = 0.0;
}

// Step 2: grid = (2 NE / 16 column slices, blocks of 64 processes); R % 16 == 0 (R = 16 M).  Operand layout of the MFMA
// as in k_born_mfma (qt_ops.h): lane (r16, kq) supplies A[r16][kq] and B[kq][r16], and receives rows
// kq + 4 r of column r16.  Column n = 2 v + part of the product is the re / im part of Choi-vector entry
// v = col * DC + row (routines.py:59-61), i.e. of element (row, col): written straight to choi[b][row][col].
// NC = 16-column tiles per workgroup.  NC = 2 (round 2): a wavefront feeds each A fragment (16 processes x 4 k-values,
// streamed from global memory) into TWO column tiles, so the frequency matrix F is re-read by 16 instead of 32 column
// slices (at B = 1024 the 32 slices pulled 151 MB of F through L2 for a 0.6 GFLOP product -- the kernel was bound by
// that, not by the matrix cores) and every wavefront carries two independent MFMA chains.  LDS: [Rp][16 NC] doubles
// (147 KB at R = 576, NC = 2: one workgroup per CU, 16 x 16 = 256 workgroups at B = 1024) + the split-K scratch.
template <int DC, int NC, int DIAG = 0>
__global__ void __launch_bounds__(512) k_lifp_gemm(const double* __restrict__ F, int B, int R, int Rp,
                                                   const double* __restrict__ pinvR, double* __restrict__ choi,
                                                   int32_t* __restrict__ status, int32_t* __restrict__ zero_iters) {
  constexpr int NE = DC * DC, N = 2 * NE, W = 16 * NC;
  extern __shared__ double s_p[];  // [Rp][W] (rows R .. Rp-1 zero), then 4 x NC x 256 doubles for the split-K sum
  // The operand is pinvR: product column n is double n of choi[b] (row-major, re / im interleaved), so a workgroup's
  // W columns are a contiguous run of a Choi row and a wavefront's results go out as 128-byte runs (round 3).
  const int c0 = blockIdx.x * W;
  // DIAG (profile build only, scripts/gemm_diag.py): phases switched off at compile time to see what bounds the kernel --
  // 1: no MFMAs, 2: no A-operand loads, 4: no slice staging, 8: no result stores.  The product is instantiated with 0.
  constexpr int diag = DIAG;
  // Staging the slice is the longest phase of the kernel (147 KB per workgroup against ~2 us of MFMA work): eighteen
  // 16-byte loads in flight per thread -- the whole slice at R = 576 in ONE round trip to L2 (it was three rounds of
  // twelve 8-byte loads).  W is even and c0 a multiple of W, so a pair never straddles a row.
  for (int e0 = 2 * threadIdx.x; e0 < Rp * W; e0 += 18 * 1024) {
    double2 t[18];
#pragma unroll
    for (int u = 0; u < 18; ++u) {
      const int e = e0 + u * 1024;
      t[u] = (e < R * W && !(diag & 4)) ? *reinterpret_cast<const double2*>(pinvR + (size_t)(e / W) * N + c0 + (e % W)) : double2{0.0, 0.0};
    }
#pragma unroll
    for (int u = 0; u < 18; ++u) {
      const int e = e0 + u * 1024;
      if (e < Rp * W) *reinterpret_cast<double2*>(s_p + e) = t[u];
    }
  }
  __syncthreads();
  // 8 wavefronts = 4 groups of 16 processes x 2 halves of K: two wavefronts per tile double the number of
  // independent MFMA chains per SIMD (a tile's MFMAs all accumulate into the same registers)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int grp = wave & 3, half = wave >> 2;
  const int kh = ((Rp / 2 + 63) / 64) * 64;  // first half: [0, kh), second: [kh, Rp); both whole 64-chunks
  const int kbeg = half ? kh : 0, kend = half ? Rp : kh;
  const int nblocks = (B + 63) / 64;
  for (int blk = blockIdx.y; blk < nblocks; blk += gridDim.y) {  // uniform per workgroup
  const int g = blk * 4 + grp;
  const int row = g * 16 + r16;
  // k-values of MFMA step u in lane group kq:  k0 + 8 (u / 2) + 2 kq + ((u ^ kq) & 1).  The A operand then comes
  // in 16-byte pairs: the four lane groups of a row read one full 64-byte line per load (the plain
  // k0 + 4 u + kq order makes every load 16 half-used lines of 8 bytes per lane), and the swapped order inside
  // the pairs of the odd lane groups keeps the B-operand rows of lane groups {0, 1} and {2, 3} on opposite
  // halves of the LDS banks.  No bounds tests in the loop: the slice is zero-padded to
  // Rp rows and so is every row of F (pitch Rp); the look-ahead of a row's last chunk reads the start of the next
  // row (or the 192 zeros behind the last one) and is not used.  Rows beyond B compute on row 0 and are not stored.
  const double2* fr = reinterpret_cast<const double2*>(F + (size_t)(row < B ? row : 0) * Rp) + kq;
  const bool swap = kq & 1;
  v4f64 acc[NC];
#pragma unroll
  for (int c = 0; c < NC; ++c) acc[c] = v4f64{0.0, 0.0, 0.0, 0.0};
  // The A operand runs TWO chunks ahead of the MFMAs (a chunk's 16 / 32 MFMAs take ~1 000 clocks, an L2 round trip
  // of a wavefront with one neighbour on its SIMD several times that: one chunk of look-ahead left the matrix cores
  // waiting -- 28 us for 0.6 GFLOP).  Look-aheads past the end wrap to the first chunk and are not used (no branch:
  // a branch would let the compiler sink the loads below the MFMA chain).
  double2 a[8], an[8], an2[8];
  const int k1 = kbeg + 64 < kend ? kbeg + 64 : kbeg;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    a[j] = fr[(kbeg + 8 * j) / 2];
    an[j] = fr[(k1 + 8 * j) / 2];
  }
  for (int k0 = kbeg; k0 < kend; k0 += 64) {
    const int kn = k0 + 128 < kend ? k0 + 128 : kbeg;
#pragma unroll
    for (int j = 0; j < 8; ++j) an2[j] = (diag & 2) ? a[j] : fr[(kn + 8 * j) / 2];
    // all B-operand reads of the chunk first (two base addresses, immediate offsets), then the MFMA chains
    const double* b_first = s_p + (k0 + 2 * kq + (swap ? 1 : 0)) * W + r16;
    const double* b_second = s_p + (k0 + 2 * kq + (swap ? 0 : 1)) * W + r16;
    double b0[NC][8], b1[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        b0[c][j] = b_first[8 * j * W + 16 * c];
        b1[c][j] = b_second[8 * j * W + 16 * c];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const double lo = swap ? a[j].y : a[j].x, hi = swap ? a[j].x : a[j].y;
      if constexpr ((diag & 1) != 0) {  // (profile build only) one vector FMA per operand pair keeps the loads alive
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c][0] = fma(lo, b0[c][j], fma(hi, b1[c][j], acc[c][0]));
        continue;
      }
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(lo, b0[c][j], acc[c], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < NC; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(hi, b1[c][j], acc[c], 0, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = an[j];
      an[j] = an2[j];
    }
  }
  double* red = s_p + Rp * W + grp * 256 * NC;
  if (half) {
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(c * 4 + r) * 64 + lane] = acc[c][r];
  }
  __syncthreads();
  if (!half) {
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int n = c0 + 16 * c + r16;  // double n of the process's Choi matrix
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int orow = g * 16 + kq + 4 * r;
        if (orow < B && !((diag & 8) && n != 0)) {
          const double x = acc[c][r] + red[(c * 4 + r) * 64 + lane];
          choi[(size_t)orow * N + n] = x;
          if (n == 0) {  // Re C[0][0], the element k_lifp_batch tests; no projection follows when these are set
            if (status) status[orow] = (x == x) ? 0 : 4;
            if (zero_iters) zero_iters[orow] = 0;
          }
        }
      }
    }
  }
  __syncthreads();  // `red` is reused by the next block of processes
  }
}

// mode 0: Dykstra CPTP, 1: TP only, 2: CP only  (process.py:231-278)
// ---- Linear inversion through the Kronecker factors of the design matrix (round 3) -------------------------------
// process.py:203-208 builds rows vec(rho_s (x) E_m^T); up to a fixed permutation of the columns that is
// vec(rho_s) (x) vec(E_m), and with PLAIN transposes (routines.py:69-71) the left inverse of a Kronecker product is the
// Kronecker product of the left inverses (qt_process64.h has the derivation; n = 3 has always run this way):
//     X = V_S^+ . F . V_P^+^T,      Choi[(a d + b)][(c d + e)] = X[(a d + c)][(e d + b)],
// V_S = [vec rho_s] (D x D), V_P = [vec E_m] (M x D), F[s][m] the frequencies.  At n = 2 that is 34 matrix instructions
// per process (real x complex 16 x M x 16, then complex 16 x 16 x 16) where the dense operator costs 288 -- the dense
// GEMM of k_lifp_gemm reads a 2.4 MB operand per 64 processes to do 8.5 x the arithmetic -- and the kernel becomes what
// SURVEY 8d expected of this path: a stream of counts in (8 D M bytes per process) and Choi matrices out (16 D^2).
//
// ONE wavefront per process, no LDS traffic between the products: v_mfma_f64_16x16x4_f64 leaves rows kq + 4 r of a
// 16 x 16 tile in accumulator element r of lane (kq, r16) -- exactly the element the B operand of k-step r wants from that
// lane (row 4 r + kq) -- so T = F V_P^+^T feeds X = V_S^+ T straight from its accumulators.  The counts are the A operand
// of the first product as they come (int64 -> double on the way), the normalisation by N_s (process.py:285) is applied
// to T's 8 numbers per lane (0 / 0 = NaN as there); V_P^+^T waits in LDS (`vp_perm`: [M][32], columns ordered
// beta(t) = (t % d) d + t / d so that a lane's result sits in 64-byte runs of a Choi row), V_S^+ in registers.
// KSC = M / 4 at compile time (9 for the 'proj-set' POVM: every load of a process in flight at once) or 0.
__global__ void k_vp_perm(const double* __restrict__ vp_pinvT, int M, int D, int d, int groups, double* __restrict__ vp_perm) {
  const int total = groups * M * 32, per = 16 / d;  // `per` values of b per group of 16 columns
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int j = t / (M * 32), m = (t / 32) % M, c = t & 31, part = c >> 4, tt = c & 15;
    const int beta = (tt % d) * d + per * j + tt / d;
    vp_perm[t] = vp_pinvT[((size_t)m * D + beta) * 2 + part];
  }
}

template <int KSC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3))) k_lifp16(const int64_t* __restrict__ counts, int B, int M,
                                                const double* __restrict__ vp_perm, const double* __restrict__ vs_pinv,
                                                double* __restrict__ choi, int32_t* __restrict__ status,
                                                int32_t* __restrict__ iters) {
  constexpr int DC = 16;
  extern __shared__ double s_vp[];  // [M][32]
  for (int k = threadIdx.x; k < M * 32; k += 256) s_vp[k] = vp_perm[k];
  const int lane = threadIdx.x & 63, r16 = lane & 15, kq = lane >> 4, w = threadIdx.x >> 6;
  cd va[4];  // A operand of the second product: V_S^+[alpha = r16][s = 4 r + kq]
#pragma unroll
  for (int r = 0; r < 4; ++r) va[r] = reinterpret_cast<const cd*>(vs_pinv)[r16 * DC + 4 * r + kq];
  __syncthreads();
  const int KS = KSC > 0 ? KSC : (M >> 2);
  const double* bp = s_vp + kq * 32 + r16;
  const v4f64 z = {0.0, 0.0, 0.0, 0.0};
  const int stride = gridDim.x * 4;
  long long nx[KSC > 0 ? KSC : 1];  // the NEXT process's counts: requested before this one's products start
  if constexpr (KSC > 0) {
    const int b0 = blockIdx.x * 4 + w;
    if (b0 < B) {
      const int64_t* c0 = counts + ((size_t)b0 * DC + r16) * M + kq;
#pragma unroll
      for (int k = 0; k < KSC; ++k) nx[k] = c0[4 * k];
    }
  }
  for (int b = blockIdx.x * 4 + w; b < B; b += stride) {  // (wave-uniform)
    const int64_t* crow = counts + ((size_t)b * DC + r16) * M + kq;
    v4f64 tre[2] = {z, z}, tim[2] = {z, z};
    double rs = 0.0;
    if constexpr (KSC > 0) {
      long long n[KSC];
#pragma unroll
      for (int k = 0; k < KSC; ++k) n[k] = nx[k];
      if (b + stride < B) {
        const int64_t* cn = crow + (size_t)stride * DC * M;
#pragma unroll
        for (int k = 0; k < KSC; ++k) nx[k] = cn[4 * k];
      }
#pragma unroll
      for (int k = 0; k < KSC; ++k) {
        const double dn = (double)n[k];
        rs += dn;
        tre[k & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(dn, bp[k * 128], tre[k & 1], 0, 0, 0);
        tim[k & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(dn, bp[k * 128 + 16], tim[k & 1], 0, 0, 0);
      }
    } else {
      for (int k = 0; k < KS; ++k) {
        const double dn = (double)crow[4 * k];
        rs += dn;
        tre[k & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(dn, bp[k * 128], tre[k & 1], 0, 0, 0);
        tim[k & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(dn, bp[k * 128 + 16], tim[k & 1], 0, 0, 0);
      }
    }
    rs += __shfl_xor(rs, 16);
    rs += __shfl_xor(rs, 32);  // N_s of input state r16 (a sum of integers: exact in any order)
    v4f64 xr1 = z, xr2 = z, xi1 = z, xi2 = z;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const double tot = __shfl(rs, kq + 4 * r);
      const double tr = (tre[0][r] + tre[1][r]) / tot, ti = (tim[0][r] + tim[1][r]) / tot;  // T[4 r + kq][beta(r16)]
      xr1 = __builtin_amdgcn_mfma_f64_16x16x4f64(va[r].re, tr, xr1, 0, 0, 0);
      xr2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-va[r].im, ti, xr2, 0, 0, 0);
      xi1 = __builtin_amdgcn_mfma_f64_16x16x4f64(va[r].re, ti, xi1, 0, 0, 0);
      xi2 = __builtin_amdgcn_mfma_f64_16x16x4f64(va[r].im, tr, xi2, 0, 0, 0);
    }
    cd* out = reinterpret_cast<cd*>(choi) + (size_t)b * DC * DC;
    const int e = r16 & 3, bb = r16 >> 2;
    bool nan = false;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int alpha = kq + 4 * r, a = alpha >> 2, c = alpha & 3;
      const cd x{xr1[r] + xr2[r], xi1[r] + xi2[r]};
      out[(a * 4 + bb) * DC + c * 4 + e] = x;
      nan = nan || !(x.re == x.re);
    }
    const bool any = __any(nan);  // (an input state without counts: NaN frequencies, process.py:285)
    if (lane == 0) {
      if (status) status[b] = any ? 4 : 0;
      if (iters) iters[b] = 0;
    }
  }
}

template <int DC>
__global__ void __launch_bounds__(ProcWG<DC>::NT, 4) k_cptp_project(const double* __restrict__ in, int B, int mode,
                                                                int n_iter, double tol, double* __restrict__ out,
                                                                int32_t* __restrict__ iters,
                                                                int32_t* __restrict__ status = nullptr) {
  using W = ProcWG<DC>;
  __shared__ typename W::Sh sh;
  const int b = blockIdx.x;
  if (b >= B) return;
  const int tid = threadIdx.x;
  const bool act = tid < W::NE;
  const int i = tid / DC, j = tid % DC;
  double xr = 0.0, xi = 0.0;
  if (act) {
    xr = in[((size_t)b * W::NE + tid) * 2];
    xi = in[((size_t)b * W::NE + tid) * 2 + 1];
  }
  int it = 0;
  QT_STAMP(0);
  if (mode == 0) it = W::dykstra(sh, act, i, j, xr, xi, n_iter, tol);
  else if (mode == 1) W::tp_project(sh, act, i, j, xr, xi);
  else W::cp_project(sh, act, i, j, xr, xi, 1e-12);
  QT_STAMP(1);
  if (act) {
    out[((size_t)b * W::NE + tid) * 2] = xr;
    out[((size_t)b * W::NE + tid) * 2 + 1] = xi;
  }
  if (tid == 0) {
    if (iters) iters[b] = it;
    if (status) status[b] = (xr == xr) ? 0 : 4;
  }
}

}  // namespace qt
