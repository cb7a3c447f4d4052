// Wave-synchronous kernels for n <= 3 qubits (D = 4^n <= 64 = one wavefront).
//
// Mapping: one trial is owned by a group of G = D lanes of a 64-wide wavefront (64/G trials per
// wave; at n = 3 one trial per wave).  Lane l of a group plays three roles at once:
//   - Bloch / parameter index k = l  (vectors b, w, x, g, p_k of length D live one element/lane),
//   - matrix element (i, j) = (l / d, l % d) of every d x d complex matrix (rho, L, G, V),
//   - row m = c*G + l of the M-row POVM contraction, chunk by chunk.
// Cross-lane traffic goes through a per-trial LDS scratch (broadcast reads) and xor-butterfly
// shuffles for reductions; a workgroup is exactly one wave, so LDS hand-offs need only the
// wave-level fence in wave_sync().  The BFGS inverse Hessian (D x D f64) lives in registers,
// one row per lane (128 VGPRs at n = 3), so nothing but the read-only operands (A', A'^T,
// left inverse: 111 KB each at n = 3, L2 resident) and the counts / rho of the trial touch HBM.
//
// Reference semantics implemented (paths into /root/reference/quantpy):
//   lin:  tomography/state.py:191-202, PSD clip :267-273
//   chol: routines.py:84-101
//   nll:  tomography/state.py:217-229 (value); gradient derived analytically
//   mle:  tomography/state.py:204-215 with scipy's BFGS control flow (qt_linesearch.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_linesearch.h"

namespace qt {

struct cd {
  double re, im;
};
__device__ __forceinline__ cd cmul(cd a, cd b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cd cmulc(cd a, cd b) {  // a * conj(b)
  return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im};
}
__device__ __forceinline__ cd cadd(cd a, cd b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cd cscale(cd a, double s) { return {a.re * s, a.im * s}; }

// LDS hand-off inside one wavefront: order this wave's LDS writes before its later reads.
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

template <int G>
__device__ __forceinline__ double gsum(double v) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
  return v;
}
template <int G>
__device__ __forceinline__ double gmax(double v) {
#pragma unroll
  for (int m = G / 2; m >= 1; m >>= 1) {
    double o = __shfl_xor(v, m, 64);
    v = (o > v || o != o) ? o : v;  // NaN wins, like np.max
  }
  return v;
}

struct PovmView {
  const double* Aw;    // [M][D]  shot-weighted A'
  const double* AwT;   // [D][M]
  const double* PinvT; // [M][D]  transpose of the left inverse of A'
  int M;
};

template <int NQ>
struct Small {
  static constexpr int d = 1 << NQ;
  static constexpr int D = d * d;
  static constexpr int G = D;
  static constexpr int TPW = 64 / G;
  static constexpr int T = d * (d - 1) / 2;
  // per-trial LDS layout, in doubles
  static constexpr int oA = 0;             // complex [d][d]
  static constexpr int oB = oA + 2 * D;    // complex [d][d]
  static constexpr int oV0 = oB + 2 * D;   // complex [d][d]
  static constexpr int oV1 = oV0 + 2 * D;  // complex [d][d]
  static constexpr int oVec = oV1 + 2 * D; // [D]
  static constexpr int oRot = oVec + D;    // [4 d]: c, off.re, off.im, (pad)
  static constexpr int oM = oRot + 4 * d;  // rbuf[Mp], freq[Mp]
  __host__ __device__ static int lds_doubles(int M) {
    int Mp = (M + 1) & ~1;
    return oM + 2 * Mp;
  }

  // ---- per-lane context ---------------------------------------------------------------
  struct Ctx {
    int l, i, j;     // lane in group, matrix element
    double* sm;      // this trial's LDS scratch
    int M, Mp;
    PovmView pv;
    // Pauli string k = l:  P_k[r][r ^ xm] = (-i)^ny (-1)^popc(r & zm)
    int xm, zm, ny;
    // Cholesky parameter owned by lane l: element (pi, pj), kind 0 diag / 1 real / 2 imag
    int pi, pj, pkind;
    __device__ __forceinline__ cd* A() const { return reinterpret_cast<cd*>(sm + oA); }
    __device__ __forceinline__ cd* Bm() const { return reinterpret_cast<cd*>(sm + oB); }
    __device__ __forceinline__ cd* V0() const { return reinterpret_cast<cd*>(sm + oV0); }
    __device__ __forceinline__ cd* V1() const { return reinterpret_cast<cd*>(sm + oV1); }
    __device__ __forceinline__ double* vec() const { return sm + oVec; }
    __device__ __forceinline__ double* rot() const { return sm + oRot; }
    __device__ __forceinline__ double* rbuf() const { return sm + oM; }
    __device__ __forceinline__ double* freq() const { return sm + oM + Mp; }
  };

  __device__ static void make_ctx(Ctx& c, double* smem_block, const PovmView& pv) {
    const int lane = threadIdx.x & 63;
    c.l = lane % G;
    const int tib = lane / G;
    c.i = c.l / d;
    c.j = c.l % d;
    c.M = pv.M;
    c.Mp = (pv.M + 1) & ~1;
    c.pv = pv;
    c.sm = smem_block + tib * lds_doubles(pv.M);
    int xm = 0, zm = 0, ny = 0;
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
      const int dig = (c.l >> (2 * b)) & 3;
      if (dig == 1 || dig == 2) xm |= 1 << b;
      if (dig == 2 || dig == 3) zm |= 1 << b;
      if (dig == 2) ++ny;
    }
    c.xm = xm;
    c.zm = zm;
    c.ny = ny & 3;
    if (c.l < d) {
      c.pi = c.pj = c.l;
      c.pkind = 0;
    } else {
      int t = c.l - d;
      c.pkind = 1;
      if (t >= T) {
        t -= T;
        c.pkind = 2;
      }
      int ii = 1;
      while ((ii * (ii + 1)) / 2 <= t) ++ii;  // row of np.tril_indices(d, -1)[t]
      c.pi = ii;
      c.pj = t - (ii * (ii - 1)) / 2;
    }
  }

  // Bloch index of the Pauli string with X-type mask x and Z-type mask z.
  __device__ __forceinline__ static int pauli_index(int x, int z) {
    int k = 0;
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
      const int xb = (x >> b) & 1, zb = (z >> b) & 1;
      const int dig = xb ? (zb ? 2 : 1) : (zb ? 3 : 0);
      k |= dig << (2 * b);
    }
    return k;
  }

  // Re[(-i)^ny * s]
  __device__ __forceinline__ static double re_phase(int ny, cd s) {
    return ny == 0 ? s.re : ny == 1 ? s.im : ny == 2 ? -s.re : -s.im;
  }
  // (-i)^ny * v for real v
  __device__ __forceinline__ static cd phase_times(int ny, double v) {
    return ny == 0 ? cd{v, 0.0} : ny == 1 ? cd{0.0, -v} : ny == 2 ? cd{-v, 0.0} : cd{0.0, v};
  }

  // b_k = Re Tr(P_k M^dagger) / d for the matrix held in LDS `m` (qobj.py:132).
  __device__ static double bloch_of(const Ctx& c, const cd* m) {
    cd s{0.0, 0.0};
#pragma unroll
    for (int r = 0; r < d; ++r) {
      const cd e = m[r * d + (r ^ c.xm)];  // conj(M[r][r^x]) summed with the sign of P_k[r][r^x]
      const double sg = (__popc(r & c.zm) & 1) ? -1.0 : 1.0;
      s.re += sg * e.re;
      s.im -= sg * e.im;
    }
    return re_phase(c.ny, s) / d;
  }

  // Element (i, j) of sum_k v[k] P_k, v in LDS (qobj.py:114-117).
  __device__ static cd matrix_of(const Ctx& c, const double* v) {
    const int x = c.i ^ c.j;
    cd s{0.0, 0.0};
#pragma unroll
    for (int z = 0; z < d; ++z) {
      const int k = pauli_index(x, z);
      const int ny = __popc(x & z) & 3;
      const double sg = (__popc(c.i & z) & 1) ? -1.0 : 1.0;
      s = cadd(s, phase_times(ny, sg * v[k]));
    }
    return s;
  }

  // counts of this trial -> freq[] in LDS (counts / sum(counts): state.py:193, :227)
  __device__ static void load_freq(const Ctx& c, const int64_t* counts) {
    double part = 0.0;
    for (int m = c.l; m < c.M; m += G) part += (double)counts[m];
    const double tot = gsum<G>(part);
    for (int m = c.l; m < c.M; m += G) c.freq()[m] = (double)counts[m] / tot;
    wave_sync();
  }

  // ---- a6: linear inversion.  Returns lane's element of rho; vec() holds the Bloch vector.
  __device__ static cd lin_invert(const Ctx& c, double& bloch_l) {
    double acc = 0.0;
    const double* P = c.pv.PinvT + c.l;
    const double* fr = c.freq();
#pragma unroll 8
    for (int m = 0; m < c.M; ++m) acc += P[(size_t)m * D] * fr[m];
    bloch_l = acc / d;
    c.vec()[c.l] = bloch_l;
    wave_sync();
    cd r = matrix_of(c, c.vec());
    wave_sync();
    return r;
  }

  __device__ __forceinline__ static int partner(int i, int r) {
    if (d == 2) return 1 - i;
    if (i == d - 1) return r;
    if (i == r) return d - 1;
    int p = 2 * r - i;
    p %= (d - 1);
    if (p < 0) p += d - 1;
    return p;
  }

  // ---- a7: eigenvalue clip + trace renormalisation by a parallel-order cyclic Jacobi.
  // In: lane's element of a Hermitian matrix.  Out: lane's element of U max(v, eps) U^dagger / Tr.
  __device__ static cd psd_project(const Ctx& c, cd a, double eps) {
    cd* A = c.A();
    cd* Tm = c.Bm();
    cd* Vc = c.V0();
    cd* Vn = c.V1();
    double* rot = c.rot();
    const int i = c.i, j = c.j;
    cd v{i == j ? 1.0 : 0.0, 0.0};
    if (i == j) a.im = 0.0;
    A[c.l] = a;
    Vc[c.l] = v;
    wave_sync();
    for (int sweep = 0; sweep < 16; ++sweep) {
      const double n2 = a.re * a.re + a.im * a.im;
      const double off = gsum<G>(i != j ? n2 : 0.0);
      const double nrm = gsum<G>(n2);
      const bool done = !(off > 1e-30 * nrm);
      if (__all(done)) break;
#pragma unroll 1
      for (int r = 0; r < (d == 2 ? 1 : d - 1); ++r) {
        // rotation for pair t = l % (d/2) of this round
        {
          const int t = c.l % (d / 2);
          int pa, pb;
          if (t == 0) {
            pa = d - 1;
            pb = (d == 2) ? 0 : r;
          } else {
            pa = (r + t) % (d - 1);
            pb = (r - t + (d - 1)) % (d - 1);
          }
          const int p = pa < pb ? pa : pb, q = pa < pb ? pb : pa;
          const double app = A[p * d + p].re, aqq = A[q * d + q].re;
          const cd apq = A[p * d + q];
          const double ab = hypot(apq.re, apq.im);
          double cs = 1.0, sn = 0.0, ere = 1.0, eim = 0.0;
          if (ab > 1e-290) {
            const double tau = (aqq - app) / (2.0 * ab);
            const double tt = copysign(1.0, tau) / (fabs(tau) + hypot(1.0, tau));
            cs = 1.0 / sqrt(1.0 + tt * tt);
            sn = tt * cs;
            ere = apq.re / ab;
            eim = apq.im / ab;
          }
          if (c.l < d / 2) {
            // off[x] = J[partner(x)][x]:  J_qp = -s e^{-i phi} (x = p),  J_pq = s e^{i phi} (x = q)
            rot[p] = cs;
            rot[q] = cs;
            rot[d + p] = -sn * ere;
            rot[2 * d + p] = sn * eim;
            rot[d + q] = sn * ere;
            rot[2 * d + q] = sn * eim;
          }
        }
        wave_sync();
        const int pj = partner(j, r), pi = partner(i, r);
        const double cj = rot[j], ci = rot[i];
        const cd oj{rot[d + j], rot[2 * d + j]};
        const cd oi{rot[d + i], rot[2 * d + i]};
        // column step: T = A J, V' = V J
        const cd t_ij = cadd(cscale(a, cj), cmul(A[i * d + pj], oj));
        v = cadd(cscale(v, cj), cmul(Vc[i * d + pj], oj));
        Tm[c.l] = t_ij;
        Vn[c.l] = v;
        wave_sync();
        // row step: A' = J^dagger T
        cd an = cadd(cscale(t_ij, ci), cmulc(Tm[pi * d + j], oi));  // ci T_ij + conj(oi) T[pi][j]
        if (j == pi) an = cd{0.0, 0.0};
        if (i == j) an.im = 0.0;
        a = an;
        A[c.l] = a;
        cd* sw = Vc;
        Vc = Vn;
        Vn = sw;
        wave_sync();
      }
    }
    // rebuild with clipped eigenvalues: R_ij = sum_k V_ik max(lam_k, eps) conj(V_jk)
    cd rr{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < d; ++k) {
      const double lam = A[k * d + k].re;
      const double lc = lam > eps ? lam : eps;  // np.maximum(eps, v)
      const cd p = cmulc(Vc[i * d + k], Vc[j * d + k]);
      rr.re += lc * p.re;
      rr.im += lc * p.im;
    }
    const double tr = gsum<G>(i == j ? rr.re : 0.0);
    wave_sync();
    return cd{rr.re / tr, rr.im / tr};
  }

  // ---- a8: lower Cholesky factor of the matrix whose element this lane holds.
  // Leaves L in Bm() (upper part zero) and returns the lane's parameter x_l; ok = 0 if not PD.
  __device__ static double cholesky_param(const Ctx& c, cd a, int& ok) {
    cd* A = c.A();
    cd* L = c.Bm();
    const int i = c.i, j = c.j;
    A[c.l] = a;
    L[c.l] = cd{0.0, 0.0};
    ok = 1;
    wave_sync();
#pragma unroll 1
    for (int k = 0; k < d; ++k) {
      const double akk = A[k * d + k].re;
      if (!(akk > 0.0)) ok = 0;
      const double lkk = sqrt(akk);
      const cd aik = A[i * d + k], ajk = A[j * d + k];
      const cd lik{aik.re / lkk, aik.im / lkk}, ljk{ajk.re / lkk, ajk.im / lkk};
      if (j == k && i >= k) L[c.l] = (i == k) ? cd{lkk, 0.0} : lik;
      if (i > k && j > k) {
        const cd p = cmulc(lik, ljk);
        a.re -= p.re;
        a.im -= p.im;
        A[c.l] = a;
      }
      wave_sync();
    }
    const cd e = L[c.pi * d + c.pj];
    const double x = c.pkind == 2 ? e.im : e.re;
    wave_sync();
    return x;
  }

  // x (one parameter per lane) -> L in Bm(), returns lane's element of L L^dagger and t = Tr.
  __device__ static cd build_llh(const Ctx& c, double xl, double& tr) {
    double* vx = c.vec();
    cd* L = c.Bm();
    vx[c.l] = xl;
    tr = gsum<G>(xl * xl);
    wave_sync();
    const int i = c.i, j = c.j;
    cd lij{0.0, 0.0};
    if (i == j) lij.re = vx[i];
    else if (i > j) {
      const int t = (i * (i - 1)) / 2 + j;
      lij = cd{vx[d + t], vx[d + T + t]};
    }
    L[c.l] = lij;
    wave_sync();
    cd m{0.0, 0.0};
    const int kmax = i < j ? i : j;
#pragma unroll
    for (int k = 0; k < d; ++k)
      if (k <= kmax) m = cadd(m, cmulc(L[i * d + k], L[j * d + k]));
    return m;
  }

  // ---- a9: NLL value and exact gradient at x.  Needs freq[] loaded.  Leaves L in Bm().
  __device__ static void nll_grad(const Ctx& c, double xl, double& f, double& gl) {
    double tr;
    const cd m = build_llh(c, xl, tr);
    cd* A = c.A();
    A[c.l] = cd{m.re / tr, m.im / tr};  // rho
    wave_sync();
    const double bl = bloch_of(c, A);
    double* vec = c.vec();
    vec[c.l] = bl;
    wave_sync();
    // p = d * A' b ;  f = -sum freq log(p + 1e-10) ;  r = freq / (p + 1e-10)
    double fpart = 0.0;
    const double* fr = c.freq();
    double* rb = c.rbuf();
    for (int m0 = 0; m0 < c.M; m0 += G) {
      const int mm = m0 + c.l;
      if (mm < c.M) {
        const double* col = c.pv.AwT + mm;
        double acc = 0.0;
#pragma unroll 8
        for (int k = 0; k < D; ++k) acc += col[(size_t)k * c.M] * vec[k];
        const double pe = acc * d + 1e-10;
        fpart += fr[mm] * log(pe);
        rb[mm] = fr[mm] / pe;
      }
    }
    f = -gsum<G>(fpart);
    wave_sync();
    // w = A'^T r ;  G = -sum_k w_k P_k ;  Gt = (G - Tr(G rho) I) / t
    double wl = 0.0;
    {
      const double* row = c.pv.Aw + c.l;
#pragma unroll 8
      for (int mm = 0; mm < c.M; ++mm) wl += row[(size_t)mm * D] * rb[mm];
    }
    const double tr_g_rho = -(double)d * gsum<G>(wl * bl);
    vec[c.l] = wl;
    wave_sync();
    cd g = matrix_of(c, vec);
    g.re = -g.re;
    g.im = -g.im;
    if (c.i == c.j) g.re -= tr_g_rho;
    g.re /= tr;
    g.im /= tr;
    A[c.l] = g;
    wave_sync();
    // Q = Gt L ; gradient entries 2 Re Q_ii, 2 Re Q_ij, 2 Im Q_ij (i > j)
    const cd* L = c.Bm();
    cd q{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < d; ++k)
      if (k >= c.pj) q = cadd(q, cmul(A[c.pi * d + k], L[k * d + c.pj]));
    gl = 2.0 * (c.pkind == 2 ? q.im : q.re);
    wave_sync();
  }
};

// =========================================================================================
// kernels
// =========================================================================================

// a6 + a7
template <int NQ>
__global__ void __launch_bounds__(64) k_lin_batch(PovmView pv, const int64_t* __restrict__ counts, int B, int physical,
                                                  double* __restrict__ rho, double* __restrict__ bloch_out,
                                                  int32_t* __restrict__ status) {
  using S = Small<NQ>;
  extern __shared__ double smem[];
  typename S::Ctx c;
  S::make_ctx(c, smem, pv);
  const int tib = (threadIdx.x & 63) / S::G;
  const int b = blockIdx.x * S::TPW + tib;
  const bool live = b < B;
  const int bb = live ? b : B - 1;  // idle groups recompute the last trial; nothing is stored
  S::load_freq(c, counts + (size_t)bb * pv.M);
  double bl;
  cd r = S::lin_invert(c, bl);
  if (physical) r = S::psd_project(c, r, 1e-15);
  if (live) {
    double* out = rho + ((size_t)b * S::D + c.l) * 2;
    out[0] = r.re;
    out[1] = r.im;
    if (bloch_out) bloch_out[(size_t)b * S::D + c.l] = bl;
    if (status && c.l == 0) status[b] = (r.re == r.re) ? 0 : 4;
  }
}

// a8 forward: rho -> x
template <int NQ>
__global__ void __launch_bounds__(64) k_chol_param(PovmView pv, const double* __restrict__ rho, int B,
                                                   double* __restrict__ x, int32_t* __restrict__ status) {
  using S = Small<NQ>;
  extern __shared__ double smem[];
  typename S::Ctx c;
  S::make_ctx(c, smem, pv);
  const int tib = (threadIdx.x & 63) / S::G;
  const int b = blockIdx.x * S::TPW + tib;
  const bool live = b < B;
  const int bb = live ? b : B - 1;
  const double* in = rho + ((size_t)bb * S::D + c.l) * 2;
  int ok;
  const double xl = S::cholesky_param(c, cd{in[0], in[1]}, ok);
  if (live) {
    x[(size_t)b * S::D + c.l] = xl;
    if (status && c.l == 0) status[b] = ok ? 0 : 1;
  }
}

// a8 backward: x -> L L^dagger
template <int NQ>
__global__ void __launch_bounds__(64) k_chol_unparam(PovmView pv, const double* __restrict__ x, int B,
                                                     double* __restrict__ llh) {
  using S = Small<NQ>;
  extern __shared__ double smem[];
  typename S::Ctx c;
  S::make_ctx(c, smem, pv);
  const int tib = (threadIdx.x & 63) / S::G;
  const int b = blockIdx.x * S::TPW + tib;
  const bool live = b < B;
  const int bb = live ? b : B - 1;
  double tr;
  const cd m = S::build_llh(c, x[(size_t)bb * S::D + c.l], tr);
  if (live) {
    double* out = llh + ((size_t)b * S::D + c.l) * 2;
    out[0] = m.re;
    out[1] = m.im;
  }
}

// a9
template <int NQ>
__global__ void __launch_bounds__(64) k_nll_batch(PovmView pv, const double* __restrict__ x,
                                                  const int64_t* __restrict__ counts, int B, double* __restrict__ f,
                                                  double* __restrict__ grad) {
  using S = Small<NQ>;
  extern __shared__ double smem[];
  typename S::Ctx c;
  S::make_ctx(c, smem, pv);
  const int tib = (threadIdx.x & 63) / S::G;
  const int b = blockIdx.x * S::TPW + tib;
  const bool live = b < B;
  const int bb = live ? b : B - 1;
  S::load_freq(c, counts + (size_t)bb * pv.M);
  double fv, gl;
  S::nll_grad(c, x[(size_t)bb * S::D + c.l], fv, gl);
  if (live) {
    if (c.l == 0) f[b] = fv;
    if (grad) grad[(size_t)b * S::D + c.l] = gl;
  }
}

// a10: the whole MLE of a trial in one launch.
template <int NQ>
__global__ void __launch_bounds__(64) k_mle_batch(PovmView pv, const int64_t* __restrict__ counts, int B, int init,
                                                  int max_iter, double gtol, double* __restrict__ rho,
                                                  int32_t* __restrict__ nit_out, int32_t* __restrict__ nfev_out,
                                                  double* __restrict__ fun_out, int32_t* __restrict__ status_out) {
  using S = Small<NQ>;
  constexpr int D = S::D, G = S::G, d = S::d;
  extern __shared__ double smem[];
  typename S::Ctx c;
  S::make_ctx(c, smem, pv);
  const int tib = (threadIdx.x & 63) / G;
  const int b = blockIdx.x * S::TPW + tib;
  const bool live = b < B;
  const int bb = live ? b : B - 1;
  S::load_freq(c, counts + (size_t)bb * pv.M);

  // ---- starting point (state.py:205-212)
  cd r0;
  if (init == 0) {
    double bl;
    r0 = S::lin_invert(c, bl);
    r0 = S::psd_project(c, r0, 1e-15);
  } else {
    r0 = cd{c.i == c.j ? 1.0 / d : 0.0, 0.0};
  }
  int ok;
  double xk = S::cholesky_param(c, r0, ok);

  // ---- BFGS (scipy _minimize_bfgs): one (value, gradient) evaluation per loop pass
  double H[D];
#pragma unroll
  for (int k = 0; k < D; ++k) H[k] = (k == c.l) ? 1.0 : 0.0;
  double gk = 0.0, pk = 0.0, fk = 0.0, old_old = 0.0, stp = 0.0;
  int kiter = 0, nfev = 0, status = ok ? 0 : 1;
  bool active = ok != 0;
  int phase = 0;  // 0: first evaluation at x0, 1: inside a line search
  LineSearch ls;
  double* vec = c.vec();
  const int eval_cap = (max_iter + 2) * 130;  // hard stop: every wave leaves the loop

  while (__any(active)) {
    const double xt = (phase == 0) ? xk : xk + stp * pk;
    double ft, gt;
    S::nll_grad(c, xt, ft, gt);  // executed by the whole wave; finished trials idle through it
    bool new_direction = false;
    if (active && ++nfev > eval_cap) {
      status = 2;
      active = false;
    }
    if (!active) {
      // nothing: this trial has finished
    } else if (phase == 0) {
      fk = ft;
      gk = gt;
      old_old = fk + sqrt(gsum<G>(gk * gk)) / 2.0;
      const double gnorm = gmax<G>(fabs(gk));
      if (!(gnorm > gtol) || !(kiter < max_iter)) active = false;
      else new_direction = true;
    } else {
      const double dphi = gsum<G>(gt * pk);
      double next = stp;
      const int r = ls.advance(stp, ft, dphi, &next);
      if (r == LS_EVAL) {
        stp = next;
      } else if (r == LS_FAIL) {
        status = 2;
        active = false;
      } else {
        // step accepted: x += s, y = g_new - g, BFGS update (scipy _optimize.py, rhok = 1000 if y.s == 0)
        const double sk = stp * pk;
        const double pnorm2 = gsum<G>(pk * pk);
        xk = xk + sk;
        const double yk = gt - gk;
        gk = gt;
        old_old = fk;
        fk = ft;
        ++kiter;
        const double gnorm = gmax<G>(fabs(gk));
        if (!(gnorm > gtol)) {
          active = false;
        } else if (stp * sqrt(pnorm2) <= 0.0) {  // xrtol = 0 test
          active = false;
        } else if (!isfinite(fk)) {
          status = 2;
          active = false;
        } else {
          const double ys = gsum<G>(yk * sk);
          const double rhok = (ys == 0.0) ? 1000.0 : 1.0 / ys;
          // H <- (I - rho s y^T) H (I - rho y s^T) + rho s s^T, expanded with u = H y
          vec[c.l] = yk;
          wave_sync();
          double u = 0.0;
#pragma unroll
          for (int k = 0; k < D; ++k) u += H[k] * vec[k];
          const double yhy = gsum<G>(yk * u);
          wave_sync();
          double* ubuf = reinterpret_cast<double*>(c.A());  // 2 D doubles: u | s
          ubuf[c.l] = u;
          ubuf[D + c.l] = sk;
          wave_sync();
          const double cc = rhok * rhok * yhy + rhok;
#pragma unroll
          for (int k = 0; k < D; ++k)
            H[k] += -rhok * (u * ubuf[D + k] + sk * ubuf[k]) + cc * sk * ubuf[D + k];
          wave_sync();
          if (!(kiter < max_iter)) active = false;
          else new_direction = true;
        }
      }
    }
    if (new_direction) {
      vec[c.l] = gk;
      wave_sync();
      double hp = 0.0;
#pragma unroll
      for (int k = 0; k < D; ++k) hp += H[k] * vec[k];
      wave_sync();
      pk = -hp;
      const double derphi0 = gsum<G>(gk * pk);
      ls.start(fk, old_old, derphi0, &stp);
      phase = 1;
    }
  }
  if (status == 0) {
    const double gn = gmax<G>(fabs(gk));
    const double xn = gmax<G>(fabs(xk));
    if (kiter >= max_iter) status = 3;
    else if (gn != gn || fk != fk || xn != xn) status = 4;
  }

  // ---- result: L L^dagger / Tr  (state.py:214-215)
  double tr;
  const cd m = S::build_llh(c, xk, tr);
  if (live) {
    double* out = rho + ((size_t)b * D + c.l) * 2;
    out[0] = m.re / tr;
    out[1] = m.im / tr;
    if (c.l == 0) {
      if (nit_out) nit_out[b] = kiter;
      if (nfev_out) nfev_out[b] = nfev;
      if (fun_out) fun_out[b] = fk;
      if (status_out) status_out[b] = status;
    }
  }
}

}  // namespace qt
