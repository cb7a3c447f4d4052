// Workgroup Jacobi eigenvalue clip  U max(lambda, eps) U^dagger  of a d x d Hermitian matrix held one element
// per thread (d = 16, 32: 256 / 1024 threads, thread t = i * d + j).  Shared by the n = 4, 5 state kernels
// (qt_large.h: a7, state.py:267-273) and the CP step of the n = 2 process kernels (qt_process.h: a15,
// process.py:267-278).  Profiling stamps (QT_STAMP*) only in the profile build and only when STAMPS is set.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_small.h"

namespace qt {

template <int d, int NT, bool STAMPS>
struct JacobiWG {
  static_assert(NT == d * d && NT % 64 == 0 && (d & (d - 1)) == 0, "one thread per matrix element, d a power of two");
  static constexpr int LDV = d + 1;  // pitch of the eigenvector image (conflict-free column reads in the rebuild)
  static constexpr int NW = NT / 64;
  // LDS regions, as offsets in doubles from the (16-byte aligned) base `sm`; all even:
  //   img0, img1: d * d complex each; rot: 6 d doubles; v: d * LDV complex; lam: d; red: 32
  struct Lds {
    int img0, img1, rot, v, lam, red;
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

  // One round = d/2 disjoint rotations; element (i, j) needs the rotation of its column pair (j, j ^ r)
  // and of its row pair (i, i ^ r).  The d rotations of a round hang on 3 d/2 pivot elements only, so they
  // are computed ONE ROUND AHEAD: while everybody applies round r, the first 2 d threads bring the
  // pivots of round r + 1 up to date themselves (thread k < d the diagonal element (k, k), thread d + k
  // the off-diagonal one of k's next pair; same arithmetic as the owners of those elements), turn them
  // into cos / w and leave them in the other of two small LDS tables; the barrier of round r + 1 publishes
  // them.  One barrier per round and nobody waits for a rotation: v9 computed them between two barriers
  // with all other wavefronts idle (~2500 clk per round at d = 32); each wavefront computing all d
  // rotations for itself was tried and is bound by the 4 x 16 copies of that arithmetic instead (slower).
  // The A image alternates between two buffers (img0 / img1); V never goes to LDS during the sweeps:
  // V[i][j ^ r] is in lane l ^ r.
  __device__ __forceinline__ static cd rotate_elem(cd a, cd a_c, cd a_r, cd a_x, double cj, cd wj, double ci, cd wi) {
    // A'_ij = ci (a_ij cj + a_i,pj wj) + conj(wi) (a_pi,j cj + a_pi,pj wj)
    const cd t0 = cadd(cscale(a, cj), cmul(a_c, wj));
    const cd t1 = cadd(cscale(a_r, cj), cmul(a_x, wj));
    return cadd(cscale(t0, ci), cmulc(t1, wi));
  }
  // rotation of index k in the round that pairs k with pk, already signed for k's side:
  // J[pk][k] = -conj(w) if k is the lower index, else w
  __device__ __forceinline__ static void signed_rotation(int k, int pk, double a_kk, double a_pkpk, cd a_pq, double* rc, cd* rw) {
    double cs;
    cd w;
    rotation(k < pk ? a_kk : a_pkpk, k < pk ? a_pkpk : a_kk, a_pq, cs, w);
    if (k < pk) w = cd{-w.re, w.im};
    rc[k] = cs;
    rw[k] = w;
  }
  // (Tried on top: wavefront 0 doing nothing but the look-ahead, its rows carried by wavefront 1 as a second
  //  element per lane -- 5 % slower at n = 5; the rounds are bound by the LDS queue right after each barrier,
  //  scripts/jacobi_wave_timing.py, not by wavefront 0.)
  // warm: the eigenvector image still holds the (unitary) V of an earlier call on a nearby matrix -- the sweeps then
  // start from V^dagger A V, which is already close to diagonal, and continue to accumulate into that V.
  __device__ static cd clip(const int t, cd a, const double eps, double* sm, const Lds o, const bool normalise,
                            const bool warm = false) {
    const int i = t / d, j = t % d;
    double* red = sm + o.red;
    // (buffers are picked by OFFSET from the LDS base: with an array of two pointers indexed at run time the
    //  compiler loses the address space and emits flat_load / flat_store for every access in the loop)
    cd* img1 = reinterpret_cast<cd*>(sm + o.img1);
    cd* Vi = reinterpret_cast<cd*>(sm + o.v);
    cd v{i == j ? 1.0 : 0.0, 0.0};
    if (i == j) a.im = 0.0;
    const double nrm = wsum(red, a.re * a.re + a.im * a.im);
    // (the rounds use the unpadded pitch d: their pivot reads A[p][q] over lanes j spread over the
    //  banks through q = j ^ r, and padding the rows makes them collide instead)
    img1[t] = a;
    __syncthreads();
    if (warm) {  // uniform
      cd* img0 = reinterpret_cast<cd*>(sm + o.img0);
      cd b{0.0, 0.0};  // (A V)_ij
#pragma unroll 4
      for (int k = 0; k < d; ++k) b = cadd(b, cmul(img1[i * d + k], Vi[k * LDV + j]));
      img0[t] = b;
      v = Vi[i * LDV + j];
      __syncthreads();
      cd w{0.0, 0.0};  // (V^dagger A V)_ij = sum_k conj(V_ki) (A V)_kj
#pragma unroll 4
      for (int k = 0; k < d; ++k) {
        const cd vk = Vi[k * LDV + i], bk = img0[k * d + j];
        w.re = fma(vk.re, bk.re, fma(vk.im, bk.im, w.re));
        w.im = fma(vk.re, bk.im, fma(-vk.im, bk.re, w.im));
      }
      a = w;
      if (i == j) a.im = 0.0;
      __syncthreads();  // every read of the old img1 / V image is done
      img1[t] = a;
      __syncthreads();
    }
    if (t < d) {  // rotations of the very first round, from the input itself
      const int k = t, pk = k ^ 1, p = k < pk ? k : pk, q = k < pk ? pk : k;
      signed_rotation(k, pk, img1[k * d + k].re, img1[pk * d + pk].re, img1[p * d + q], sm + o.rot,
                      reinterpret_cast<cd*>(sm + o.rot + d));
    }
    int which = 0, cur = 0;
#ifdef QT_PHASE_TIMING  // per wavefront: clocks spent working / waiting at the round barrier (slots 19 / 31)
    long long t_work = 0, t_wait = 0, t_last = (long long)__builtin_readcyclecounter();
#endif
    for (int sweep = 0; sweep < 30; ++sweep) {
      const double off = wsum(red, i != j ? a.re * a.re + a.im * a.im : 0.0);
      if (STAMPS) QT_STAMP_VAL(20 + (sweep < 11 ? sweep : 11), (long long)(off / nrm * 1e30));
      // off-diagonal norm <= 1e-13 of the matrix norm: by the Lipschitz bound of the clip the result moves by
      // no more than that (~2e-14 here).  At d = 16, 32 most trials arrive at 1e-27 .. 1e-28 after their last
      // useful sweep; a 1e-28 threshold sent them through one more (measured: profile build, slot 20+).
      if (!(off > 1e-26 * nrm)) break;  // uniform: every thread holds the same sums
      for (int r = 1; r < d; ++r) {
        const int rn = r + 1 < d ? r + 1 : 1;  // the round after this one (round 1 of the next sweep)
        cd* Ai = reinterpret_cast<cd*>(sm + (which ? o.img1 : o.img0));
        const double* rc = sm + o.rot + (cur ? 3 * d : 0);  // cos [d], w [d] complex, twice
        const cd* rw = reinterpret_cast<const cd*>(rc + d);
        double* nc = sm + o.rot + (cur ? 0 : 3 * d);
        cd* nw = reinterpret_cast<cd*>(nc + d);
        which ^= 1;
        cur ^= 1;
        Ai[t] = a;
#ifdef QT_PHASE_TIMING
        const long long t_b0 = (long long)__builtin_readcyclecounter();
#endif
        __syncthreads();
#ifdef QT_PHASE_TIMING
        {
          const long long t_b1 = (long long)__builtin_readcyclecounter();
          t_work += t_b0 - t_last;
          t_wait += t_b1 - t_b0;
          t_last = t_b1;
        }
#endif
        // partner elements (i, j ^ r), (i ^ r, j), (i ^ r, j ^ r): with t = i * d + j and d a power of two these
        // are t ^ r, t ^ (r d), t ^ (r d + r) -- one XOR each instead of rebuilding the index from (i, j)
        const cd a_c = Ai[t ^ r], a_r = Ai[t ^ (r * d)], a_x = Ai[t ^ (r * d + r)];
        const cd v_c{__shfl_xor(v.re, r, 64), __shfl_xor(v.im, r, 64)};
        const double cj = rc[j], ci = rc[i];
        const cd wj = rw[j], wi = rw[i];
        if (t < 2 * d) {  // i = 0: diagonal pivots, i = 1: off-diagonal pivots; k = j
          const int k = j, pk = k ^ rn, p = k < pk ? k : pk, q = k < pk ? pk : k;
          const int x = i ? p : k, y = i ? q : k, px = x ^ r, py = y ^ r;
          const cd e = rotate_elem(Ai[x * d + y], Ai[x * d + py], Ai[px * d + y], Ai[px * d + py], rc[y], rw[y], rc[x],
                                   rw[x]);
          const double d_pk = __shfl_xor(e.re, rn, 64);                     // A'[pk][pk] from lane pk
          const cd a_pq{__shfl_down(e.re, d, 64), __shfl_down(e.im, d, 64)};  // A'[p][q] from lane d + k
          if (i == 0) signed_rotation(k, pk, e.re, d_pk, a_pq, nc, nw);
        }
        a = rotate_elem(a, a_c, a_r, a_x, cj, wj, ci, wi);
        v = cadd(cscale(v, cj), cmul(v_c, wj));
        if (i == j) a.im = 0.0;
      }
    }
    if (STAMPS) {
      QT_STAMP_VAL(19, t_work);
      QT_STAMP_VAL(31, t_wait);
    }
    double* lam = sm + o.lam;
    Vi[i * LDV + j] = v;
    if (i == j) lam[i] = a.re;
    __syncthreads();
    cd rr{0.0, 0.0};
#pragma unroll 4  // fully unrolled, hipcc keeps all 2 d operands live for the imaginary part and spills them
    for (int k = 0; k < d; ++k) {
      const double lc = lam[k] > eps ? lam[k] : eps;
      const cd p = cmulc(Vi[i * LDV + k], Vi[j * LDV + k]);
      rr.re += lc * p.re;
      rr.im += lc * p.im;
    }
    if (!normalise) return rr;
    const double tr = wsum(red, i == j ? rr.re : 0.0);
    return cd{rr.re / tr, rr.im / tr};
  }
};

}  // namespace qt
