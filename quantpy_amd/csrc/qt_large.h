// Workgroup-per-trial kernels for n = 4, 5 qubits (D = 4^n = 256 / 1024 = one workgroup).
//
// The mapping of qt_small.h one level up: thread t of the workgroup is Bloch / parameter index
// k = t AND matrix element (i, j) = (t / d, t % d) of every d x d complex matrix (d = 16 / 32), so the
// same formulas apply with __syncthreads() in place of the wave fence and a two-level (DPP + LDS)
// reduction in place of the DPP butterfly.  What changes is where things live:
//   * LDS (<= 160 KB): L, the broadcast vector, and two overlaid work regions X / Y that hold the
//     stages of the factorised POVM contraction (R-order, see qt_small.h) and, outside of it, the
//     Jacobi images A / V and the gradient matrix.  At n = 5 with 'proj-set' (M = 6^5 = 7776):
//     X = 7776, Y = 5184 doubles -> 127 KB.  Frequencies are not stored: they are re-read from the
//     counts (int64, gathered through the R-order row map) when needed.
//   * The BFGS inverse Hessian is never formed.  A dense D x D f64 H (8 MB per trial at n = 5) had to be
//     swept three times per iteration and made the loop HBM-bound (measured: 1.3 ms per iteration at
//     256 concurrent trials = 4.6 TB/s).  Instead the (s_i, y_i) pairs of all iterations so far are kept
//     (2 D doubles per iteration, in a caller-provided workspace; thread t only ever touches element t
//     of each pair, so they are private, coalesced streams) and H_k g is evaluated by the two-loop
//     recursion with H_0 = I -- algebraically the same matrix as scipy's update
//     H <- (I - rho s y^T) H (I - rho y s^T) + rho s s^T applied k times, at 4 k D flops and 2 k
//     workgroup reductions instead of 24 MB of traffic.
// Product POVMs (every built-in one) take the factorised contractions; a plain (S, K, 4^n) tensor takes the dense
// operand path (col_dot_dense / row_dot_dense below): 2 M D doubles streamed per evaluation and trial -- fine at
// n = 4 (5.3 MB, mostly L2 hits since every trial reads the same operand), slow but correct at n = 5 (128 MB).
//
// Reference semantics: identical to qt_small.h (state.py:191-229, 267-273; routines.py:84-101).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_jacobi_wg.h"
#include "qt_linesearch.h"
#include "qt_signclip_wg.h"
#include "qt_small.h"

namespace qt {

template <int NQ>
struct Large {
  static constexpr int d = 1 << NQ;
  static constexpr int D = d * d;
  static constexpr int NT = D;          // threads per workgroup = one trial
  static constexpr int NW = NT / 64;
  static constexpr int T = d * (d - 1) / 2;
  // Row pitch of the d x d complex images: d + 1.  With pitch d (256 / 512 bytes) the d rows that one
  // column is read from all start on the same LDS bank -- a 16- / 32-way conflict on every
  // A[j][k]-over-j read (Cholesky, L L^dagger, eigenvector products); d + 1 leaves a 2-way one.
  static constexpr int LD = d + 1;
  static constexpr int MAT = 2 * LD * d;       // doubles per matrix image
  // LDS layout in doubles
  static constexpr int oL = 0;                 // complex [d][LD]
  static constexpr int oVec = oL + MAT;        // [D]
  static constexpr int oLam = oVec + D;        // [d]
  static constexpr int oRed = oLam + d;        // [32] reduction scratch
  static constexpr bool kJacobiClip = false;   // a7 through the cyclic Jacobi eigensolver instead of the sign-function clip
  static constexpr int oRot = oRed + 32;       // Jacobi rotations of this round and the next: 2 x (cos [d], w [d] complex)
  static constexpr int oLs = oRot + (kJacobiClip ? 6 * d : 0);  // line-search state parked across an evaluation
  static constexpr int oTab = oLs + LineSearch::SLOTS;  // tabT [R1][4], tabP [R1][4]
  __host__ __device__ static int x_doubles(int M) { return (M > MAT ? M : MAT) + (M & 1); }
  __host__ __device__ static int y_doubles(int M, int R1) {
    int y = R1 > 0 ? (M / R1) * 4 : 0;  // R1^(n-1) * 4: the largest stage that lands in Y (dense POVM: none)
    if (y < MAT) y = MAT;
    return y + (y & 1);
  }
  __host__ __device__ static size_t lds_bytes(int M, int R1, int max_iter = 0) {
    return (size_t)(oTab + 8 * R1 + x_doubles(M) + y_doubles(M, R1) + 2 * max_iter) * sizeof(double);
  }

  struct Ctx {
    int t, i, j, e;  // thread, matrix element, its slot i * LD + j in a matrix image
    double* sm;
    int M;
    ProductView pr;
    const double *Aw, *AwT, *PinvT;  // dense operands ([M][D], [D][M], [M][D]); read only when the POVM is not a product
    const int64_t* counts;  // this trial's counts [M]
    const uint32_t* cnt;    // LDS copy of the counts in R-order, or null (frequencies are re-read three times per MLE)
    double tot;             // sum of counts
    bool shots_ok;          // per-setting totals proportional to the registered shots (PovmView::Ns)
    int xm, zm, ny;
    int pi, pj, pkind;
    __device__ __forceinline__ cd* L() const { return reinterpret_cast<cd*>(sm + oL); }
    __device__ __forceinline__ double* vec() const { return sm + oVec; }
    __device__ __forceinline__ double* lam() const { return sm + oLam; }
    __device__ __forceinline__ double* red() const { return sm + oRed; }
    __device__ __forceinline__ double* lsbuf() const { return sm + oLs; }
    __device__ __forceinline__ double* tabT() const { return sm + oTab; }
    __device__ __forceinline__ double* tabP() const { return sm + oTab + 4 * pr.R1; }
    __device__ __forceinline__ double* X() const { return sm + oTab + 8 * pr.R1; }
    __device__ __forceinline__ double* Y() const { return X() + x_doubles(M); }
    __device__ __forceinline__ double* pair_rho() const { return Y() + y_doubles(M, pr.R1); }  // [max_iter]
    __device__ __forceinline__ cd* Aimg() const { return reinterpret_cast<cd*>(Y()); }  // overlays Y
    __device__ __forceinline__ cd* Vimg() const { return reinterpret_cast<cd*>(X()); }  // overlays X
  };

  // ---- workgroup reductions: identical bits in every thread ----------------------------------
  __device__ static double bsum(const Ctx& c, double v) {
    v = gsum<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) c.red()[threadIdx.x >> 6] = v;
    __syncthreads();
    if constexpr (NW == 16) {
      // Round 3: ONE LDS read per wavefront -- lane l takes partial l % 16 (four lanes per address: a broadcast) and the 16
      // partials are summed on the DPP network; every row of 16 lanes runs the same tree on the same numbers, so every
      // thread still ends with the same bits.  Sixteen reads per thread were 256 wave-wide LDS instructions per reduction
      // and workgroup: ~1000 clocks of LDS issue, a third of a round of the BFGS two-loop recursion at n = 5
      // (scripts/phase_timing_large_bfgs.py).
      return gsum<16>(c.red()[threadIdx.x & 15]);
    } else {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += c.red()[w];
      return s;
    }
  }
  // The same with ONE barrier, for a loop of reductions with nothing else in the red() scratch: round k uses half k % 2 of
  // it, so a wavefront that is two rounds ahead is the first that could overwrite what a slow one still reads -- and it
  // cannot be: it has passed the barrier of the round in between, which the slow one reaches only after its reads.  The
  // caller puts a barrier before the first round (the standard reductions before it read red()[0..15]) -- one after the
  // last is not needed: the next standard reduction starts with one.
  __device__ static double bsum_alt(const Ctx& c, double v, int half) {
    static_assert(NW <= 16, "two halves of the 32-entry scratch");
    v = gsum<64>(v);
    double* r = c.red() + 16 * half;
    if ((threadIdx.x & 63) == 0) r[threadIdx.x >> 6] = v;
    __syncthreads();
    if constexpr (NW == 16) {
      return gsum<16>(r[threadIdx.x & 15]);
    } else {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < NW; ++w) s += r[w];
      return s;
    }
  }
  __device__ static double bmax(const Ctx& c, double v) {
    v = gmax<64>(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) c.red()[threadIdx.x >> 6] = v;
    __syncthreads();
    if constexpr (NW == 16) {
      return gmax<16>(c.red()[threadIdx.x & 15]);
    } else {
      double s = c.red()[0];
#pragma unroll
      for (int w = 1; w < NW; ++w) s = nanmax(s, c.red()[w]);
      return s;
    }
  }

  // Result of the workgroup's trial, element (i, j) = r per thread: store it and / or its Hilbert-Schmidt distance to
  // o.centre (geometry.py:16-20; Tr(Delta Delta) = sum_ij Delta_ij Delta_ji, the transposed element through the L image,
  // which no estimator reads after its last build_llh).  Called by every thread of the workgroup.
  __device__ static void emit(const Ctx& c, const EstOut& o, int b, cd r) {
    if (o.dist) {  // uniform over the launch
      const double2 cc = *reinterpret_cast<const double2*>(o.centre + 2 * c.t);
      const cd dl{r.re - cc.x, r.im - cc.y};
      cd* L = c.L();
      __syncthreads();
      L[c.e] = dl;
      __syncthreads();
      const cd dt = L[c.j * LD + c.i];
      const cd tm = hs_term(dl, dt);
      const double sr = bsum(c, tm.re);
      const double si = bsum(c, tm.im);
      const double v = sqrt(hypot(sr, si)) / sqrt(2.0);
      if (c.t == 0) o.dist[b] = v < 1e-15 ? 0.0 : v;
    }
    if (o.rho) {
      double* out = o.rho + ((size_t)b * D + c.t) * 2;
      out[0] = r.re;
      out[1] = r.im;
    }
  }

  __device__ static void make_ctx(Ctx& c, double* smem, const PovmView& pv, const int64_t* counts) {
    c.t = threadIdx.x;
    c.i = c.t / d;
    c.j = c.t % d;
    c.e = c.i * LD + c.j;
    c.sm = smem;
    c.M = pv.M;
    c.pr = pv.pr;
    c.Aw = pv.Aw;
    c.AwT = pv.AwT;
    c.PinvT = pv.PinvT;
    c.counts = counts;
    int xm = 0, zm = 0, ny = 0;
#pragma unroll
    for (int b = 0; b < NQ; ++b) {
      const int dig = (c.t >> (2 * b)) & 3;
      if (dig == 1 || dig == 2) xm |= 1 << b;
      if (dig == 2 || dig == 3) zm |= 1 << b;
      if (dig == 2) ++ny;
    }
    c.xm = xm;
    c.zm = zm;
    c.ny = ny & 3;
    if (c.t < d) {
      c.pi = c.pj = c.t;
      c.pkind = 0;
    } else {
      int tt = c.t - d;
      c.pkind = 1;
      if (tt >= T) {
        tt -= T;
        c.pkind = 2;
      }
      int ii = (int)((1.0 + sqrt(1.0 + 8.0 * tt)) * 0.5);  // row of np.tril_indices(d, -1)[tt]
      while ((ii * (ii - 1)) / 2 > tt) --ii;
      while ((ii * (ii + 1)) / 2 <= tt) ++ii;
      c.pi = ii;
      c.pj = tt - (ii * (ii - 1)) / 2;
    }
    for (int e = c.t; e < 4 * c.pr.R1; e += NT) {
      c.tabT()[e] = c.pr.T[e];
      c.tabP()[e] = c.pr.P1T[e];
    }
    double part = 0.0;
    bool wide = false;
    uint32_t* cache = nullptr;
    // pv.extra > 0: offset (in doubles) of a 4 M-byte block behind everything else in the LDS allocation, sized by the host
    // when it fits (QT_LAUNCH_LARGE_X)
    if (pv.extra > 0 && counts) cache = reinterpret_cast<uint32_t*>(smem + pv.extra);
    c.cnt = cache;
    // Per-setting totals for the shots check, from the values of the load pass itself: the K outcomes of a setting are K
    // neighbouring rows, i.e. K neighbouring lanes of one of the eight values a thread holds, and a DPP butterfly sums
    // them (row totals, plus the neighbouring row for K = 32).  The check used to re-read the counts afterwards, thread s
    // walking the K values of setting s one dependent L2 round trip at a time: ~12 k of the 22 k clocks this function
    // took at n = 5.  Needs K = 16 or 32 and the whole trial in one pass; otherwise the re-read below.
    const bool seg = counts && pv.Ns && (pv.K == 16 || pv.K == 32) && c.M <= 8 * NT;  // uniform
    double segsum[8];
    if (counts) {
      // eight rows per thread and pass, every load of a pass requested before the first is used (as a plain loop each
      // iteration waited out its own HBM round trip: the counts are the only cold read of a trial, 62 KB at n = 5)
      for (int m0 = c.t; m0 < c.M; m0 += 8 * NT) {
        int64_t v[8];
        int ri[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int m = m0 + u * NT;
          const int mm = m < c.M ? m : c.M - 1;
          v[u] = counts[mm];
          ri[u] = cache ? c.pr.rinv[mm] : 0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool valid = m0 + u * NT < c.M;
          const double dv = valid ? (double)v[u] : 0.0;
          part += dv;
          if (valid && cache) {  // the natural-order count lands at its R-order slot
            cache[ri[u]] = (uint32_t)v[u];
            wide |= (v[u] >> 32) != 0;
          }
          if (seg) {  // (every lane takes part: rows past M count as zero)
            double sv = dv;
            sv += dpp_f64<0xB1>(sv);
            sv += dpp_f64<0x4E>(sv);
            sv += dpp_f64<0x141>(sv);
            sv += dpp_f64<0x140>(sv);                          // every lane: the total of its row of 16
            if (pv.K == 32) sv += dpp_f64<0x142, 0xA>(sv);     // rows 1, 3: + the row before (row_bcast15)
            segsum[u] = sv;
          }
        }
      }
    }
    c.tot = bsum(c, part);  // (barriers inside also publish the tables and the count cache)
    // shots check (state.py:138-141, 194-197)
    double bad = 0.0;
    if (seg) {
      const int lane = c.t & 63;
      const bool holder = pv.K == 32 ? (lane & 31) == 31 : (lane & 15) == 15;  // last lane of a setting has its total
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int m = c.t + u * NT;
        if (holder && m < c.M && !shots_match(segsum[u], c.tot, pv.Ns[m / pv.K], pv.ns_tot)) bad = 1.0;
      }
    } else if (counts && pv.Ns) {  // thread s sums the K outcomes of setting s (L2 hits: just read)
      for (int s = c.t; s < pv.S; s += NT) {
        double t = 0.0;
        for (int k = 0; k < pv.K; ++k) t += (double)counts[s * pv.K + k];
        if (!shots_match(t, c.tot, pv.Ns[s], pv.ns_tot)) bad = 1.0;
      }
    }
    c.shots_ok = !(counts && pv.Ns) || !(bmax(c, bad) > 0.0);
    if (cache && bmax(c, wide ? 1.0 : 0.0) > 0.0) c.shots_ok = false;  // a count beyond 32 bits: not this POVM's data
  }
  // frequency of R-order row o (state.py:193, :227)
  __device__ __forceinline__ static double freq(const Ctx& c, int o) {
    return (c.cnt ? (double)c.cnt[o] : (double)c.counts[c.pr.rmap[o]]) / c.tot;
  }

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
  // Bloch vector <-> matrix as Walsh-Hadamard transforms in registers.  For a fixed x-mask the d Pauli strings
  // (x, z), z < d, have P[r][r ^ x] = (-i)^popc(x & z) (-1)^popc(r & z): the sum over r (bloch_of) or over z (matrix_of)
  // is a length-d WHT of one (shifted) diagonal of the matrix.  Thread t = (x, lane) holds one entry of it; log2(d)
  // butterfly stages across the d lanes of its group (quad_perm and row_ror DPP moves, ds_bpermute for the strides 4
  // and 16) replace the d-term sums of rounds 1-2, which read d 16-byte LDS entries per thread -- 512 KB through the
  // LDS per call at n = 5, ~9 k + ~6 k of the ~75 k clocks of an evaluation (profiles/round2_final_phase_timing_n5.txt).
  // The result lands in the thread that owns (x, lane), not in the thread that needs it: one scatter through LDS.
  __device__ __forceinline__ static cd wht_butterfly(cd s, cd p, bool upper) {
    return upper ? cd{p.re - s.re, p.im - s.im} : cd{s.re + p.re, s.im + p.im};
  }
  __device__ __forceinline__ static cd wht_lanes(cd s) {
    const int lane = threadIdx.x & 63;
    s = wht_butterfly(s, cd{dpp_f64<0xB1>(s.re), dpp_f64<0xB1>(s.im)}, lane & 1);  // quad_perm [1,0,3,2]
    s = wht_butterfly(s, cd{dpp_f64<0x4E>(s.re), dpp_f64<0x4E>(s.im)}, lane & 2);  // quad_perm [2,3,0,1]
    s = wht_butterfly(s, cd{__shfl_xor(s.re, 4), __shfl_xor(s.im, 4)}, lane & 4);
    s = wht_butterfly(s, cd{dpp_f64<0x128>(s.re), dpp_f64<0x128>(s.im)}, lane & 8);  // row_ror:8 = lane ^ 8 within a row
    if constexpr (d == 32) s = wht_butterfly(s, cd{__shfl_xor(s.re, 16), __shfl_xor(s.im, 16)}, lane & 16);
    return s;
  }
  // Bloch component k = c.t of the matrix image m; uses c.vec() as the exchange buffer (ends with a barrier)
  __device__ static double bloch_of(const Ctx& c, const cd* m) {
    static_assert(d == 16 || d == 32, "one WHT group = d lanes of a wavefront");
    const int x = c.i, r = c.j;  // this thread's entry of diagonal x: M[r][r ^ x]
    const cd e = m[r * LD + (r ^ x)];
    const cd s = wht_lanes(cd{e.re, -e.im});  // lane index z: sum_r (-1)^popc(r & z) conj(M[r][r ^ x])
    const int z = r, ny = __popc(x & z) & 3;
    const double v = ny == 0 ? s.re : ny == 1 ? s.im : ny == 2 ? -s.re : -s.im;
    double* vec = c.vec();
    vec[pauli_index(x, z)] = v / d;
    __syncthreads();
    return vec[c.t];
  }
  // spread(b): bit q of b moved to bit 2q.  pauli_index(x, z) has digit bits (hi, lo) = (z_q, x_q ^ z_q), i.e.
  // index = spread(x) ^ 3 * spread(z).
  __host__ __device__ static constexpr int spread(int b) {
    int r = 0;
    for (int q = 0; q < NQ; ++q) r |= ((b >> q) & 1) << (2 * q);
    return r;
  }
  // Element (i, j) = c.e of sum_k v[k] P_k; `img` = a free d x LD image for the exchange (ends with a barrier)
  __device__ static cd matrix_of(const Ctx& c, const double* v, cd* img) {
    const int x = c.i, z = c.j;  // this thread's term of diagonal x
    const double val = v[spread(x) ^ (3 * spread(z))];
    const int ny = __popc(x & z);              // phase (-i)^ny: 1, -i, -1, i
    const double sv = ((ny ^ (ny >> 1)) & 1) ? -val : val;
    const cd s = wht_lanes((ny & 1) ? cd{0.0, sv} : cd{sv, 0.0});  // lane index i: M[i][i ^ x]
    const int i = z;
    img[i * LD + (i ^ x)] = s;
    __syncthreads();
    return img[c.e];
  }

  // ---- factorised contractions (qt_small.h for the scheme) -------------------------------------
  __device__ __forceinline__ static int ipow(int b, int e) {
    int r = 1;
    for (int q = 0; q < e; ++q) r *= b;
    return r;
  }
  template <bool FWD>
  __device__ __forceinline__ static double stage_value(const double* tb, int R1, int ent, int stride, const double* in) {
    const int base = ent & 0xffff, sel = ent >> 16;
    if (FWD) {
      const double* row = tb + sel * 4;
      return fma(row[3], in[base + 3 * stride],
                 fma(row[2], in[base + 2 * stride], fma(row[1], in[base + stride], row[0] * in[base])));
    }
    double acc = 0.0;
    if (R1 == 6) {  // table height known: all 12 reads issued together (a run-time trip count serialises them)
      double tv[6], iv[6];
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        tv[r] = tb[r * 4 + sel];
        iv[r] = in[base + r * stride];
      }
#pragma unroll
      for (int r = 0; r < 6; ++r) acc = fma(tv[r], iv[r], acc);
      return acc;
    }
    if (R1 == 4) {
      double tv[4], iv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        tv[r] = tb[r * 4 + sel];
        iv[r] = in[base + r * stride];
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = fma(tv[r], iv[r], acc);
      return acc;
    }
    for (int r = 0; r < R1; ++r) acc = fma(tb[r * 4 + sel], in[base + r * stride], acc);
    return acc;
  }
  // (base | sel << 16) of output o of a stage whose untouched Pauli indices span 4^(n-q) = 1 << lk entries
  // (the closed form of the host tables of qt_set_povm_product; at n = 4, 5 those are 135 KB -- too
  // big for LDS -- and a dependent L2 load per output costs more than these few integer operations):
  //   forward : o = [r_pre][r_q][k_rest] -> base = r_pre * 4 * Kq + k_rest, sel = r_q
  //   backward: o = [r_pre][k_q][k_rest] -> base = r_pre * R1 * Kq + k_rest, sel = k_q
  template <bool FWD>
  __device__ __forceinline__ static int stage_entry(int R1, int lk, int o) {
    const int krest = o & ((1 << lk) - 1);
    if (FWD) {
      const int t = o >> lk;
      int rpre, rq;
      if (R1 == 6) {
        rpre = t / 6;
        rq = t - 6 * rpre;
      } else if (R1 == 4) {
        rpre = t >> 2;
        rq = t & 3;
      } else {
        rpre = t / R1;
        rq = t - R1 * rpre;
      }
      return ((rpre << (lk + 2)) + krest) | (rq << 16);
    }
    const int rpre = o >> (lk + 2), kq = (o >> lk) & 3;
    return ((rpre * R1 << lk) + krest) | (kq << 16);
  }
  template <bool FWD>
  __device__ static void stage(const Ctx& c, const double* tb, int lk, int n_out, const double* in, double* out) {
    const int R1 = c.pr.R1;
    for (int o = c.t; o < n_out; o += NT) out[o] = stage_value<FWD>(tb, R1, stage_entry<FWD>(R1, lk, o), 1 << lk, in);
    __syncthreads();
  }
  // Backward pass from Y_n held in X (R-order) to this thread's Y_0[k = t].  Stage n: X -> Y, then
  // alternating; tb = tabT (A^T y) or tabP (A^+ f).
  __device__ static double prod_backward(const Ctx& c, const double* tb) {
    const int R1 = c.pr.R1;
    // (X / Y picked by offset from the LDS base, not through an array of pointers: see psd_project)
    const int xo = (int)(c.X() - c.sm), yo = (int)(c.Y() - c.sm);
    int in_off = xo, which = 0;
    for (int q = NQ; q >= 2; --q) {
      const int lk = 2 * (NQ - q);
      const int n_out = ipow(R1, q - 1) * 4 << lk;
      const int out_off = which ? xo : yo;
      stage<false>(c, tb, lk, n_out, c.sm + in_off, c.sm + out_off);
      in_off = out_off;
      which ^= 1;
    }
    const double* in = c.sm + in_off;
    const double r = stage_value<false>(tb, R1, stage_entry<false>(R1, 2 * (NQ - 1), c.t), 1 << (2 * (NQ - 1)), in);
    __syncthreads();
    return r;
  }

  // ---- dense operand path (a plain (S, K, 4^n) tensor, or a product POVM with unequal shots in 'lin'): the operand
  // streams from L2 / HBM with the lanes along its contiguous axis (512-byte segments per wavefront), the vector is
  // broadcast from LDS.  2 M D doubles per evaluation and trial: 5.3 MB at n = 4 -- a fallback, not the fast path.
  // sum_m op[m * D + t] * v[m]   (thread t = column; op row-major [M][D])
  __device__ __forceinline__ static double col_dot_dense(const Ctx& c, const double* __restrict__ op, const double* v) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const double* p = op + c.t;
    int m = 0;
#pragma unroll 2
    for (; m + 4 <= c.M; m += 4) {
      a0 = fma(p[(size_t)(m + 0) * D], v[m + 0], a0);
      a1 = fma(p[(size_t)(m + 1) * D], v[m + 1], a1);
      a2 = fma(p[(size_t)(m + 2) * D], v[m + 2], a2);
      a3 = fma(p[(size_t)(m + 3) * D], v[m + 3], a3);
    }
    for (; m < c.M; ++m) a0 = fma(p[(size_t)m * D], v[m], a0);
    return (a0 + a1) + (a2 + a3);
  }
  // sum_k opT[k * M + row] * v[k]   (opT = [D][M]: consecutive rows are consecutive addresses)
  __device__ __forceinline__ static double row_dot_dense(const Ctx& c, const double* __restrict__ opT, int row, const double* v) {
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    const double* p = opT + row;
#pragma unroll 2
    for (int k = 0; k < D; k += 4) {
      a0 = fma(p[(size_t)(k + 0) * c.M], v[k + 0], a0);
      a1 = fma(p[(size_t)(k + 1) * c.M], v[k + 1], a1);
      a2 = fma(p[(size_t)(k + 2) * c.M], v[k + 2], a2);
      a3 = fma(p[(size_t)(k + 3) * c.M], v[k + 3], a3);
    }
    return (a0 + a1) + (a2 + a3);
  }

  // ---- a6: linear inversion.  Product POVM with equal shots: pinv(T)^(x n) / w; otherwise the dense left inverse.
  __device__ static cd lin_invert(const Ctx& c, double& bloch_t) {
    if (!(c.pr.enabled && c.pr.uniform)) {
      for (int m = c.t; m < c.M; m += NT) c.X()[m] = (double)c.counts[m] / c.tot;  // caller's (S, K) order
      __syncthreads();
      bloch_t = col_dot_dense(c, c.PinvT, c.X()) / d;  // bloch_k = sum_m Pinv[k][m] f_m / d
      c.vec()[c.t] = bloch_t;
      __syncthreads();
      const cd r = matrix_of(c, c.vec(), c.Aimg());
      __syncthreads();
      return r;
    }
    for (int o = c.t; o < c.M; o += NT) c.X()[o] = freq(c, o);
    __syncthreads();
    bloch_t = prod_backward(c, c.tabP()) / (c.pr.wuni * d);
    c.vec()[c.t] = bloch_t;
    __syncthreads();
    const cd r = matrix_of(c, c.vec(), c.Aimg());
    __syncthreads();
    return r;
  }

  // ---- a7: eigenvalue clip U max(v, eps) U^dagger / Tr.  Default: through the matrix sign function on the FP64 matrix
  // cores (qt_signclip_wg.h; three d x d images: the Y overlay, the L region -- which holds nothing live here, the
  // factorisation that called us has failed -- and the X overlay).  kJacobiClip selects the cyclic Jacobi
  // eigensolver of qt_jacobi_wg.h instead (the version before round 2; kept for cross-checks).
  __device__ static cd psd_project(const Ctx& c, cd a, double eps) {
    if constexpr (kJacobiClip) {
      using J = JacobiWG<d, NT, true>;
      static_assert(J::LDV == LD, "the eigenvector image uses the pitch of the other d x d images");
      const typename J::Lds o{(int)(c.Y() - c.sm), oL, oRot, (int)(c.X() - c.sm), oLam, oRed};
      return J::clip(c.t, a, eps, c.sm, o, true);
    } else {
      using SC = SignClipWG<d, NT>;
      static_assert(SC::P == LD, "the images use the pitch of the other d x d images");
      const typename SC::Lds o{(int)(c.Y() - c.sm), oL, (int)(c.X() - c.sm), oRed};
      return SC::clip(c.t, a, eps, c.sm, o, true);
    }
  }

  // ---- a8: Cholesky (image in the A overlay, factor in L) ------------------------------------------
  // ONE wavefront runs the whole sweep, the matrix in its registers: lane (i, g) holds row i, columns
  // [g CPL, (g + 1) CPL) (d = 32: 2 column groups x 16 columns, 64 VGPRs; d = 16: 4 x 4).  A step sends the pivot column
  // through a double-buffered d-entry LDS buffer (written by the column group that owns it), every lane reads its row's
  // multiplier and the CPL multipliers of its columns, and updates its registers -- the hand-offs stay inside the
  // wavefront (wave_sync: no hardware wait).  The workgroup-wide version of rounds 1-2 was one column per BARRIER:
  // ~1000 clocks per step at n = 5 for a chain of LDS round trip, 1/sqrt, LDS write and a 16-wave barrier, 29-32 k per
  // sweep (two columns per step, idling the waves above the pivot row: measured, no gain -- the barrier chain stayed).
  // Same arithmetic per element in the same order (a_ij -= (a_ik rs)(conj(a_jk) rs) for k = 0 .. min(i, j) - 1): same
  // bits.  The other wavefronts wait at the closing barrier.  A non-positive pivot clears `ok`; the (unrolled, branch-free) sweep runs to its end.
  template <int WAVES>
  __device__ static double cholesky_param_wave(const Ctx& c, cd a, int& ok) {
    // WAVES wavefronts hold the matrix in registers: lane (i, g) of the first 64 WAVES threads has row i, columns
    // [g CPL, (g + 1) CPL).  WAVES = 1: hand-offs inside the wavefront (wave_sync).  WAVES = 4 (n = 5: 4 columns per lane):
    // one workgroup barrier per step, which the other wavefronts only pass through -- against the one-element-per-thread
    // form this moves 32 + 256 x 5 LDS entries per step instead of 1024 x 4 (that form is LDS-bandwidth bound: 64 KB per
    // step), and against one wavefront alone it has a quarter of the multiply-adds per lane.
    constexpr int LANES = 64 * WAVES, CG = LANES / d, CPL = d / CG;  // column groups, columns per lane
    static_assert(CG * CPL == d && LANES <= NT, "the column groups tile the matrix");
    cd* A = c.Aimg();
    cd* L = c.L();
    A[c.e] = a;
    L[c.e] = cd{0.0, 0.0};
    double* flag = c.red() + 24;  // [WAVES] verdicts
    __syncthreads();
    const bool worker = __builtin_amdgcn_readfirstlane((int)threadIdx.x) < LANES;  // provably wave-uniform: scalar branches below
    const int i = threadIdx.x % d, g = (threadIdx.x / d) % CG, j0 = g * CPL;
    cd* col = c.Vimg();  // 2 x (d + 1) entries of the X overlay (scratch here)
    cd r[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) r[q] = worker ? A[i * LD + j0 + q] : cd{0.0, 0.0};
    bool pd = true;
    // (fully unrolled so that r[] is indexed by constants and stays in registers; branch-free: after a non-positive
    //  pivot the sweep runs on, on numbers nobody reads)
#pragma unroll
    for (int k = 0; k < d; ++k) {
      const int kg = k / CPL, kq = k % CPL;
      // the pivot from the register of lane (k, kg) -- it sits in the wavefront that owns column k, the only one whose
      // value of it is used; the owners publish the column already scaled: l_ik = a_ik / sqrt(a_kk)
      const double akk = readlane_f64(r[kq].re, (kg * d + k) & 63);
      const bool own = worker && g == kg;
      pd = pd && (!own || akk > 0.0);
      const double rs = fast_rsqrt(akk > 1e-300 ? akk : 1e-300);
      // Branch-free on purpose: with an `if (owner)` block per unrolled step the compiler's control-flow structurisation
      // stretched the live ranges until the d = 32 sweep spilled ~1300 registers (2.9 KB of scratch per lane) even with
      // 256 registers allowed; the owners are picked by ADDRESS instead -- everybody else writes to a dummy slot (entry d
      // of the column buffer, the padding column of its own row of L).
      cd* cb = col + (k & 1) * (d + 1);
      const cd lik_k = (i == k) ? cd{akk * rs, 0.0} : cd{r[kq].re * rs, r[kq].im * rs};
      // rows and columns up to the pivot take no update: their published multiplier is zero (x - 0 * y = x exactly
      // for finite y; after a non-positive pivot the numbers may be anything, and nobody reads them)
      if (worker) {  // (scalar branch: the other wavefronts only keep the barrier count)
        cb[own ? i : d] = (i > k) ? lik_k : cd{0.0, 0.0};
        L[(own && i >= k) ? i * LD + k : i * LD + d] = lik_k;
      }
      if constexpr (WAVES == 1) wave_sync();
      else __syncthreads();
      if (worker && k + 1 < d) {
        const cd lik = cb[i];
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
          const cd ljk = cb[j0 + q];
          r[q].re -= lik.re * ljk.re + lik.im * ljk.im;  // l_ik conj(l_jk), as cmulc has it
          r[q].im -= lik.im * ljk.re - lik.re * ljk.im;
        }
      }
    }
    // every worker wavefront reports the pivots of the columns it owned
    const bool all_pd = __all(pd);
    if (worker && (threadIdx.x & 63) == 0) flag[threadIdx.x >> 6] = all_pd ? 1.0 : 0.0;
    __syncthreads();
    bool okb = true;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) okb = okb && flag[w] != 0.0;
    ok = okb;
    const cd e = L[c.pi * LD + c.pj];
    const double x = c.pkind == 2 ? e.im : e.re;
    __syncthreads();
    return x;
  }

  // n = 4: one wavefront (4 columns per lane, no barrier).  n = 5: four wavefronts (4 columns per lane) and a barrier per
  // step -- measured against the one-element-per-thread sweep of rounds 1-2 (0.152 ms per 256 'mle' trials) and against one
  // wavefront with 16 columns per lane (0.153 ms, 32 B of scratch): 0.148 ms; the 16-wavefront barrier itself is most of
  // what a step still costs.
  __device__ static double cholesky_param(const Ctx& c, cd a, int& ok) {
    if constexpr (NQ == 4) return cholesky_param_wave<1>(c, a, ok);
    else return cholesky_param_wave<4>(c, a, ok);
  }

  __device__ static cd make_feasible(const Ctx& c, cd r, double* xl, int* ok_out) {
    int ok;
    double x = cholesky_param(c, r, ok);
    QT_STAMP(3);
    const double tr = bsum(c, c.i == c.j ? r.re : 0.0);
    cd out{r.re / tr, r.im / tr};
    x = x / sqrt(tr);
    if (!ok) {  // uniform across the workgroup (one trial)
      out = psd_project(c, r, 1e-15);
      QT_STAMP(6);
      if (xl) x = cholesky_param(c, out, ok);
      else ok = 1;
      QT_STAMP(7);
    }
    if (xl) *xl = x;
    if (ok_out) *ok_out = ok;
    return out;
  }

  __device__ static cd build_llh(const Ctx& c, double xt, double& tr) {
    double* vx = c.vec();
    cd* L = c.L();
    vx[c.t] = xt;
    tr = bsum(c, xt * xt);  // barriers inside publish vx
    const int i = c.i, j = c.j;
    cd lij{0.0, 0.0};
    if (i == j) lij.re = vx[i];
    else if (i > j) {
      const int tt = (i * (i - 1)) / 2 + j;
      lij = cd{vx[d + tt], vx[d + T + tt]};
    }
    L[c.e] = lij;
    __syncthreads();
    cd m{0.0, 0.0};
    const int kmax = i < j ? i : j;
    for (int k = 0; k <= kmax; ++k) m = cadd(m, cmulc(L[i * LD + k], L[j * LD + k]));
    return m;
  }

  // ---- a9: NLL value + exact gradient (product POVM) -------------------------------------------------
  // Always inlined.  Inside the BFGS loop the caller hands over a Ctx whose LDS base and thread indices it has just
  // re-derived from values the compiler cannot see through (k_mle_large_bfgs): otherwise hipcc hoists the ~100
  // loop-invariant LDS addresses of the unrolled Pauli transforms out of that loop and spills them (1.4 KB of scratch
  // per lane, reloaded inside every Cholesky step).  An out-of-line version (rounds 1-2) paid the callee-saved
  // register saves instead: 176-288 bytes of scratch per lane.
  __device__ __forceinline__ static void nll_grad_inl(const Ctx& c, double xt, double& f, double& gt) {
    double tr;
    QT_STAMP(11);
    const cd m = build_llh(c, xt, tr);
    QT_STAMP(12);
    cd* A = c.Aimg();
    A[c.e] = cd{m.re / tr, m.im / tr};
    __syncthreads();
    const double bl = bloch_of(c, A);
    QT_STAMP(13);
    double* vec = c.vec();
    vec[c.t] = bl;
    __syncthreads();
    double wl;
    if (!c.pr.enabled) {  // dense operands: p = d A' b, r = f / (p + eps), w = A'^T r   (uniform per launch)
      double fpart = 0.0;
      for (int m = c.t; m < c.M; m += NT) {
        const double pe = row_dot_dense(c, c.AwT, m, vec) * d + 1e-10;
        const double fr = (double)c.counts[m] / c.tot;
        fpart += fr * fast_log(pe);
        c.X()[m] = fr * recip_nr(pe);
      }
      f = -bsum(c, fpart);  // barriers inside publish X
      QT_STAMP(15);
      wl = col_dot_dense(c, c.Aw, c.X());
      QT_STAMP(16);
    } else {
    const int R1 = c.pr.R1;
    const double* in = vec;
    for (int q = 1; q < NQ; ++q) {  // stages 1 .. n-1 alternate so that stage n-1 lands in Y
      const int lk = 2 * (NQ - q);
      const int n_out = ipow(R1, q) << lk;
      double* out = ((NQ - 1 - q) & 1) ? c.X() : c.Y();
      stage<true>(c, c.tabT(), lk, n_out, in, out);
      in = out;
    }
    QT_STAMP(14);
    double fpart = 0.0;
    for (int o = c.t; o < c.M; o += NT) {  // stage n fused with the likelihood terms; Y_n -> X
      const double xn = stage_value<true>(c.tabT(), R1, stage_entry<true>(R1, 0, o), 1, in);
      const double wrow = c.pr.wrowR[o];
      const double pe = xn * wrow * d + 1e-10;
      const double fr = freq(c, o);
      fpart += fr * fast_log(pe);
      c.X()[o] = wrow * fr * recip_nr(pe);
    }
    f = -bsum(c, fpart);  // barriers inside publish X
    QT_STAMP(15);
    wl = prod_backward(c, c.tabT());
    QT_STAMP(16);
    }
    const double tr_g_rho = -(double)d * bsum(c, wl * bl);
    vec[c.t] = wl;
    __syncthreads();
    cd g = matrix_of(c, vec, c.Vimg());  // (exchange through the X overlay: free since prod_backward returned)
    QT_STAMP(17);
    g.re = -g.re;
    g.im = -g.im;
    if (c.i == c.j) g.re -= tr_g_rho;
    g.re /= tr;
    g.im /= tr;
    A[c.e] = g;
    __syncthreads();
    // Q = Gt L on the FP64 matrix cores (the product helper of the sign-function clip; result in the X overlay, which
    // is free here): as a 2 d-read loop per thread it was LDS-bound, 14 k of the 78 k clocks of an evaluation at n = 5
    cd q{0.0, 0.0};
    if constexpr (NQ >= 5) {
      cd* Q = c.Vimg();
      SignClipWG<d, NT>::matmul(A, c.L(), Q);
      q = Q[c.pi * LD + c.pj];
    } else {  // d = 16: the 16-term loop is cheaper than a product with its barrier (measured: 0.101 vs 0.122 ms per 1024)
      const cd* L = c.L();
#pragma unroll 8
      for (int k = 0; k < d; ++k) {  // all k (L[k][pj] = 0 for k < pj): a fixed trip count can be pipelined
        const cd u = A[c.pi * LD + k], v = L[k * LD + c.pj];
        q.re = fma(u.re, v.re, fma(-u.im, v.im, q.re));
        q.im = fma(u.re, v.im, fma(u.im, v.re, q.im));
      }
    }
    gt = 2.0 * (c.pkind == 2 ? q.im : q.re);
    __syncthreads();
    QT_STAMP(18);
  }
};

// =========================================================================================
// kernels: one workgroup (D threads) per trial; dynamic LDS = Large<NQ>::lds_bytes(M, R1)
// =========================================================================================

template <int NQ>
__global__ void __launch_bounds__(Large<NQ>::NT) k_lin_large(PovmView pv, const int64_t* __restrict__ counts, int B,
                                                             int physical, EstOut rho,
                                                             double* __restrict__ bloch_out, int32_t* __restrict__ status) {
  using S = Large<NQ>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x;
  if (b >= B) return;
  typename S::Ctx c;
  S::make_ctx(c, smem, pv, counts + (size_t)b * pv.M);
  double bl;
  cd r = S::lin_invert(c, bl);
  if (physical) r = S::make_feasible(c, r, nullptr, nullptr);
  S::emit(c, rho, b, r);
  if (bloch_out) bloch_out[(size_t)b * S::D + c.t] = bl;
  if (status && c.t == 0) status[b] = !c.shots_ok ? 5 : (r.re == r.re) ? 0 : 4;
}

template <int NQ>
__global__ void __launch_bounds__(Large<NQ>::NT) k_chol_param_large(PovmView pv, const double* __restrict__ rho, int B,
                                                                    double* __restrict__ x, int32_t* __restrict__ status) {
  using S = Large<NQ>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x;
  if (b >= B) return;
  typename S::Ctx c;
  S::make_ctx(c, smem, pv, nullptr);
  const double* in = rho + ((size_t)b * S::D + c.t) * 2;
  int ok;
  const double xt = S::cholesky_param(c, cd{in[0], in[1]}, ok);
  x[(size_t)b * S::D + c.t] = xt;
  if (status && c.t == 0) status[b] = ok ? 0 : 1;
}

template <int NQ>
__global__ void __launch_bounds__(Large<NQ>::NT) k_chol_unparam_large(PovmView pv, const double* __restrict__ x, int B,
                                                                      double* __restrict__ llh) {
  using S = Large<NQ>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x;
  if (b >= B) return;
  typename S::Ctx c;
  S::make_ctx(c, smem, pv, nullptr);
  double tr;
  const cd m = S::build_llh(c, x[(size_t)b * S::D + c.t], tr);
  double* out = llh + ((size_t)b * S::D + c.t) * 2;
  out[0] = m.re;
  out[1] = m.im;
}

// a4 for a product POVM at n = 4, 5: p = d * K b through the same n contraction stages as the forward half
// of the NLL (1.2e5 FMAs per state at n = 5 instead of the 1.6e7 of the dense 7776 x 1024 product);
// one workgroup per state, output in the caller's (S, K) order through the R-order row map.
template <int NQ>
__global__ void __launch_bounds__(Large<NQ>::NT) k_born_large(PovmView pv, const double* __restrict__ bloch, int B,
                                                              double* __restrict__ p) {
  using S = Large<NQ>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x;
  if (b >= B) return;
  typename S::Ctx c;
  S::make_ctx(c, smem, pv, nullptr);
  double* vec = c.vec();
  vec[c.t] = bloch[(size_t)b * S::D + c.t];
  __syncthreads();
  const int R1 = c.pr.R1;
  const double* in = vec;
  for (int q = 1; q < NQ; ++q) {
    const int lk = 2 * (NQ - q);
    const int n_out = S::ipow(R1, q) << lk;
    double* out = ((NQ - 1 - q) & 1) ? c.X() : c.Y();
    S::template stage<true>(c, c.tabT(), lk, n_out, in, out);
    in = out;
  }
  for (int o = c.t; o < c.M; o += S::NT) {
    double v = S::template stage_value<true>(c.tabT(), R1, S::template stage_entry<true>(R1, 0, o), 1, in) * S::d;
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    p[(size_t)b * c.M + c.pr.rmap[o]] = v;
  }
}

template <int NQ>
__global__ void __launch_bounds__(Large<NQ>::NT) k_nll_large(PovmView pv, const double* __restrict__ x,
                                                             const int64_t* __restrict__ counts, int B,
                                                             double* __restrict__ f, double* __restrict__ grad) {
  using S = Large<NQ>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x;
  if (b >= B) return;
  typename S::Ctx c;
  S::make_ctx(c, smem, pv, counts + (size_t)b * pv.M);
  double fv, gt;
  S::nll_grad_inl(c, x[(size_t)b * S::D + c.t], fv, gt);
  if (c.t == 0) f[b] = fv;
  if (grad) grad[(size_t)b * S::D + c.t] = gt;
}

// a10 for n = 4, 5 in two kernels (as at n <= 3): the whole MLE of one trial used to be ONE kernel whose BFGS loop --
// line-search state, out-of-line evaluation -- dictated the register allocation of the start phase too (110 VGPRs
// spilled, 432 bytes of scratch per lane at n = 5, every phase reloading through scratch).
//   k_mle_large_start: start point (state.py:205-212), Cholesky parametrisation, first (value, gradient); trials whose
//     gradient already meets gtol -- every full-rank high-shot one, SURVEY 0 fact 2 -- are finished here; the rest hand
//     x0, g0, f0 to
//   k_mle_large_bfgs: scipy's BFGS loop for the trials that iterate (workgroups of finished trials leave at once).
//     `pairs` = nb x max_iter x 2 x D doubles of workspace ((s_i, y_i) of every accepted step); LDS carries rho_i and
//     the two-loop alphas (2 x max_iter doubles) and, across each out-of-line evaluation, the line-search state.
// Four wavefronts per SIMD (128 registers, AGPRs of the sign-clip MFMA included): at n = 4 the allocator otherwise
// settles on 122 + 8, one past the step, three workgroups per CU instead of four and 20 % slower (A/B in
// profiles/round2_n4_start_occupancy.txt); at n = 5 the 1024-thread workgroup implies the same cap.
template <int NQ>
__global__ void __launch_bounds__(Large<NQ>::NT) __attribute__((amdgpu_waves_per_eu(4)))
k_mle_large_start(PovmView pv, const int64_t* __restrict__ counts, int B,
                                                                   int init, int max_iter, double gtol,
                                                                   EstOut rho, int32_t* __restrict__ nit_out,
                                                                   int32_t* __restrict__ nfev_out, double* __restrict__ fun_out,
                                                                   int32_t* __restrict__ status_out, double* __restrict__ ws_x,
                                                                   double* __restrict__ ws_g, double* __restrict__ ws_f,
                                                                   int32_t* __restrict__ ws_active) {
  using S = Large<NQ>;
  constexpr int D = S::D, d = S::d;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x;
  if (b >= B) return;
  typename S::Ctx c;
  QT_STAMP(0);
  S::make_ctx(c, smem, pv, counts + (size_t)b * pv.M);
  QT_STAMP(1);
  int ok;
  double xk;
  if (init == 0) {
    double bl;
    const cd lin = S::lin_invert(c, bl);
    QT_STAMP(2);
    S::make_feasible(c, lin, &xk, &ok);
  } else {
    xk = S::cholesky_param(c, cd{c.i == c.j ? 1.0 / d : 0.0, 0.0}, ok);
  }
  QT_STAMP(8);
  double fk = 0.0, gk = 0.0;
  int status = !c.shots_ok ? 5 : (ok ? 0 : 1);
  bool iterate = false;
  if (status == 0) {  // uniform: one trial per workgroup
    S::nll_grad_inl(c, xk, fk, gk);
    QT_STAMP(9);
    const double gnorm = S::bmax(c, fabs(gk));
    iterate = (gnorm > gtol) && (0 < max_iter);
    if (!iterate) {
      const double xn = S::bmax(c, fabs(xk));
      if (0 >= max_iter) status = 3;
      else if (gnorm != gnorm || fk != fk || xn != xn) status = 4;
    }
  }
  // what the trial returns if BFGS does not move: L L^dagger / Tr at x_k (state.py:214-215)
  double tr;
  const cd m = S::build_llh(c, xk, tr);
  S::emit(c, rho, b, cd{m.re / tr, m.im / tr});
  if (iterate) {
    ws_x[(size_t)b * D + c.t] = xk;
    ws_g[(size_t)b * D + c.t] = gk;
  }
  QT_STAMP(10);
  if (c.t == 0) {
    if (iterate) ws_f[b] = fk;
    ws_active[b] = iterate ? 1 : 0;
    if (nit_out) nit_out[b] = 0;
    if (nfev_out) nfev_out[b] = status == 0 || status >= 3 ? (status == 5 ? 0 : 1) : 0;
    if (fun_out) fun_out[b] = fk;
    if (status_out) status_out[b] = status;
  }
}

template <int NQ>
__global__ void __launch_bounds__(Large<NQ>::NT) __attribute__((amdgpu_waves_per_eu(4)))
k_mle_large_bfgs(PovmView pv, const int64_t* __restrict__ counts, int B,
                                                                  int max_iter, double gtol, EstOut rho,
                                                                  int32_t* __restrict__ nit_out, int32_t* __restrict__ nfev_out,
                                                                  double* __restrict__ fun_out, int32_t* __restrict__ status_out,
                                                                  const double* __restrict__ ws_x, const double* __restrict__ ws_g,
                                                                  const double* __restrict__ ws_f,
                                                                  const int32_t* __restrict__ ws_active,
                                                                  double* __restrict__ pairs) {
  using S = Large<NQ>;
  constexpr int D = S::D;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = blockIdx.x;
  if (b >= B || ws_active[b] == 0) return;  // uniform per workgroup
  typename S::Ctx c;
  S::make_ctx(c, smem, pv, counts + (size_t)b * pv.M);
  double xk = ws_x[(size_t)b * D + c.t], gk = ws_g[(size_t)b * D + c.t], fk = ws_f[b];
  int kiter = 0, nfev = 1, status = 0;
  double* my = pairs + (size_t)b * max_iter * 2 * D + c.t;  // s_i[t] at my[2 i D], y_i[t] at my[(2 i + 1) D]
  double* prho = c.pair_rho();
  double* palpha = prho + max_iter;
  double* lsb = c.lsbuf();
  double old_old = fk + sqrt(S::bsum(c, gk * gk)) / 2.0;
  double pk = -gk, stp = 0.0;  // H_0 = I
  {
    LineSearch ls;
    ls.start(fk, old_old, S::bsum(c, gk * pk), &stp);
    if (c.t == 0) ls.save(lsb);
  }
  const int eval_cap = (max_iter + 2) * 130;
  while (true) {  // uniform: one trial per workgroup, every thread holds the same scalars
    double ft, gt;
    QT_STAMP(21);  // (profile build, scripts/phase_timing_large_bfgs.py: the slots keep the LAST iteration's clocks)
    // Inlined, with the LDS base and the per-thread indices laundered through an empty asm every iteration: the
    // compiler can then neither hoist the ~100 loop-invariant LDS addresses of the evaluation out of the loop (that is
    // what spilled 1.4 KB per lane when it was first inlined) nor does the kernel pay the callee-saved-register saves of
    // an out-of-line call (224 / 288 bytes of scratch per lane).  Both sizes are held to 128 registers -- four
    // wavefronts per SIMD: the 1024-thread workgroup of n = 5 implies it, and at n = 4 four 256-thread workgroups per CU
    // instead of two are worth more than the registers (209 VGPRs / 0 scratch: 0.686 ms per 1024 mixed-start trials;
    // 128 VGPRs / 124 B: 0.570 ms).  What remains in scratch (n = 5: 188 B) is stored before the loop and reloaded in
    // its outer blocks, none of it inside a Cholesky / contraction loop.
    {
      typename S::Ctx ci = c;
      int off = 0, v0 = 0;
      asm volatile("" : "+s"(off));
      ci.sm = c.sm + off;
      asm volatile("" : "+v"(v0));
      ci.t = c.t + v0, ci.i = c.i + v0, ci.j = c.j + v0, ci.e = c.e + v0;
      ci.xm = c.xm + v0, ci.zm = c.zm + v0, ci.ny = c.ny + v0, ci.pi = c.pi + v0, ci.pj = c.pj + v0;
      S::nll_grad_inl(ci, xk + stp * pk, ft, gt);  // (barriers inside publish the parked line-search state)
    }
    QT_STAMP(22);
    if (++nfev > eval_cap) {
      status = 2;
      break;
    }
    const double dphi = S::bsum(c, gt * pk);
    LineSearch ls;
    ls.load(lsb);
    double next = stp;
    const int r = ls.advance(stp, ft, dphi, &next);
    if (r == LS_EVAL) {
      stp = next;
      __syncthreads();  // every thread has read the old state
      if (c.t == 0) ls.save(lsb);
      continue;
    }
    if (r == LS_FAIL) {
      status = 2;
      break;
    }
    const double sk = stp * pk;
    const double pnorm2 = S::bsum(c, pk * pk);
    xk = xk + sk;
    const double yk = gt - gk;
    gk = gt;
    old_old = fk;
    fk = ft;
    ++kiter;
    const double gnorm = S::bmax(c, fabs(gk));
    if (!(gnorm > gtol)) break;
    if (stp * sqrt(pnorm2) <= 0.0) break;
    if (!isfinite(fk)) {
      status = 2;
      break;
    }
    if (!(kiter < max_iter)) break;
    const double ys = S::bsum(c, yk * sk);
    const double rhok = (ys == 0.0) ? 1000.0 : 1.0 / ys;
    const int np = kiter - 1;  // index of the new pair
    my[(size_t)(2 * np) * D] = sk;
    my[(size_t)(2 * np + 1) * D] = yk;
    if (c.t == 0) prho[np] = rhok;
    __syncthreads();
    QT_STAMP(23);
    // two-loop recursion: p = -H_k g with H_0 = I
    // (round 3: the pair of the NEXT round is requested before this round's reduction; arithmetic and order unchanged.
    //  Four rounds ahead was measured too: the rings cost 90 more bytes of scratch and the run came out 25 % slower.)
    double q = gk;
    double si = sk, yi = yk;
    for (int i = np; i >= 0; --i) {
      double sn = 0.0, yn = 0.0;
      if (i > 0) {
        sn = my[(size_t)(2 * (i - 1)) * D];
        yn = my[(size_t)(2 * (i - 1) + 1) * D];
      }
      const double a = prho[i] * S::bsum_alt(c, si * q, i & 1);
      if (c.t == 0) palpha[i] = a;
      q = fma(-a, yi, q);
      if (i > 0) si = sn, yi = yn;  // (the last round's pair, i = 0, is the first of the second loop)
    }
    __syncthreads();
    QT_STAMP(24);
    for (int i = 0; i <= np; ++i) {
      double sn = sk, yn = yk;
      if (i + 1 < np) {
        sn = my[(size_t)(2 * (i + 1)) * D];
        yn = my[(size_t)(2 * (i + 1) + 1) * D];
      }
      const double bb = prho[i] * S::bsum_alt(c, yi * q, i & 1);
      q = fma(si, palpha[i] - bb, q);
      si = sn, yi = yn;
    }
    pk = -q;
    QT_STAMP(19);
    ls.start(fk, old_old, S::bsum(c, gk * pk), &stp);  // (the barriers of bsum are behind every read of lsb)
    if (c.t == 0) ls.save(lsb);
    QT_STAMP(20);
  }
  if (status == 0) {
    const double gn = S::bmax(c, fabs(gk));
    const double xn = S::bmax(c, fabs(xk));
    if (kiter >= max_iter) status = 3;
    else if (gn != gn || fk != fk || xn != xn) status = 4;
  }
  double tr;
  const cd m = S::build_llh(c, xk, tr);
  S::emit(c, rho, b, cd{m.re / tr, m.im / tr});
  if (c.t == 0) {
    if (nit_out) nit_out[b] = kiter;
    if (nfev_out) nfev_out[b] = nfev;
    if (fun_out) fun_out[b] = fk;
    if (status_out) status_out[b] = status;
  }
}

}  // namespace qt
