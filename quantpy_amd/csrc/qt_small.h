// Wave-synchronous kernels for n <= 3 qubits (D = 4^n <= 64 = one wavefront).
//
// Mapping: one trial is owned by a group of G = D lanes of a 64-wide wavefront (64/G trials per
// wave; at n = 3 one trial per wave).  Lane l of a group plays three roles at once:
//   - Bloch / parameter index k = l  (vectors b, w, x, g of length D live one element per lane),
//   - matrix element (i, j) = (l / d, l % d) of every d x d complex matrix (rho, L, G, V),
//   - row m = c*G + l of the M-row POVM contraction, chunk by chunk.
// A workgroup is WPB = 4 waves, one per SIMD of a CU.  The waves share ONE read-only LDS image of
// the current (M x D) operand (left inverse for 'lin', then the weighted POVM A' for the NLL),
// row-padded to D + 1 doubles so that both access directions -- lane = column (A'^T r, A^+ f) and
// lane = row (A' b) -- are bank-conflict free.  Measured motivation (profiles/round1_v1_*): with
// every wave streaming its own 111 KB of A' from L2 per evaluation the batch pulled 14-22 TB/s out
// of L2 and each wave, alone on its SIMD, sat on L2 latency; from LDS the operand is read at LDS
// rate and HBM/L2 see it once per workgroup.  After the cooperative image load the waves run
// independently (wave-level fences only), so trials diverge freely inside BFGS.
// Per-trial scratch (rho, L, V, b, r, f: ~7 KB) also lives in LDS; the BFGS inverse Hessian
// (D x D f64) lives in registers, one row per lane.  POVMs too large for the image (M (D+1) 8 B
// + scratch > 160 KB) take the ALDS = false instantiation, which streams the operand from L2.
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

// Optional per-wave phase stamps (shader clock) for scripts/phase_timing.py: compiled in only with
// -DQT_PHASE_TIMING (lib/libqtomo_prof.so); the product library carries none of this.
#ifdef QT_PHASE_TIMING
__device__ long long* g_qt_prof = nullptr;  // [waves][32]
#define QT_STAMP(slot)                                                                                   \
  do {                                                                                                   \
    if (g_qt_prof && (threadIdx.x & 63) == 0)                                                            \
      g_qt_prof[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32 + (slot)] =           \
          (long long)__builtin_readcyclecounter();                                                       \
  } while (0)
#define QT_STAMP_VAL(slot, val)                                                                          \
  do {                                                                                                   \
    if (g_qt_prof && (threadIdx.x & 63) == 0)                                                            \
      g_qt_prof[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 32 + (slot)] = (val);    \
  } while (0)
#else
#define QT_STAMP(slot) do {} while (0)
#define QT_STAMP_VAL(slot, val) do {} while (0)
#endif

// 16-byte aligned: a complex element then moves with ONE ds_read_b128 / ds_write_b128 (4 LDS cycles per
// wavefront read, conflict-free on consecutive elements).  With 8-byte alignment hipcc emits ds_read2_b64 /
// ds_write2_b64, whose two 8-byte halves each stride 16 bytes across the lanes: a built-in 2-way bank
// conflict, 16 cycles per read (measured on the n = 5 Jacobi rounds: 2700 -> see DESIGN 4.5).  Every cd
// array in LDS starts at an even double offset (checked where the layouts are defined).
struct alignas(16) cd {
  double re, im;
};
__device__ __forceinline__ cd cmul(cd a, cd b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
__device__ __forceinline__ cd cmulc(cd a, cd b) {  // a * conj(b)
  return {a.re * b.re + a.im * b.im, a.im * b.re - a.re * b.im};
}
__device__ __forceinline__ cd cadd(cd a, cd b) { return {a.re + b.re, a.im + b.im}; }
// One term Delta_ij Delta_ji of Tr(Delta^2) (geometry.py:16), with its rounding pinned (one product rounded, then one
// fma) so that every kernel that forms a Hilbert-Schmidt distance gets the same bits from the same operands.
__device__ __forceinline__ cd hs_term(cd a, cd b) {
  return {__builtin_fma(a.re, b.re, -(a.im * b.im)), __builtin_fma(a.re, b.im, a.im * b.re)};
}
__device__ __forceinline__ cd cscale(cd a, double s) { return {a.re * s, a.im * s}; }

// LDS hand-off inside ONE wavefront: order this wave's LDS writes before its later reads.  The hardware already does:
// the LDS executes the DS instructions of a wavefront in issue order, each for all 64 lanes before the next starts, so a
// ds_read behind a ds_write of the same wave sees the data whichever lane wrote it.  What is needed is that the
// COMPILER keeps the order (to it, another lane's slot is just a different address): a scheduling barrier plus an
// empty asm with a memory clobber.  Rounds 1-2 used release / acquire fences at workgroup scope here, which cost an
// `s_waitcnt lgkmcnt(0)` -- a full drain of the LDS queue, ~100 clocks -- at each of the ~100 hand-offs of a trial.
// (Never a cross-wave hand-off: every use is inside one trial's own scratch; tables shared by a workgroup are
// published by __syncthreads.)
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// ---- group reductions on the DPP network (no LDS round trips) ---------------------------------
// A butterfly inside rows of 16 lanes (quad_perm, row_half_mirror, row_mirror) and, for G = 64,
// four v_readlane across the rows.  Every lane of a group ends with the same bits.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double v) {
  // Full row mask: every lane has a source under the permutations used here, so bound_ctrl lets the
  // compiler drop the zero-initialisation of the destination.  Partial row mask (row_bcast steps):
  // the rows left out must read as 0.
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(b & 0xffffffffLL), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROW_MASK, 0xf, ROW_MASK == 0xf);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), lane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double nanmax(double a, double b) { return (b > a || b != b) ? b : a; }  // NaN wins

// Across the four rows of a wavefront: row_bcast15 adds the total of row 0 / 2 into row 1 / 3,
// row_bcast31 adds lane 31 (rows 0+1) into rows 2 and 3, lane 63 then holds (r2 + r3) + (r0 + r1) and
// is handed to every lane through an SGPR pair (6 + 2 instructions instead of 8 readlanes + 7).
template <int G>
__device__ __forceinline__ double gsum(double v) {
  v += dpp_f64<0xB1>(v);                 // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);                 // quad_perm [2,3,0,1]
  if (G >= 8) v += dpp_f64<0x141>(v);    // row_half_mirror
  if (G >= 16) v += dpp_f64<0x140>(v);   // row_mirror
  if (G == 64) {
    v += dpp_f64<0x142, 0xA>(v);         // row_bcast15 -> rows 1, 3
    v += dpp_f64<0x143, 0xC>(v);         // row_bcast31 -> rows 2, 3
    v = readlane_f64(v, 63);
  }
  return v;
}
// Maximum of non-negative values (every caller passes fabs(.)); a NaN anywhere wins.  The rows a
// row_bcast step leaves out read 0, which is neutral for such inputs.
template <int G>
__device__ __forceinline__ double gmax(double v) {
  v = nanmax(v, dpp_f64<0xB1>(v));
  v = nanmax(v, dpp_f64<0x4E>(v));
  if (G >= 8) v = nanmax(v, dpp_f64<0x141>(v));
  if (G >= 16) v = nanmax(v, dpp_f64<0x140>(v));
  if (G == 64) {
    v = nanmax(v, dpp_f64<0x142, 0xA>(v));
    v = nanmax(v, dpp_f64<0x143, 0xC>(v));
    v = readlane_f64(v, 63);
  }
  return v;
}

// log(x) and 1/x for the likelihood terms.  The OCML log is correctly rounded through ~85 FP64
// instructions of double-double arithmetic; the NLL needs neither that nor the special cases beyond
// "x <= 0 or NaN gives NaN / -inf", and a lone wave pays for every instruction.  x = m 2^e with
// m in [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1), log m = 2 s (1 + s^2/3 + s^4/5 + ...): with
// s^2 <= 0.0295 ten terms reach 2^-55; error ~1.5 ulp.  About 30 instructions.
__device__ __forceinline__ double recip_nr(double x) {  // 1/x to ~1 ulp: hardware seed + two Newton steps
  double r = __builtin_amdgcn_rcp(x);
  r = fma(r, fma(-x, r, 1.0), r);
  return fma(r, fma(-x, r, 1.0), r);
}
__device__ __forceinline__ double fast_log(double x) {
  if (!(x > 0.0) || !(x < 1.7e308)) return x == 0.0 ? -__builtin_huge_val() : (x > 0.0 ? x : __builtin_nan(""));
  double m = __builtin_amdgcn_frexp_mant(x);  // [0.5, 1)
  int e = __builtin_amdgcn_frexp_exp(x);
  const bool low = m < 0.70710678118654752440;
  m = low ? m + m : m;
  e = low ? e - 1 : e;
  const double num = m - 1.0, den = m + 1.0;
  const double r = recip_nr(den);
  double sq = num * r;
  sq = fma(r, fma(-sq, den, num), sq);  // s = num / den, correctly rounded up to the last bit
  const double z = sq * sq;
  // P(z) = 1/3 + z/5 + ... + z^9/21, Estrin's scheme: pairs, then powers z^2, z^4, z^8 (dependency depth 5
  // instead of the 10 of Horner's rule)
  const double z2 = z * z, z4 = z2 * z2, z8 = z4 * z4;
  const double q0 = fma(z, 1.0 / 5.0, 1.0 / 3.0), q1 = fma(z, 1.0 / 9.0, 1.0 / 7.0);
  const double q2 = fma(z, 1.0 / 13.0, 1.0 / 11.0), q3 = fma(z, 1.0 / 17.0, 1.0 / 15.0);
  const double q4 = fma(z, 1.0 / 21.0, 1.0 / 19.0);
  const double r0 = fma(z2, q1, q0), r1 = fma(z2, q3, q2);
  const double pl = fma(z8, q4, fma(z4, r1, r0));
  const double lm = fma(sq + sq, pl * z, sq + sq);  // 2 s (1 + z P(z))
  const double ed = (double)e;
  return fma(ed, 6.93147180369123816490e-01, fma(ed, 1.90821492927058770002e-10, lm));  // e ln2 (hi + lo) + log m
}

constexpr bool kLiftSingleNegative = true;  // a7 short cut of Small::lift_single_negative (n = 3)

// 1/sqrt(x) for normal-range positive x: hardware seed (v_rsq_f64) plus two Newton steps -- ~8
// dependent FP64 instructions instead of the ~25 of a correctly rounded sqrt-then-divide,
// accurate to ~1 ulp, which is all a Jacobi rotation / Cholesky pivot needs.
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double e = fma(-x * y, y, 1.0);
  y = fma(y * e, fma(e, 0.375, 0.5), y);
  e = fma(-x * y, y, 1.0);
  return fma(y * e, 0.5, y);
}

// sum_m base[m * stride + lane_off] * lds_vec[m], m < n, operand streamed from global memory (L2):
// the loads of a block of UNR rows are issued back to back before any is consumed (the
// sched_group_barriers pin that order; hipcc otherwise interleaves load / wait / FMA pairs), and
// four accumulators keep the FP64 FMAs from forming a single dependency chain.
template <int UNR>
__device__ __forceinline__ void dot_block(const double* __restrict__ base, size_t stride, unsigned lane_off,
                                          const double* lds_vec, int m, double (&acc)[4]) {
  double av[UNR];
#pragma unroll
  for (int u = 0; u < UNR; ++u) av[u] = base[(size_t)(m + u) * stride + lane_off];
#pragma unroll
  for (int u = 0; u < UNR; ++u) acc[u & 3] = fma(av[u], lds_vec[m + u], acc[u & 3]);
  __builtin_amdgcn_sched_group_barrier(0x020, UNR, 0);
#pragma unroll
  for (int u = 0; u < UNR; u += 4) {
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
  }
}
template <int UNR>
__device__ __forceinline__ double dot_global(const double* __restrict__ base, size_t stride, unsigned lane_off,
                                             const double* lds_vec, int n) {
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  int m = 0;
  for (; m + UNR <= n; m += UNR) dot_block<UNR>(base, stride, lane_off, lds_vec, m, acc);
  if (UNR > 8 && m + 8 <= n) {
    dot_block<8>(base, stride, lane_off, lds_vec, m, acc);
    m += 8;
  }
  if (UNR > 4 && m + 4 <= n) {
    dot_block<4>(base, stride, lane_off, lds_vec, m, acc);
    m += 4;
  }
  for (; m < n; ++m) acc[0] = fma(base[(size_t)m * stride + lane_off], lds_vec[m], acc[0]);
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
// The same product with the operand in the LDS image: img[m * stride + lane_off].
__device__ __forceinline__ double dot_lds(const double* img, int stride, int lane_off, const double* lds_vec, int n) {
  double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
  const double* p = img + lane_off;
  int m = 0;
#pragma unroll 4
  for (; m + 4 <= n; m += 4) {
    a0 = fma(p[(m + 0) * stride], lds_vec[m + 0], a0);
    a1 = fma(p[(m + 1) * stride], lds_vec[m + 1], a1);
    a2 = fma(p[(m + 2) * stride], lds_vec[m + 2], a2);
    a3 = fma(p[(m + 3) * stride], lds_vec[m + 3], a3);
  }
  for (; m < n; ++m) a0 = fma(p[m * stride], lds_vec[m], a0);
  return (a0 + a1) + (a2 + a3);
}

// Unitary 2x2 rotation J = [[c, w], [-conj(w), c]] that (nearly) annihilates a_pq of
// [[app, apq], [conj(apq), aqq]]: with delta = (aqq - app)/2 and
// u = t/|apq| = sign(delta) / (|delta| + sqrt(delta^2 + |apq|^2)):  c = 1/sqrt(1 + u^2 |apq|^2),
// w = s e^{i phi} = c u apq.  Only c has to be accurate (it makes J unitary for whatever u is
// used); u comes from hardware seeds with one Newton step each (~1e-13): the rotated a_pq is
// kept as computed instead of being set to zero, so an inexact angle costs convergence speed
// (nothing measurable), never accuracy.
__device__ __forceinline__ void rotation(double app, double aqq, cd apq, double& cs, cd& w) {
  const double ab2 = apq.re * apq.re + apq.im * apq.im;
  cs = 1.0;
  w = cd{0.0, 0.0};
  if (ab2 > 1e-300) {
    const double dl = 0.5 * (aqq - app);
    const double x = fma(dl, dl, ab2);
    double y = __builtin_amdgcn_rsq(x);
    y = fma(y * fma(-x * y, y, 1.0), 0.5, y);
    const double den = fabs(dl) + x * y;  // |delta| + sqrt(delta^2 + |apq|^2)
    double z = __builtin_amdgcn_rcp(den);
    z = fma(z, fma(-den, z, 1.0), z);
    const double u = copysign(z, dl);
    cs = fast_rsqrt(fma(u * u, ab2, 1.0));
    const double cu = cs * u;
    w = cd{cu * apq.re, cu * apq.im};
  }
}


// A POVM that is the n-fold tensor product of a one-qubit table T[R1][4] (every built-in POVM and
// every array with a last axis of 4 -- measurements.py:88-93): the M x D contractions A' b, A'^T r and
// A'^+ f factor into n small ones, qubit by qubit (R1^q 4^(n-q) outputs of 4 or R1 terms each), which
// cuts the FP64 work per evaluation ~8x at n = 3 and removes every global load from the inner loops.
// Intermediate arrays live in the trial's LDS scratch in "R-order": index [r_1 .. r_q][k_(q+1) .. k_n]
// with r = s * K1 + o the row of T.  The (base, selector) of every output is tabulated on the host.
struct ProductView {
  const double* T;      // [R1][4]
  const double* P1T;    // [R1][4]  P1T[r][k] = pinv(T)[k][r]  (used when the shot weights are uniform)
  const double* wrowR;  // [M]  N_s / sum(N) of each row, R-order
  const int* rmap;      // [M]  R-order index -> row m of the (S, K) layout
  const int* rinv;      // [M]  the inverse map: row m of the (S, K) layout -> R-order index
  const int* fwd;       // stage tables of the forward pass, stage 1 .. n concatenated: base | sel << 16
  const int* bwd;       // stage tables of the backward pass, stage n .. 1 concatenated
  int R1;
  int uniform;          // all N_s equal
  double wuni;          // the common weight
  int enabled;
};

struct PovmView {
  const double* Aw;    // [M][D]  shot-weighted A'
  const double* AwT;   // [D][M]
  const double* PinvT; // [M][D]  transpose of the left inverse of A'
  int M;
  ProductView pr;
  double jtol2 = 1e-28;  // Jacobi stops when off-diagonal norm^2 <= jtol2 * Frobenius norm^2
  // shots per setting as registered with qt_set_povm (the weights N_s / sum N inside A' come from these);
  // every trial's own per-setting totals must be proportional to them (state.py:138-141, 194-197: the
  // reference takes the weights from the trial's results) or the trial is flagged QT_TRIAL_SHOTS
  const double* Ns = nullptr;  // [S]
  int S = 0, K = 0;
  double ns_tot = 0.0;
  int extra = 0;  // extra doubles of per-trial LDS behind the scratch (k_mle_bfgs: line-search state + two-loop scalars)
};

// t_s / total == N_s / ns_tot, compared as cross products of integer-valued doubles (exact below 2^53; a few
// ulp of slack beyond that -- one stray count out of N_s < 1e14 is still seen)
__device__ __forceinline__ bool shots_match(double t_s, double total, double n_s, double ns_tot) {
  const double a = t_s * ns_tot, b = n_s * total;
  return fabs(a - b) <= 2e-15 * fabs(b);
}

// Where an estimator leaves a trial's result: the density matrix and / or -- the body of the bootstrap loop,
// interval.py:600-609, `dist[i] = dst(point_estimate(...), state)` -- its Hilbert-Schmidt distance (geometry.py:16-20) to
// `centre`.  With `dist` set and `rho` null a resample costs 8 bytes of HBM writes instead of 16 d^2 and no second pass.
struct EstOut {
  double* rho;           // [B][d][d][2], or nullptr when only the distance is wanted
  const double* centre;  // [d][d][2], read when dist != nullptr
  double* dist;          // [B], or nullptr
};

template <int NQ, bool ALDS>
struct Small {
  static constexpr int d = 1 << NQ;
  static constexpr int D = d * d;
  static constexpr int G = D;
  static constexpr int TPW = 64 / G;       // trials per wave
  static constexpr int WPB = 4;            // waves per workgroup
  static constexpr int NT = 64 * WPB;      // threads per workgroup
  static constexpr int TPB = TPW * WPB;    // trials per workgroup
  static constexpr int T = d * (d - 1) / 2;
  static constexpr int LDA = D + 1;        // row pitch of the LDS operand image
  // Row pitch of the d x d complex matrix images: 9 at d = 8, so that the rows one column is read
  // from (pitch 144 B = 36 banks) fall on distinct LDS banks; a pitch of 8 (128 B) makes every such
  // ds_read_b128 a 4-way bank conflict.  d = 2, 4 have no conflict to begin with.
  static constexpr int LD = d == 8 ? 9 : d;
  static constexpr int MAT = 2 * LD * d;   // doubles per matrix image
  // per-trial LDS scratch, in doubles
  static constexpr int oA = 0;             // complex [d][LD]
  static constexpr int oB = oA + MAT;      // complex [d][LD]
  static constexpr int oVec = oB + MAT;    // [D] + one slot that always holds 0 (source of L's zero entries)
  static constexpr int oLam = oVec + D + 2;  // [d] (+ pad to even)
  static constexpr int oM = oLam + 2 * ((d + 1) / 2);  // rbuf[Mp], freq[Mp], bufB[Mp]
  // The third matrix image V (eigenvectors / projector / speculative inverse: make_feasible only) lives on top of rbuf
  // whenever rbuf is large enough: rbuf is a staging buffer of load_freq, lin_invert and nll_grad, none of which is
  // in flight while make_feasible runs.  At n = 3 with the 216-row POVM that takes a trial from 9.2 to 8.1 KB and a
  // 4-trial workgroup from 42.9 to 38.3 KB of LDS: FOUR workgroups per CU instead of three (the saturated batches
  // were LDS-limited to 3 waves per SIMD with registers for 4).
  __host__ __device__ static bool v_on_rbuf(int M) { return ((M + 1) & ~1) >= MAT; }
  __host__ __device__ static int v_offset(int M, int R1) {
    const int Mp = (M + 1) & ~1;
    return v_on_rbuf(M) ? oM : oM + (R1 > 0 ? 3 : 2) * Mp;
  }
  __host__ __device__ static int trial_doubles(int M, int R1 = 0) {
    const int Mp = (M + 1) & ~1;
    return oM + (R1 > 0 ? 3 : 2) * Mp + (v_on_rbuf(M) ? 0 : MAT);
  }
  __host__ __device__ static int image_doubles(int M) { return ALDS ? ((M * LDA + 1) & ~1) : 0; }
  // Index tables of a product POVM, staged once per workgroup (shared by its waves): forward stage
  // tables, backward stage tables, R-order row map.
  __host__ __device__ static int ipow_h(int b, int e) {
    int r = 1;
    for (int t = 0; t < e; ++t) r *= b;
    return r;
  }
  __host__ __device__ static int fwd_ints(int R1) {
    int t = 0;
    for (int q = 1; q <= NQ; ++q) t += ipow_h(R1, q) << (2 * (NQ - q));
    return t;
  }
  __host__ __device__ static int bwd_ints(int R1) {
    int t = 0;
    for (int q = 1; q <= NQ; ++q) t += ipow_h(R1, q - 1) << (2 * (NQ - q + 1));
    return t;
  }
  __host__ __device__ static int table_ints(int M, int R1) { return (fwd_ints(R1) + bwd_ints(R1) + M + 1) & ~1; }
  __host__ __device__ static int table_doubles(int M, int R1) {
    // index tables, then the row weights wrowR[M] (padded to even), then the one-qubit tables T and pinv(T)^T
    // ([R1][4] each, 16-byte aligned): staged ONCE per workgroup.  (The one-qubit tables used to be copied per trial
    // inside load_freq: two dependent global loads on the critical path of a lone wave, ~1 us of the 16 us step.)
    return R1 > 0 ? (table_ints(M, R1) / 2 + ((M + 1) & ~1) + 8 * R1) : 0;
  }
  __host__ __device__ static size_t lds_bytes(int M, int R1 = 0, int extra = 0) {
    return ((size_t)image_doubles(M) + table_doubles(M, R1) + (size_t)TPB * (trial_doubles(M, R1) + extra)) * sizeof(double);
  }

  // ---- per-lane context ---------------------------------------------------------------
  struct Ctx {
    int l, i, j, e;  // lane in group, matrix element, its slot i * LD + j in a matrix image
    double* sm;      // this trial's LDS scratch
    double* img;     // the workgroup's operand image (ALDS)
    const int *tfwd, *tbwd, *trmap;  // the workgroup's copy of the product-POVM index tables (LDS)
    const double* twrow;             // ... and of the row weights N_s / sum(N), R-order
    const double *ttab, *ptab;       // ... and of the one-qubit tables T [R1][4], pinv(T)^T [R1][4]
    int M, Mp;
    PovmView pv;
    // Pauli string k = l:  P_k[r][r ^ xm] = (-i)^ny (-1)^popc(r & zm)
    int xm, zm, ny;
    // Cholesky parameter owned by lane l: element (pi, pj), kind 0 diag / 1 real / 2 imag
    int pi, pj, pkind;
    // where element (i, j) of L finds its real / imaginary part in the parameter vector (D = the zero slot)
    int src_re, src_im;
    __device__ __forceinline__ cd* A() const { return reinterpret_cast<cd*>(sm + oA); }
    __device__ __forceinline__ cd* Bm() const { return reinterpret_cast<cd*>(sm + oB); }
    __device__ __forceinline__ cd* V() const { return reinterpret_cast<cd*>(sm + v_offset(M, pv.pr.enabled ? pv.pr.R1 : 0)); }
    __device__ __forceinline__ double* vec() const { return sm + oVec; }
    __device__ __forceinline__ double* lam() const { return sm + oLam; }
    __device__ __forceinline__ double* rbuf() const { return sm + oM; }
    __device__ __forceinline__ double* freq() const { return sm + oM + Mp; }
    __device__ __forceinline__ double* bufB() const { return sm + oM + 2 * Mp; }
    __device__ __forceinline__ const double* tabT() const { return ttab; }  // [R1][4]
    __device__ __forceinline__ const double* tabP() const { return ptab; }  // [R1][4]
    __device__ __forceinline__ bool prod() const { return pv.pr.enabled != 0; }
    __device__ __forceinline__ double* extra() const { return sm + trial_doubles(M, pv.pr.enabled ? pv.pr.R1 : 0); }  // [pv.extra]
  };

  // Trial handled by this lane's group; *live = false for the padding groups of the last block.
  __device__ __forceinline__ static int trial_index(int B, bool* live) {
    const int wave = threadIdx.x >> 6, tib = (threadIdx.x & 63) / G;
    const int b = (blockIdx.x * WPB + wave) * TPW + tib;
    *live = b < B;
    return b;
  }

  __device__ static void make_ctx(Ctx& c, double* smem_block, const PovmView& pv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    c.l = lane % G;
    const int slot = wave * TPW + lane / G;
    c.i = c.l / d;
    c.j = c.l % d;
    c.e = c.i * LD + c.j;
    c.M = pv.M;
    c.Mp = (pv.M + 1) & ~1;
    c.pv = pv;
    c.img = smem_block;
    const int r1 = pv.pr.enabled ? pv.pr.R1 : 0;
    int* tabs = reinterpret_cast<int*>(smem_block + image_doubles(pv.M));
    c.tfwd = tabs;
    c.tbwd = tabs + fwd_ints(r1);
    c.trmap = c.tbwd + bwd_ints(r1);
    double* wrow = smem_block + image_doubles(pv.M) + table_ints(pv.M, r1) / 2;
    double* t1 = wrow + ((pv.M + 1) & ~1);
    c.twrow = wrow;
    c.ttab = t1;
    c.ptab = t1 + 4 * r1;
    if (r1 > 0) {  // every thread of the workgroup comes through here once, before anything else
      const int nf = fwd_ints(r1), nb = bwd_ints(r1);
      for (int e = threadIdx.x; e < 4 * r1; e += NT) {
        t1[e] = pv.pr.T[e];
        t1[4 * r1 + e] = pv.pr.P1T[e];
      }
      for (int e = threadIdx.x; e < nf; e += NT) tabs[e] = pv.pr.fwd[e];
      for (int e = threadIdx.x; e < nb; e += NT) tabs[nf + e] = pv.pr.bwd[e];
      for (int e = threadIdx.x; e < pv.M; e += NT) {
        tabs[nf + nb + e] = pv.pr.rmap[e];
        wrow[e] = pv.pr.wrowR[e];
      }
      __syncthreads();
    }
    c.sm = smem_block + image_doubles(pv.M) + table_doubles(pv.M, r1) + slot * (trial_doubles(pv.M, r1) + pv.extra);
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
    c.src_re = c.src_im = D;
    if (c.i == c.j) c.src_re = c.i;
    else if (c.i > c.j) {
      const int tt = (c.i * (c.i - 1)) / 2 + c.j;
      c.src_re = d + tt;
      c.src_im = d + T + tt;
    }
    if (c.l == 0) c.sm[oVec + D] = 0.0;  // (this trial's scratch: c.sm is set above)
  }

  // Cooperative copy of a row-major [M][D] operand into the padded LDS image (whole workgroup).
  __device__ static void load_image(const Ctx& c, const double* __restrict__ g) {
    if (!ALDS) return;
    __syncthreads();  // every wave is done with the previous image
    const int total = c.M * D;
    for (int e = threadIdx.x; e < total; e += NT) c.img[(e / D) * LDA + (e % D)] = g[e];
    __syncthreads();
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
  // n = 3 (one trial per wavefront, lane = 8 x + r): as in qt_large.h, for a fixed x-mask the eight strings (x, z) make
  // the sum over r a length-8 Walsh-Hadamard transform of the shifted diagonal M[r][r ^ x] -- one LDS read and three
  // butterfly stages across the lanes (two quad_perm DPP moves, one ds_bpermute) instead of eight reads and eight
  // sign-and-accumulate steps per lane (~85 -> ~40 vector instructions); the value lands in lane (x, z) and goes to
  // lane k = pauli_index(x, z) through c.vec(), which the caller fills with the Bloch vector anyway.
  __device__ __forceinline__ static cd wht_step(cd s, cd p, bool upper) {
    return upper ? cd{p.re - s.re, p.im - s.im} : cd{s.re + p.re, s.im + p.im};
  }
  __device__ static double bloch_of(const Ctx& c, const cd* m) {
    if constexpr (NQ == 3) {
      const int x = c.i, r = c.j, lane = threadIdx.x & 63;
      const cd e = m[r * LD + (r ^ x)];
      cd s{e.re, -e.im};
      s = wht_step(s, cd{dpp_f64<0xB1>(s.re), dpp_f64<0xB1>(s.im)}, lane & 1);
      s = wht_step(s, cd{dpp_f64<0x4E>(s.re), dpp_f64<0x4E>(s.im)}, lane & 2);
      s = wht_step(s, cd{__shfl_xor(s.re, 4), __shfl_xor(s.im, 4)}, lane & 4);
      const int z = r;
      const double v = re_phase(__popc(x & z) & 3, s) / d;
      // pauli_index(x, z): digit bits (hi, lo) = (z_q, x_q ^ z_q) = spread(x) ^ 3 spread(z)
      const int sx = (x & 1) | ((x & 2) << 1) | ((x & 4) << 2), sz = (z & 1) | ((z & 2) << 1) | ((z & 4) << 2);
      double* vec = c.vec();
      vec[sx ^ (3 * sz)] = v;
      wave_sync();
      const double mine = vec[c.l];
      wave_sync();
      return mine;
    }
    cd s{0.0, 0.0};
#pragma unroll
    for (int r = 0; r < d; ++r) {
      const cd e = m[r * LD + (r ^ c.xm)];  // conj(M[r][r^x]) summed with the sign of P_k[r][r^x]
      const double sg = (__popc(r & c.zm) & 1) ? -1.0 : 1.0;
      s.re += sg * e.re;
      s.im -= sg * e.im;
    }
    return re_phase(c.ny, s) / d;
  }

  // Element (i, j) of sum_k v[k] P_k, v in LDS (qobj.py:114-117).
  // Element (i, j) of sum_k v[k] P_k needs d of the D terms: k = pauli_index(i ^ j, z), z < d, each
  // with phase (-i)^popc(x & z) (-1)^popc(i & z).  Which k, real or imaginary, which sign: one byte
  // per z (k | imaginary << 6 | negative << 7), a pure function of the lane, folded at compile time
  // into a 64-entry table -- the run-time version spent ~25 integer instructions per term on it.
  static constexpr uint64_t mat_pack(int l) {
    const int i = l / d, j = l % d, x = i ^ j;
    uint64_t pack = 0;
    for (int z = 0; z < d; ++z) {
      int k = 0, ny = 0, par = 0;
      for (int b = 0; b < NQ; ++b) {
        const int xb = (x >> b) & 1, zb = (z >> b) & 1;
        const int dig = xb ? (zb ? 2 : 1) : (zb ? 3 : 0);
        k |= dig << (2 * b);
        ny += xb & zb;
        par ^= ((i >> b) & 1) & zb;
      }
      ny &= 3;
      const int imag = ny & 1;                       // (-i)^ny: 1, -i, -1, i
      const int neg = ((ny == 1 || ny == 2) ? 1 : 0) ^ par;
      pack |= (uint64_t)(k | (imag << 6) | (neg << 7)) << (8 * z);
    }
    return pack;
  }
  template <int... L>
  struct MatTab {
    static constexpr uint64_t v[sizeof...(L)] = {mat_pack(L)...};
  };
  template <int N, int... L>
  struct MakeMatTab : MakeMatTab<N - 1, N - 1, L...> {};
  template <int... L>
  struct MakeMatTab<0, L...> {
    using type = MatTab<L...>;
  };
  static_assert(NQ <= 3, "the packed table holds d <= 8 terms of 8 bits");
  __device__ static cd matrix_of(const Ctx& c, const double* v) {
    using Tab = typename MakeMatTab<D>::type;
    const uint64_t pack = Tab::v[c.l];
    cd s{0.0, 0.0};
#pragma unroll
    for (int z = 0; z < d; ++z) {
      const uint32_t e = (uint32_t)(pack >> (8 * z)) & 0xffu;
      double val = v[e & 63u];
      val = (e & 0x80u) ? -val : val;
      if (e & 0x40u) s.im += val;
      else s.re += val;
    }
    return s;
  }

  // counts of this trial -> freq[] in LDS (counts / sum(counts): state.py:193, :227).  For a product
  // POVM the frequencies are stored in R-order and the one-qubit tables are staged next to them.
  // The counts are the only HBM read of a trial.  Issued at the very top of the kernel (before the
  // workgroup stages its tables and meets at the barrier of make_ctx) their latency runs in the shadow
  // of that set-up; up to 4 values per lane are held in registers until load_freq picks them up.
  struct Prefetch {
    int64_t v[4];
    double ns[4];  // N_s of the settings rows l + q G belong to (segmented check) or of setting s = lane (ns[0])
    bool ok, seg;  // seg: K in {2, 4, 8, 16} divides G -- the K outcomes of a setting sit in K neighbouring lanes
  };
  __device__ __forceinline__ static void prefetch_counts(Prefetch& pf, const int64_t* counts, const PovmView& pv) {
    const int l = (threadIdx.x & 63) % G, M = pv.M, K = pv.K;
    pf.ok = M <= 4 * G;
    pf.seg = pf.ok && pv.Ns && (K == 2 || K == 4 || K == 8 || K == 16) && G % K == 0;
    const int lgk = __builtin_ctz((unsigned)(K > 0 ? K : 1));  // K is a power of two on the seg path: m / K is a shift
    if (pf.ok) {                                               // (a 32-bit division by a run-time K is ~25 instructions)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = l + q * G;
        pf.v[q] = counts[m < M ? m : M - 1];
        if (pf.seg) pf.ns[q] = pv.Ns[(m < M ? m : M - 1) >> lgk];
      }
    }
    if (!pf.seg) pf.ns[0] = pv.Ns ? pv.Ns[l < pv.S ? l : pv.S - 1] : 0.0;
  }
  // sum over the aligned group of KK neighbouring lanes, KK a compile-time constant (no branch per step)
  template <int KK>
  __device__ __forceinline__ static double segsum_c(double v) {
    v += dpp_f64<0xB1>(v);
    if (KK >= 4) v += dpp_f64<0x4E>(v);
    if (KK >= 8) v += dpp_f64<0x141>(v);
    if (KK >= 16) v += dpp_f64<0x140>(v);
    return v;
  }
  template <int KK>
  __device__ __forceinline__ static bool seg_check(const double (&dv)[4], const bool (&valid)[4], double total,
                                                   const Prefetch& pf, double ns_tot) {
    bool bad = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) bad |= valid[q] & !shots_match(segsum_c<KK>(dv[q]), total, pf.ns[q], ns_tot);
    return bad;
  }
  __device__ static bool load_freq(const Ctx& c, const int64_t* counts, const Prefetch* pf = nullptr) {
    // one coalesced pass over the counts: values parked in rbuf
    double part = 0.0;
    double* raw = c.rbuf();
    double* fq = c.freq();
    bool bad = false;
    double total, inv;
    if (pf && pf->ok) {
      // The prefetched path (M <= 4 G: every built-in POVM at n <= 3) is written WITHOUT branches: out-of-range rows
      // carry 0.0 and go to one dummy slot behind raw[] (the first word of freq[], overwritten below), the frequency
      // pass re-writes row M - 1 from the lanes beyond it, and the shots check is a compile-time butterfly per K.  As
      // `if (m < M)` blocks and a run-time K it compiled to ~25 scalar branches on the critical path of a lone wave.
      double dv[4];
      bool valid[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = c.l + q * G;
        valid[q] = m < c.M;
        dv[q] = valid[q] ? (double)pf->v[q] : 0.0;
        raw[valid[q] ? m : c.Mp] = dv[q];
        part += dv[q];
      }
      total = gsum<G>(part);
      inv = 1.0 / total;
      if (pf->seg) {  // (uniform) the K outcomes of a setting are K neighbouring lanes of one prefetched value
        switch (c.pv.K) {
          case 2: bad = seg_check<2>(dv, valid, total, *pf, c.pv.ns_tot); break;
          case 4: bad = seg_check<4>(dv, valid, total, *pf, c.pv.ns_tot); break;
          case 8: bad = seg_check<8>(dv, valid, total, *pf, c.pv.ns_tot); break;
          default: bad = seg_check<16>(dv, valid, total, *pf, c.pv.ns_tot); break;
        }
      }
      wave_sync();
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = c.l + q * G, mc = m < c.M ? m : c.M - 1;
        fq[mc] = raw[c.prod() ? c.trmap[mc] : mc] * inv;
      }
    } else {
      for (int m = c.l; m < c.M; m += G) {
        const double v = (double)counts[m];
        raw[m] = v;
        part += v;
      }
      total = gsum<G>(part);
      inv = 1.0 / total;
      wave_sync();
      if (c.prod()) {
        for (int m = c.l; m < c.M; m += G) fq[m] = raw[c.trmap[m]] * inv;
      } else {
        for (int m = c.l; m < c.M; m += G) fq[m] = raw[m] * inv;
      }
    }
    // shots check (state.py:138-141, 194-197) for the shapes the butterfly does not cover: lane s sums the K outcomes
    // of setting s (raw is in the caller's (S, K) order)
    if (c.pv.Ns && !(pf && pf->seg)) {
      const int K = c.pv.K;
      for (int s = c.l; s < c.pv.S; s += G) {
        double t = 0.0;
        for (int k = 0; k < K; ++k) t += raw[s * K + k];
        const double ns = (pf && s == c.l) ? pf->ns[0] : c.pv.Ns[s];
        bad |= !shots_match(t, total, ns, c.pv.ns_tot);
      }
    }
    const unsigned long long votes = __builtin_amdgcn_ballot_w64(bad);
    const int lane = threadIdx.x & 63;
    const unsigned long long mine = (G == 64) ? ~0ull : (((1ull << (G & 63)) - 1ull) << (lane / G * G));
    wave_sync();
    return (votes & mine) == 0ull;
  }

  // ---- factorised contractions for product POVMs -----------------------------------------------
  __device__ __forceinline__ static int ipow(int b, int e) {
    int r = 1;
    for (int q = 0; q < e; ++q) r *= b;
    return r;
  }
  // One stage: out[o] = sum_t tbl(sel, t) * in[base + t * stride], (base, sel) = tab[o].
  // FWD: 4 terms, tbl(sel, t) = tb[sel*4 + t] (row `sel` of the table);  backward: R1 terms,
  // tbl(sel, t) = tb[t*4 + sel] (column `sel`).
  // One output of a contraction stage.  R1C > 0 fixes the table height at compile time (backward
  // stages sum R1 terms): the R1 pairs of LDS reads are then issued together instead of one
  // read-wait-FMA round trip per term, which is what a run-time trip count compiles to.
  template <bool FWD, int R1C = 0>
  __device__ __forceinline__ static double stage_value(const double* tb, int R1, int ent, int stride, const double* in) {
    const int base = ent & 0xffff, sel = ent >> 16;
    if (FWD) {
      const double* row = tb + sel * 4;
      return fma(row[3], in[base + 3 * stride],
                 fma(row[2], in[base + 2 * stride], fma(row[1], in[base + stride], row[0] * in[base])));
    }
    double acc = 0.0;
    if (R1C > 0) {
      double tv[R1C > 0 ? R1C : 1], iv[R1C > 0 ? R1C : 1];
#pragma unroll
      for (int t = 0; t < R1C; ++t) {
        tv[t] = tb[t * 4 + sel];
        iv[t] = in[base + t * stride];
      }
#pragma unroll
      for (int t = 0; t < R1C; ++t) acc = fma(tv[t], iv[t], acc);
      return acc;
    }
    for (int t = 0; t < R1; ++t) acc = fma(tb[t * 4 + sel], in[base + t * stride], acc);
    return acc;
  }
  // All outputs of a stage, two per lane and pass so that their reads overlap.
  template <bool FWD, int R1C>
  __device__ __forceinline__ static void stage_run(const Ctx& c, const double* tb, const int* tab, int n_out, int stride,
                                                   const double* in, double* out) {
    const int R1 = c.pv.pr.R1;
    for (int o = c.l; o < n_out; o += 2 * G) {
      const int o2 = o + G;
      const bool two = o2 < n_out;
      const int e0 = tab[o], e1 = tab[two ? o2 : o];
      const double v0 = stage_value<FWD, R1C>(tb, R1, e0, stride, in);
      const double v1 = stage_value<FWD, R1C>(tb, R1, e1, stride, in);
      out[o] = v0;
      if (two) out[o2] = v1;
    }
  }
  template <bool FWD>
  __device__ static void stage(const Ctx& c, const double* tb, const int* tab, int n_out, int stride, const double* in,
                               double* out) {
    if (FWD) stage_run<true, 0>(c, tb, tab, n_out, stride, in, out);
    else if (c.pv.pr.R1 == 6) stage_run<false, 6>(c, tb, tab, n_out, stride, in, out);
    else if (c.pv.pr.R1 == 4) stage_run<false, 4>(c, tb, tab, n_out, stride, in, out);
    else stage_run<false, 0>(c, tb, tab, n_out, stride, in, out);
    wave_sync();
  }
  template <int R1C>
  __device__ __forceinline__ static double stage_last(const Ctx& c, const double* tb, const int* tab, const double* in) {
    return stage_value<false, R1C>(tb, c.pv.pr.R1, tab[c.l], 1 << (2 * (NQ - 1)), in);
  }
  // Backward pass from Y_n (R-order, in `yn`) to the lane's Y_0[k = l]; tb = tabT (A^T y) or tabP (A^+ f).
  // Intermediates ping-pong between bufB and rbuf; `yn` itself is only read.
  __device__ static double prod_backward(const Ctx& c, const double* tb, const double* yn) {
    const int R1 = c.pv.pr.R1;
    const int* tab = c.tbwd;
    const double* in = yn;
    double* bufs[2] = {c.bufB(), c.rbuf()};
    int which = 0;
#pragma unroll
    for (int q = NQ; q >= 2; --q) {
      const int stride = 1 << (2 * (NQ - q));             // 4^(n-q)
      const int n_out = ipow(R1, q - 1) * 4 * stride;
      stage<false>(c, tb, tab, n_out, stride, in, bufs[which]);
      tab += n_out;
      in = bufs[which];
      which ^= 1;
    }
    // stage 1: D outputs, one per lane
    if (R1 == 6) return stage_last<6>(c, tb, tab, in);
    if (R1 == 4) return stage_last<4>(c, tb, tab, in);
    return stage_last<0>(c, tb, tab, in);
  }
  // sum_m Op[m][lane] * vec[m]   (lane = column)
  __device__ __forceinline__ static double col_dot(const Ctx& c, const double* g_rowmajor, const double* vec) {
    if (ALDS) return dot_lds(c.img, LDA, c.l, vec, c.M);
    return dot_global<16>(g_rowmajor, D, (unsigned)c.l, vec, c.M);
  }
  // sum_k Op[row][k] * vec[k]    (lane = row), g_transposed = [D][M]
  __device__ __forceinline__ static double row_dot(const Ctx& c, const double* g_transposed, int row, const double* vec) {
    if (ALDS) return dot_lds(c.img + row * LDA, 1, 0, vec, D);
    return dot_global<(D < 16 ? D : 16)>(g_transposed, (size_t)c.M, (unsigned)row, vec, D);
  }

  // ---- a6: linear inversion (image = PinvT).  Returns lane's element of rho; vec() = Bloch vector.
  __device__ static cd lin_invert(const Ctx& c, double& bloch_l) {
    if (c.prod()) {
      if (c.pv.pr.uniform) {
        // pinv(w K) = pinv(T)^(x n) / w : the same backward pass with pinv(T) instead of T
        bloch_l = prod_backward(c, c.tabP(), c.freq()) / (c.pv.pr.wuni * d);
      } else {  // unequal shots per setting: dense left inverse, rows visited in R-order
        double acc = 0.0;
        for (int m = 0; m < c.M; ++m) acc = fma(c.pv.PinvT[(size_t)c.trmap[m] * D + c.l], c.freq()[m], acc);
        bloch_l = acc / d;
      }
    } else {
      bloch_l = col_dot(c, c.pv.PinvT, c.freq()) / d;  // bloch_k = sum_m Pinv[k][m] f_m / d
    }
    c.vec()[c.l] = bloch_l;
    wave_sync();
    cd r = matrix_of(c, c.vec());
    wave_sync();
    return r;
  }

  // ---- a7: eigenvalue clip + trace renormalisation by a parallel-order cyclic Jacobi.
  // In: lane's element of a Hermitian matrix.  Out: lane's element of U max(v, eps) U^dagger / Tr.
  // Round r = 1 .. d-1 rotates the d/2 disjoint pairs (k, k ^ r): every pair once per sweep.  A
  // round is ONE LDS round trip: the lanes publish A and V, then each lane reads the two 2x2
  // blocks that define the rotations of its column pair {j, j^r} and row pair {i, i^r} together
  // with its three partner elements, and applies A' = J^dagger A J, V' = V J in registers.
  __device__ static cd psd_project(const Ctx& c, cd a, double eps) {
    const int i = c.i, j = c.j;
    cd* Ai = c.A();
    cd* Vi = c.V();
    cd v{i == j ? 1.0 : 0.0, 0.0};
    if (i == j) a.im = 0.0;
    const double nrm = gsum<G>(a.re * a.re + a.im * a.im);  // Frobenius norm: invariant
    for (int sweep = 0; sweep < 20; ++sweep) {
      const double off = gsum<G>(i != j ? a.re * a.re + a.im * a.im : 0.0);
      if (__all(!(off > c.pv.jtol2 * nrm))) break;
#pragma unroll 1
      for (int r = 1; r < d; ++r) {
        Ai[c.e] = a;
        Vi[c.e] = v;
        wave_sync();
        const int pj = j ^ r, pi = i ^ r;
        const int cp = j < pj ? j : pj, cq = j < pj ? pj : j;  // column pair, ordered
        const int rp = i < pi ? i : pi, rq = i < pi ? pi : i;  // row pair, ordered
        const double c_pp = Ai[cp * LD + cp].re, c_qq = Ai[cq * LD + cq].re;
        const cd c_pq = Ai[cp * LD + cq];
        const double r_pp = Ai[rp * LD + rp].re, r_qq = Ai[rq * LD + rq].re;
        const cd r_pq = Ai[rp * LD + rq];
        const cd a_c = Ai[i * LD + pj], a_r = Ai[pi * LD + j], a_x = Ai[pi * LD + pj];
        const cd v_c = Vi[i * LD + pj];
        wave_sync();  // all reads of this image are issued before the next round overwrites it
        double cj, ci;
        cd wj, wi;
        rotation(c_pp, c_qq, c_pq, cj, wj);
        rotation(r_pp, r_qq, r_pq, ci, wi);
        if (j < pj) wj = cd{-wj.re, wj.im};  // J[pj][j]: -conj(w) if j is the lower index, else w
        if (i < pi) wi = cd{-wi.re, wi.im};
        // A'_ij = ci (a_ij cj + a_i,pj wj) + conj(wi) (a_pi,j cj + a_pi,pj wj) ;  V' = V J
        const cd t0 = cadd(cscale(a, cj), cmul(a_c, wj));
        const cd t1 = cadd(cscale(a_r, cj), cmul(a_x, wj));
        a = cadd(cscale(t0, ci), cmulc(t1, wi));
        v = cadd(cscale(v, cj), cmul(v_c, wj));
        if (i == j) a.im = 0.0;
      }
    }
    // rebuild with clipped eigenvalues: R_ij = sum_k V_ik max(lam_k, eps) conj(V_jk)
    double* lam = c.lam();
    Vi[c.e] = v;
    if (i == j) lam[i] = a.re;
    wave_sync();
    cd rr{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < d; ++k) {
      const double lc = lam[k] > eps ? lam[k] : eps;  // np.maximum(eps, v)
      const cd p = cmulc(Vi[i * LD + k], Vi[j * LD + k]);
      rr.re += lc * p.re;
      rr.im += lc * p.im;
    }
    const double tr = gsum<G>(i == j ? rr.re : 0.0);
    wave_sync();
    return cd{rr.re / tr, rr.im / tr};
  }

  // ---- a8: lower Cholesky factor of the matrix whose element this lane holds.
  // Leaves L in Bm() (upper part zero) and returns the lane's parameter x_l; ok = 0 if not PD.
  // The elimination is carried on past a non-positive pivot as L S L^dagger (S = signs), so that by
  // Sylvester's law `neg` is the number of negative eigenvalues (99 when a pivot is too small for
  // the count to mean anything) and `kneg` the first index where the leading block stops being PD.
  // SPEC (latency-bound launches only: one wave per SIMD, the fused kernel): the Gauss-Jordan inverse that
  // lift_single_negative needs when the LAST pivot is the negative one (kneg = d - 1, the generic case for one small
  // negative eigenvalue: natural pivot order) runs in the same sweep, step k of it beside step k of the elimination --
  // two independent dependency chains sharing the LDS round trips.  A trial that turns out not positive definite then
  // skips ~3 k of its ~39 k clocks (the launch ends with its slowest wave, and those are the clipped trials); a
  // positive definite one throws the inverse away, ~1.5 k clocks it had to spare.  Same arithmetic, same bits, as the
  // loop in lift_single_negative.  *spec_inv = this lane's element of the inverse (garbage unless kneg = d - 1).
  template <bool SPEC = false>
  __device__ static double cholesky_param(const Ctx& c, cd a, int& ok, int* neg_out = nullptr, int* kneg_out = nullptr,
                                          cd* spec_inv = nullptr) {
    cd* A = c.A();
    cd* L = c.Bm();
    cd* Vs = c.V();
    const int i = c.i, j = c.j;
    A[c.e] = a;
    int neg = 0, kneg = d - 1;
    cd lmine{0.0, 0.0};  // this lane's element of L (its column j is final after step k = j)
    cd b = a;            // SPEC: the matrix on its way to its inverse
    if (SPEC && i == j) b.im = 0.0;
    if (SPEC) Vs[c.e] = b;
    wave_sync();
#pragma unroll
    for (int k = 0; k < d; ++k) {
      // The pivot sits in the register of lane (k, k): with one trial per wave it is taken from there
      // (v_readlane), so that 1/sqrt runs while the column is still on its way through LDS.
      const double akk = (G == 64) ? readlane_f64(a.re, k * d + k) : A[k * LD + k].re;
      const cd aik = A[i * LD + k], ajk = A[j * LD + k];
      if constexpr (SPEC) {  // step p = k of the inverse (lift_single_negative's loop body)
        const double piv = readlane_f64(b.re, k * d + k);
        const cd bip = Vs[i * LD + k], bpj = Vs[k * LD + j];
        double inv = __builtin_amdgcn_rcp(piv);
        inv = fma(inv, fma(-piv, inv, 1.0), inv);
        inv = fma(inv, fma(-piv, inv, 1.0), inv);
        const cd t{bpj.re * inv, bpj.im * inv};
        const cd e = cmul(bip, t);
        cd nb{b.re - e.re, b.im - e.im};
        if (j == k) nb = cd{-bip.re * inv, -bip.im * inv};
        if (i == k) nb = (j == k) ? cd{inv, 0.0} : t;
        b = nb;
        if (k + 1 < d) Vs[c.e] = b;  // published by the elimination's wave_sync below
      }
      const double mag = fabs(akk);
      const bool pos = akk > 0.0;
      if (!pos) {
        if (neg == 0) kneg = k;
        ++neg;
      }
      if (!(mag > 1e-13)) neg = neg > 0 ? 99 : neg;  // NaN lands here too
      // In the sweep only 1 / a_kk is needed (a_ij -= s a_ik conj(a_jk) / |a_kk|); the square roots that turn
      // the columns into L are taken once, for all columns in parallel, after the sweep.
      const double safe = mag > 1e-300 ? mag : 1e-300;
      if (k + 1 < d) {
        const double inv = recip_nr(safe);
        const double sinv = pos ? -inv : inv;
        const cd ajs{ajk.re * sinv, ajk.im * sinv};  // -s a_jk / |a_kk|
        const bool upd = i > k && j > k;
        const double nre = fma(aik.re, ajs.re, fma(aik.im, ajs.im, a.re));
        const double nim = fma(aik.im, ajs.re, fma(-aik.re, ajs.im, a.im));
        a.re = upd ? nre : a.re;
        a.im = upd ? nim : a.im;
        A[c.e] = a;  // branch-free: lanes outside the trailing block store their old value again
        wave_sync();
      }
    }
    {
      // lane (i, j), i >= j: its register still holds a_ij as it was when column j became the pivot column
      const double dj = A[j * LD + j].re;
      const double mj = fabs(dj);
      const double rs = fast_rsqrt(mj > 1e-300 ? mj : 1e-300);  // 1 / |l_jj|
      if (i >= j) lmine = (i == j) ? cd{dj * rs, 0.0} : cd{a.re * rs, a.im * rs};
    }
    ok = neg == 0;
    if (neg_out) *neg_out = neg;
    if (kneg_out) *kneg_out = kneg;
    if (SPEC && spec_inv) *spec_inv = b;
    L[c.e] = lmine;
    wave_sync();
    const cd e = L[c.pi * LD + c.pj];
    const double x = c.pkind == 2 ? e.im : e.re;
    wave_sync();
    return x;
  }

  // ---- a7, short cut for exactly ONE negative eigenvalue (lam_1 < 0 < lam_2 <= ...):
  //   U max(v, eps) U^dagger = A + (eps - lam_1) v_1 v_1^dagger,
  // so only the projector N = v_1 v_1^dagger is needed.  M = A^{-1} (Gauss-Jordan, pivot order with
  // `kneg` last: the leading block is positive definite, so no pivoting is needed) has v_1 as its
  // dominant direction whenever |lam_1| < lam_2; N <- N N^dagger / Tr squares the separation
  // ratio each time (k squarings = 2^k inverse-iteration steps at the cost of k 8x8 products).
  // Certified, not assumed: purity Tr N^2 -> 1 (then one more squaring: contamination < 1e-14),
  // lam_1 = Tr(A N) < eps, and ||A N - lam_1 N||_F <= 1e-13 ||A||_F.  Returns false (caller runs the
  // Jacobi eigensolver on the untouched input) on a slow ratio, a positive lam, or a failed check.
  // `inverse`: the lane's element of A^{-1} when the caller already has it (cholesky_param<true>, kneg = d - 1).
  __device__ static bool lift_single_negative(const Ctx& c, cd r, int kneg, double eps, cd& out,
                                              const cd* inverse = nullptr) {
    static_assert(G == 64, "one trial per wavefront: the flags below are wave-uniform");
    cd* Ai = c.A();
    cd* Vi = c.V();
    const int i = c.i, j = c.j;
    if (i == j) r.im = 0.0;
    cd b = r;
    if (inverse) b = *inverse;
#pragma unroll 1
    for (int s = inverse ? d : 0; s < d; ++s) {
      const int p = s < kneg ? s : (s < d - 1 ? s + 1 : kneg);
      Ai[c.e] = b;
      const double piv = readlane_f64(b.re, p * d + p);  // from lane (p, p): no LDS wait before 1 / pivot
      wave_sync();
      const cd bip = Ai[i * LD + p], bpj = Ai[p * LD + j];
      wave_sync();
      double inv = __builtin_amdgcn_rcp(piv);
      inv = fma(inv, fma(-piv, inv, 1.0), inv);
      inv = fma(inv, fma(-piv, inv, 1.0), inv);
      const cd t{bpj.re * inv, bpj.im * inv};
      const cd e = cmul(bip, t);
      cd nb{b.re - e.re, b.im - e.im};
      if (j == p) nb = cd{-bip.re * inv, -bip.im * inv};
      if (i == p) nb = (j == p) ? cd{inv, 0.0} : t;
      b = nb;
    }
    QT_STAMP(4);
    // X_k is kept at Frobenius norm 1, i.e. sum sigma_i^2 = 1; then ||X_k X_k^dagger||_F^2 = sum sigma_i^4
    // is the purity of that spectrum: one reduction per squaring gives both the scale and the test.
    b = cscale(b, fast_rsqrt(gsum<G>(b.re * b.re + b.im * b.im)));
    bool certified = false, done = false;
#pragma unroll 1
    for (int k = 1; k <= 7 && !done; ++k) {
      Ai[c.e] = b;
      wave_sync();
      cd n{0.0, 0.0}, n2{0.0, 0.0};  // even / odd terms: two dependency chains per component
#pragma unroll
      for (int q = 0; q < d; q += 2) {
        const cd u = Ai[i * LD + q], v = Ai[j * LD + q];
        n.re = fma(u.re, v.re, fma(u.im, v.im, n.re));
        n.im = fma(u.im, v.re, fma(-u.re, v.im, n.im));
        if (q + 1 < d) {
          const cd u2 = Ai[i * LD + q + 1], v2 = Ai[j * LD + q + 1];
          n2.re = fma(u2.re, v2.re, fma(u2.im, v2.im, n2.re));
          n2.im = fma(u2.im, v2.re, fma(-u2.re, v2.im, n2.im));
        }
      }
      n = cadd(n, n2);
      wave_sync();
      const double pur = gsum<G>(n.re * n.re + n.im * n.im);
      b = cscale(n, fast_rsqrt(pur));
      const double defect = 1.0 - pur;  // ~ 2 (sigma_2 / sigma_1)^(2^(k+1))
      QT_STAMP_VAL(20, k);
      if (certified) done = true;
      else if (defect < 1e-7) certified = true;
      else if (k == 3 && !(defect < 0.05)) return false;  // ratio > ~0.8: the eigensolver is cheaper
    }
    QT_STAMP(5);
    if (!done) return false;
    if (i == j) b.im = 0.0;
    b = cscale(b, 1.0 / gsum<G>(i == j ? b.re : 0.0));      // N = X / Tr X
    const double lam = gsum<G>(r.re * b.re + r.im * b.im);  // Tr(A N) = sum_ij A_ij conj(N_ij)
    if (!(lam < eps)) return false;
    Ai[c.e] = r;
    Vi[c.e] = b;
    wave_sync();
    cd q{-lam * b.re, -lam * b.im};
#pragma unroll
    for (int k = 0; k < d; ++k) {  // (A N)_ij, N Hermitian
      const cd u = Ai[i * LD + k], v = Vi[j * LD + k];
      q.re = fma(u.re, v.re, fma(u.im, v.im, q.re));
      q.im = fma(u.im, v.re, fma(-u.re, v.im, q.im));
    }
    wave_sync();
    const double res2 = gsum<G>(q.re * q.re + q.im * q.im);
    const double nrm2 = gsum<G>(r.re * r.re + r.im * r.im);
    if (!(res2 <= 1e-26 * nrm2)) return false;
    const double w = eps - lam;
    cd o{fma(w, b.re, r.re), fma(w, b.im, r.im)};
    const double tr = gsum<G>(i == j ? o.re : 0.0);
    out = cd{o.re / tr, o.im / tr};
    return true;
  }

  // a7 with the positive-definite shortcut: when the Hermitian input is numerically positive
  // definite (its Cholesky factorisation runs through) no eigenvalue is below the clip, so
  // U max(v, 1e-15) U^dagger is the input itself (to rounding) and only the trace division is left.
  // Returns the projected element; if `xl` is non-null also the Cholesky parameter of the result.
  template <bool SPEC = false>
  __device__ static cd make_feasible(const Ctx& c, cd r, double* xl, int* ok_out, double* lscale_out = nullptr) {
    int ok, neg, kneg;
    cd spec_inv{0.0, 0.0};
    double x = cholesky_param<SPEC && G == 64>(c, r, ok, &neg, &kneg, &spec_inv);
    QT_STAMP(3);
    const double tr = gsum<G>(c.i == c.j ? r.re : 0.0);
    cd out{r.re / tr, r.im / tr};
    // 1 / sqrt(tr) for the Cholesky parameters of r / tr: hardware seed + Newton (~1 ulp) instead of an IEEE sqrt and
    // an IEEE division (~60 instructions on the critical path); tr = 1 up to the noise of the linear inversion
    const double rst = tr > 0.0 ? fast_rsqrt(tr) : 1.0 / sqrt(tr);
    x = x * rst;  // L of r/tr
    double lscale = rst;  // Bm() holds the factor of r, not of r/tr
    if (!__all(ok)) {
      cd proj;
      bool lifted = false;
      if constexpr (G == 64) {  // one trial per wave: neg / kneg are the same in every lane
        if (kLiftSingleNegative && __builtin_amdgcn_readfirstlane(neg) == 1) {
          const int kn = __builtin_amdgcn_readfirstlane(kneg);
          lifted = __builtin_amdgcn_readfirstlane(
                       (int)lift_single_negative(c, r, kn, 1e-15, proj, SPEC && kn == d - 1 ? &spec_inv : nullptr)) != 0;
        }
      }
      if (!lifted) proj = psd_project(c, r, 1e-15);  // whole wave runs it; PD trials keep their shortcut
      QT_STAMP(6);
      int ok2 = 1;
      double x2 = xl ? cholesky_param(c, proj, ok2) : 0.0;
      if (lifted && xl && !__all(ok2 || ok)) {
        // the short cut left something non-positive behind (an inertia count thrown off by a pivot at
        // rounding level): take the eigensolver after all
        proj = psd_project(c, r, 1e-15);
        x2 = cholesky_param(c, proj, ok2);
      }
      QT_STAMP(7);
      if (xl) lscale = 1.0;  // the whole wave went through the second factorisation: Bm() = factor of proj
      if (!ok) {
        out = proj;
        x = x2;
        ok = ok2;
      }
    }
    if (xl) *xl = x;
    if (ok_out) *ok_out = ok;
    if (lscale_out) *lscale_out = lscale;
    return out;
  }

  // x (one parameter per lane) -> L in Bm(), returns lane's element of L L^dagger and t = Tr.
  __device__ static cd build_llh(const Ctx& c, double xl, double& tr) {
    double* vx = c.vec();
    cd* L = c.Bm();
    vx[c.l] = xl;
    tr = gsum<G>(xl * xl);
    wave_sync();
    const int i = c.i, j = c.j;
    L[c.e] = cd{vx[c.src_re], vx[c.src_im]};  // branch-free: the zero slot feeds the upper triangle
    wave_sync();
    cd m{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < d; ++k) {  // all k: the zeros of L contribute exact zeros
      const cd u = L[i * LD + k], v = L[j * LD + k];
      m.re = fma(u.re, v.re, fma(u.im, v.im, m.re));
      m.im = fma(u.im, v.re, fma(-u.re, v.im, m.im));
    }
    return m;
  }

  // ---- a9: NLL value and exact gradient at x (image = Aw).  Needs freq[] loaded.  Leaves L in Bm().
  // `start` (first evaluation of a trial, at the Cholesky parameters of a matrix that is still at hand):
  // rho(x) = L L^dagger / Tr is that matrix, and its factor already sits in Bm() up to the scale
  // start->lscale, so L L^dagger is not formed again.
  struct StartPoint {
    cd rho;         // this lane's element of the normalised start matrix
    double lscale;  // L(x) = lscale * Bm()
  };
  __device__ static void nll_grad(const Ctx& c, double xl, double& f, double& gl, cd* rho_l = nullptr,
                                  bool want_grad = true, const StartPoint* start = nullptr) {
    double tr;
    QT_STAMP(11);
    cd m;
    if (start) {
      tr = gsum<G>(xl * xl);
      m = cd{start->rho.re * tr, start->rho.im * tr};
    } else {
      m = build_llh(c, xl, tr);
    }
    QT_STAMP(12);
    cd* A = c.A();
    const cd rho_e = start ? start->rho : cd{m.re / tr, m.im / tr};
    if (rho_l) *rho_l = rho_e;
    A[c.e] = rho_e;
    wave_sync();
    const double bl = bloch_of(c, A);
    QT_STAMP(13);
    double* vec = c.vec();
    vec[c.l] = bl;
    wave_sync();
    // p = d * A' b ;  f = -sum freq log(p + 1e-10) ;  r = freq / (p + 1e-10)
    double fpart = 0.0;
    const double* fr = c.freq();
    double* rb = c.rbuf();
    double wl;
    if (c.prod()) {
      const int R1 = c.pv.pr.R1;
      const int* tab = c.tfwd;
      const double* in = vec;
#pragma unroll
      for (int q = 1; q < NQ; ++q) {  // stages 1 .. n-1; the last of them lands in bufB
        const int stride = 1 << (2 * (NQ - q));
        const int n_out = ipow(R1, q) * stride;
        double* out = ((NQ - 1 - q) & 1) ? rb : c.bufB();
        stage<true>(c, c.tabT(), tab, n_out, stride, in, out);
        tab += n_out;
        in = out;
      }
      QT_STAMP(14);
      // stage n fused with the log-likelihood terms, two outputs per pass so that their LDS round trips
      // (table entry -> operands) overlap; the second of a pair is a recomputation of the first when the
      // row count runs out, and is not stored
      for (int o = c.l; o < c.M; o += 2 * G) {
        const int o2 = o + G;
        const bool two = o2 < c.M;
        const int oo = two ? o2 : o;
        const int e0 = tab[o], e1 = tab[oo];
        const double x0 = stage_value<true>(c.tabT(), R1, e0, 1, in);
        const double x1 = stage_value<true>(c.tabT(), R1, e1, 1, in);
        const double w0 = c.twrow[o], w1 = c.twrow[oo];
        const double f0 = fr[o], f1 = fr[oo];
        const double p0 = x0 * w0 * d + 1e-10, p1 = x1 * w1 * d + 1e-10;
        const double l0 = fast_log(p0), l1 = fast_log(p1);
        fpart += f0 * l0;
        rb[o] = w0 * f0 * recip_nr(p0);  // Y_n = w (.) r : A'^T r = K^T (w (.) r)
        if (two) {
          fpart += f1 * l1;
          rb[o2] = w1 * f1 * recip_nr(p1);
        }
      }
      f = -gsum<G>(fpart);
      wave_sync();
      QT_STAMP(15);
      if (!want_grad) return;  // (uniform) the Metropolis chain only needs the value
      wl = prod_backward(c, c.tabT(), rb);
      QT_STAMP(16);
    } else {
      for (int m0 = 0; m0 < c.M; m0 += G) {
        const int mm = m0 + c.l;
        if (mm < c.M) {
          const double pe = row_dot(c, c.pv.AwT, mm, vec) * d + 1e-10;
          fpart += fr[mm] * fast_log(pe);
          rb[mm] = fr[mm] * recip_nr(pe);
        }
      }
      f = -gsum<G>(fpart);
      wave_sync();
      if (!want_grad) return;
      // w = A'^T r ;  G = -sum_k w_k P_k ;  Gt = (G - Tr(G rho) I) / t
      wl = col_dot(c, c.pv.Aw, rb);
    }
    const double tr_g_rho = -(double)d * gsum<G>(wl * bl);
    vec[c.l] = wl;
    wave_sync();
    cd g = matrix_of(c, vec);
    QT_STAMP(17);
    g.re = -g.re;
    g.im = -g.im;
    if (c.i == c.j) g.re -= tr_g_rho;
    g.re /= tr;
    g.im /= tr;
    A[c.e] = g;
    wave_sync();
    // Q = Gt L ; gradient entries 2 Re Q_ii, 2 Re Q_ij, 2 Im Q_ij (i > j)
    const cd* L = c.Bm();
    cd q{0.0, 0.0};
#pragma unroll
    for (int k = 0; k < d; ++k) {  // all k: L[k][pj] = 0 above the diagonal
      const cd u = A[c.pi * LD + k], v = L[k * LD + c.pj];
      q.re = fma(u.re, v.re, fma(-u.im, v.im, q.re));
      q.im = fma(u.re, v.im, fma(u.im, v.re, q.im));
    }
    gl = 2.0 * (c.pkind == 2 ? q.im : q.re);
    if (start) gl *= start->lscale;
    wave_sync();
    QT_STAMP(18);
  }

  // sqrt(|Tr((R - C)^2)|) / sqrt(2) for the matrix R held one element (i, j) per lane (geometry.py:16-20):
  // Tr(Delta Delta) = sum_ij Delta_ij Delta_ji, the transposed element fetched from lane (j, i) of the group.
  // Executed by every lane of the wavefront (DPP reductions); every lane of a group returns the same bits.
  __device__ __forceinline__ static double hs_to_centre(const Ctx& c, cd r, const double* __restrict__ centre) {
    const double2 cc = *reinterpret_cast<const double2*>(centre + 2 * c.l);
    const cd dl{r.re - cc.x, r.im - cc.y};
    const int src = (int)(threadIdx.x & 63) - c.l + c.j * d + c.i;
    const cd dt{__shfl(dl.re, src, 64), __shfl(dl.im, src, 64)};
    const cd t = hs_term(dl, dt);
    const double sr = gsum<G>(t.re);
    const double si = gsum<G>(t.im);
    const double v = sqrt(hypot(sr, si)) / sqrt(2.0);
    return v < 1e-15 ? 0.0 : v;
  }
  // Result of trial b: element (i, j) = r.  `store` = this lane's group owns a live trial; the distance is computed by
  // all lanes (o.dist is uniform over the launch), only the stores are masked.
  __device__ __forceinline__ static void emit(const Ctx& c, const EstOut& o, int b, bool store, cd r) {
    if (o.dist) {
      const double v = hs_to_centre(c, r, o.centre);
      if (store && c.l == 0) o.dist[b] = v;
    }
    if (store && o.rho) {
      double* out = o.rho + ((size_t)b * D + c.l) * 2;
      out[0] = r.re;
      out[1] = r.im;
    }
  }
};

// =========================================================================================
// kernels (workgroup = 4 waves; a wave owns 64/G trials; `live` masks the padding of the last block)
// =========================================================================================

// a6 + a7
template <int NQ, bool ALDS>
__global__ void __launch_bounds__(256) k_lin_batch(PovmView pv, const int64_t* __restrict__ counts, int B, int physical,
                                                   EstOut rho, double* __restrict__ bloch_out,
                                                   int32_t* __restrict__ status) {
  using S = Small<NQ, ALDS>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  typename S::Ctx c;
  bool live;
  const int b = S::trial_index(B, &live);
  const int bb = live ? b : B - 1;  // padding groups recompute the last trial; nothing is stored
  typename S::Prefetch pf;
  S::prefetch_counts(pf, counts + (size_t)bb * pv.M, pv);
  S::make_ctx(c, smem, pv);
  QT_STAMP(0);
  S::load_image(c, pv.PinvT);
  const bool shots_ok = S::load_freq(c, counts + (size_t)bb * pv.M, &pf);
  QT_STAMP(1);
  double bl;
  cd r = S::lin_invert(c, bl);
  QT_STAMP(2);
  if (physical) r = S::make_feasible(c, r, nullptr, nullptr);
  QT_STAMP(9);
  S::emit(c, rho, b, live, r);
  if (live) {
    if (bloch_out) bloch_out[(size_t)b * S::D + c.l] = bl;
    if (status && c.l == 0) status[b] = !shots_ok ? 5 : (r.re == r.re) ? 0 : 4;
  }
}

// a8 forward: rho -> x
template <int NQ>
__global__ void __launch_bounds__(256) k_chol_param(PovmView pv, const double* __restrict__ rho, int B,
                                                    double* __restrict__ x, int32_t* __restrict__ status) {
  using S = Small<NQ, false>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  typename S::Ctx c;
  S::make_ctx(c, smem, pv);
  bool live;
  const int b = S::trial_index(B, &live);
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
__global__ void __launch_bounds__(256) k_chol_unparam(PovmView pv, const double* __restrict__ x, int B,
                                                      double* __restrict__ llh) {
  using S = Small<NQ, false>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  typename S::Ctx c;
  S::make_ctx(c, smem, pv);
  bool live;
  const int b = S::trial_index(B, &live);
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
template <int NQ, bool ALDS>
__global__ void __launch_bounds__(256) k_nll_batch(PovmView pv, const double* __restrict__ x,
                                                   const int64_t* __restrict__ counts, int B, double* __restrict__ f,
                                                   double* __restrict__ grad) {
  using S = Small<NQ, ALDS>;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  typename S::Ctx c;
  S::make_ctx(c, smem, pv);
  bool live;
  const int b = S::trial_index(B, &live);
  const int bb = live ? b : B - 1;
  S::load_image(c, pv.Aw);
  S::load_freq(c, counts + (size_t)bb * pv.M);
  double fv, gl;
  S::nll_grad(c, x[(size_t)bb * S::D + c.l], fv, gl);
  if (live) {
    if (c.l == 0) f[b] = fv;
    if (grad) grad[(size_t)b * S::D + c.l] = gl;
  }
}

// a10, part 1: starting point (state.py:205-212), Cholesky parametrisation and the first
// (value, gradient) evaluation.  Trials whose gradient already meets gtol -- every full-rank
// high-shot trial, where BFGS exits at iteration 0 -- are finished here; the
// rest hand x0, g0, f0 to k_mle_bfgs.  Keeping the D x D inverse Hessian out of this kernel
// keeps its register footprint small.
// Held to four wavefronts per SIMD (128 VGPRs): the saturated batches run four workgroups per CU, and one
// register more -- the EstOut pointers of round 3 made it 129 -- takes a workgroup off every CU (measured: 0.372 -> 0.416 ms
// per 65 536 trials).
template <int NQ, bool ALDS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) k_mle_start(PovmView pv, const int64_t* __restrict__ counts, int B, int init,
                                                   int max_iter, double gtol, EstOut rho,
                                                   int32_t* __restrict__ nit_out, int32_t* __restrict__ nfev_out,
                                                   double* __restrict__ fun_out, int32_t* __restrict__ status_out,
                                                   double* __restrict__ ws_x, double* __restrict__ ws_g,
                                                   double* __restrict__ ws_f, int32_t* __restrict__ ws_active) {
  using S = Small<NQ, ALDS>;
  constexpr int D = S::D, G = S::G, d = S::d;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  typename S::Ctx c;
  bool live;
  const int b = S::trial_index(B, &live);
  const int bb = live ? b : B - 1;
  typename S::Prefetch pf;
  S::prefetch_counts(pf, counts + (size_t)bb * pv.M, pv);
  S::make_ctx(c, smem, pv);
  const bool shots_ok = S::load_freq(c, counts + (size_t)bb * pv.M, &pf);
  int ok;
  double xk;
  typename S::StartPoint sp;
  if (init == 0) {
    S::load_image(c, pv.PinvT);
    double bl;
    const cd lin = S::lin_invert(c, bl);
    S::load_image(c, pv.Aw);  // (barrier inside: every wave is past its read of the left inverse)
    sp.rho = S::make_feasible(c, lin, &xk, &ok, &sp.lscale);  // physical 'lin' estimate, Cholesky-parametrised
  } else {
    S::load_image(c, pv.Aw);
    sp.rho = cd{c.i == c.j ? 1.0 / d : 0.0, 0.0};
    sp.lscale = 1.0;
    xk = S::cholesky_param(c, sp.rho, ok);
  }
  double fk, gk;
  cd rho_l;  // L L^dagger / Tr at x_k: what the trial returns if BFGS does not move
  S::nll_grad(c, xk, fk, gk, &rho_l, true, &sp);
  const double gnorm = gmax<G>(fabs(gk));
  const bool iterate = shots_ok && ok && (gnorm > gtol) && (0 < max_iter);
  int status = 0;
  if (!shots_ok) status = 5;
  else if (!ok) status = 1;
  else if (!iterate) {
    const double xn = gmax<G>(fabs(xk));
    if (0 >= max_iter) status = 3;
    else if (gnorm != gnorm || fk != fk || xn != xn) status = 4;
  }
  S::emit(c, rho, b, live, rho_l);
  if (live) {
    if (iterate) {  // the hand-off is written only for trials that go on to k_mle_bfgs
      ws_x[(size_t)b * D + c.l] = xk;
      ws_g[(size_t)b * D + c.l] = gk;
    }
    if (c.l == 0) {
      if (iterate) ws_f[b] = fk;
      ws_active[b] = iterate ? 1 : 0;
      if (nit_out) nit_out[b] = 0;
      if (nfev_out) nfev_out[b] = ok ? 1 : 0;
      if (fun_out) fun_out[b] = fk;
      if (status_out) status_out[b] = status;
    }
  }
}

// a10, part 2: the BFGS iterations (scipy _minimize_bfgs) of the trials whose first gradient did not
// meet gtol.  One (value, gradient) evaluation per loop pass; the line search is the state machine of
// qt_linesearch.h; the inverse Hessian is one row per lane in registers.  `mine` marks the groups that
// iterate; the others idle through the evaluations on finite dummy values.
template <int NQ, bool ALDS>
__device__ __forceinline__ void bfgs_iterate(const typename Small<NQ, ALDS>::Ctx& c, bool mine, double xk, double gk,
                                             double fk, int b, int max_iter, double gtol, EstOut rho,
                                             int32_t* __restrict__ nit_out, int32_t* __restrict__ nfev_out,
                                             double* __restrict__ fun_out, int32_t* __restrict__ status_out) {
  using S = Small<NQ, ALDS>;
  constexpr int D = S::D, G = S::G;
  bool active = mine;
  double H[D];
#pragma unroll
  for (int k = 0; k < D; ++k) H[k] = (k == c.l) ? 1.0 : 0.0;
  double old_old = fk + sqrt(gsum<G>(gk * gk)) / 2.0;
  double pk = -gk, stp = 0.0;  // H0 = I
  int kiter = 0, nfev = 1, status = 0;
  LineSearch ls;
  double* vec = c.vec();
  ls.start(fk, old_old, gsum<G>(gk * pk), &stp);
  const int eval_cap = (max_iter + 2) * 130;  // hard stop: every wave leaves the loop

#ifdef QT_PHASE_TIMING  // slots 21 / 22: clocks in this loop / inside its nll_grad calls; 23 / 24: iterations, evaluations
  long long t_nll = 0;
  const long long t_loop0 = (long long)__builtin_readcyclecounter();
#endif
  while (__any(active)) {  // per wave: the waves of a workgroup do not synchronise here
    double ft, gt;
#ifdef QT_PHASE_TIMING
    const long long t_e0 = (long long)__builtin_readcyclecounter();
#endif
    S::nll_grad(c, xk + stp * pk, ft, gt);  // executed by the whole wave; finished trials idle through it
#ifdef QT_PHASE_TIMING
    t_nll += (long long)__builtin_readcyclecounter() - t_e0;
#endif
    if (active && ++nfev > eval_cap) {
      status = 2;
      active = false;
    }
    if (active) {
      const double dphi = gsum<G>(gt * pk);
      double next = stp;
      const int r = ls.advance(stp, ft, dphi, &next);
      if (r == LS_EVAL) {
        stp = next;
      } else if (r == LS_FAIL) {
        status = 2;
        active = false;
      } else {
        // step accepted: x += s, y = g_new - g, BFGS update (rhok = 1000 if y.s == 0)
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
          // four partial sums: a lone wave pays ~16 ns per DEPENDENT FP64 instruction under load
          // (profiles/round1_v14_ubench_valu_f64.txt); a 64-term chain would be 1 us of latency
          double uu[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int k = 0; k < D; ++k) uu[k & 3] = fma(H[k], vec[k], uu[k & 3]);
          double u = (uu[0] + uu[1]) + (uu[2] + uu[3]);
          const double yhy = gsum<G>(yk * u);
          wave_sync();
          double* ubuf = reinterpret_cast<double*>(c.A());  // 2 D doubles: u | s
          ubuf[c.l] = u;
          ubuf[D + c.l] = sk;
          wave_sync();
          const double cc = rhok * rhok * yhy + rhok;
          // (regrouped as H_lk += a_l s_k + b_l u_k with (u_k, s_k) read as one 16-byte pair: two FMAs per element
          //  instead of five operations, and measured 6 % SLOWER per iteration -- scripts/bfgs_phase_timing.py)
#pragma unroll
          for (int k = 0; k < D; ++k)
            H[k] += -rhok * (u * ubuf[D + k] + sk * ubuf[k]) + cc * sk * ubuf[D + k];
          wave_sync();
          if (!(kiter < max_iter)) {
            active = false;
          } else {
            vec[c.l] = gk;
            wave_sync();
            double hh[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < D; ++k) hh[k & 3] = fma(H[k], vec[k], hh[k & 3]);
            const double hp = (hh[0] + hh[1]) + (hh[2] + hh[3]);
            wave_sync();
            pk = -hp;
            ls.start(fk, old_old, gsum<G>(gk * pk), &stp);
          }
        }
      }
    }
  }
#ifdef QT_PHASE_TIMING
  QT_STAMP_VAL(21, (long long)__builtin_readcyclecounter() - t_loop0);
  QT_STAMP_VAL(22, t_nll);
  QT_STAMP_VAL(23, (long long)kiter);
  QT_STAMP_VAL(24, (long long)nfev);
#endif
  if (status == 0) {
    const double gn = gmax<G>(fabs(gk));
    const double xn = gmax<G>(fabs(xk));
    if (kiter >= max_iter) status = 3;
    else if (gn != gn || fk != fk || xn != xn) status = 4;
  }
  // ---- result: L L^dagger / Tr  (state.py:214-215)
  double tr;
  const cd m = S::build_llh(c, xk, tr);
  S::emit(c, rho, b, mine, cd{m.re / tr, m.im / tr});
  if (mine) {
    if (c.l == 0) {
      if (nit_out) nit_out[b] = kiter;
      if (nfev_out) nfev_out[b] = nfev;
      if (fun_out) fun_out[b] = fk;
      if (status_out) status_out[b] = status;
    }
  }
}

// The same iterations WITHOUT the inverse Hessian in registers, for batches that fill the chip (k_mle_bfgs): with
// H_0 = I the matrix scipy updates, H <- (I - rho s y^T) H (I - rho y s^T) + rho s s^T, is exactly the product form
// the two-loop recursion evaluates, so p = -H_k g is computed from the (s_i, y_i) pairs of the accepted steps (the
// n = 4, 5 kernels have always done this, qt_large.h).  Lane l only ever touches element l of each pair: private,
// coalesced streams in a global workspace `pairs` [B][max_iter][2][D] that stay in L2; rho_i, alpha_i and -- across
// every evaluation -- the line-search state live in the trial's LDS (`pv.extra` doubles).  Registers: ~150 instead
// of 256 + 91 AGPRs, i.e. three waves per SIMD instead of one, and 2 k reductions + 4 k FMAs per iteration instead
// of two 64-term products and a 64-entry rank-two update.  Same iterates as the dense form to rounding
// (tests/test_gpu_state.py runs the reference's 70 golden trials through both).
// LP > 0 (k_mle_fused: one workgroup per CU, LDS to spare): the first LP pairs stay in the trial's LDS and only later
// ones go to the global workspace -- a lone wave would otherwise wait out an L2 round trip per block of pairs.
template <int NQ, bool ALDS, int LP = 0>
__device__ __forceinline__ void bfgs_iterate_2l(const typename Small<NQ, ALDS>::Ctx& c, bool mine, double xk, double gk,
                                                double fk, int b, int max_iter, double gtol, EstOut rho,
                                                int32_t* __restrict__ nit_out, int32_t* __restrict__ nfev_out,
                                                double* __restrict__ fun_out, int32_t* __restrict__ status_out,
                                                double* __restrict__ pairs) {
  using S = Small<NQ, ALDS>;
  constexpr int D = S::D, G = S::G;
  bool active = mine;
  double* my = pairs + (size_t)b * max_iter * 2 * D + c.l;  // s_i[l] at my[2 i D], y_i[l] at my[(2 i + 1) D]
  double* lsb = c.extra();                                  // [LineSearch::SLOTS]
  double* prho = lsb + LineSearch::SLOTS;                   // [max_iter]
  double* palpha = prho + max_iter;                         // [max_iter]
  double* lp = palpha + max_iter + c.l;                     // [LP][2][D]: s_i[l] at lp[2 i D], y_i[l] at lp[(2 i + 1) D]
  double old_old = fk + sqrt(gsum<G>(gk * gk)) / 2.0;
  double pk = -gk, stp = 0.0;  // H0 = I
  int kiter = 0, nfev = 1, status = 0;
  // The line-search state lives IN LDS (every lane of the group reads and writes the same words with the same values):
  // as a register-resident struct its ~35 doubles plus the temporaries of dcstep made the allocation 240 VGPRs.
  static_assert(sizeof(LineSearch) <= LineSearch::SLOTS * sizeof(double), "LDS slot of the line-search state");
  LineSearch& ls = *reinterpret_cast<LineSearch*>(lsb);
  ls.start(fk, old_old, gsum<G>(gk * pk), &stp);
  wave_sync();
  const int eval_cap = (max_iter + 2) * 130;  // hard stop: every wave leaves the loop
  while (__any(active)) {  // per wave: the waves of a workgroup do not synchronise here
    double ft, gt;
    S::nll_grad(c, xk + stp * pk, ft, gt);  // executed by the whole wave; finished trials idle through it
    if (active && ++nfev > eval_cap) {
      status = 2;
      active = false;
    }
    if (active) {
      const double dphi = gsum<G>(gt * pk);
      double next = stp;
      const int r = ls.advance(stp, ft, dphi, &next);
      if (r == LS_EVAL) {
        stp = next;
      } else if (r == LS_FAIL) {
        status = 2;
        active = false;
      } else {
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
        } else if (!(kiter < max_iter)) {
          active = false;
        } else {
          const double ys = gsum<G>(yk * sk);
          const double rhok = (ys == 0.0) ? 1000.0 : 1.0 / ys;
          const int np = kiter - 1;  // index of the new pair
          if (np < LP) {
            lp[(2 * np) * D] = sk;
            lp[(2 * np + 1) * D] = yk;
          } else {
            my[(size_t)(2 * np) * D] = sk;
            my[(size_t)(2 * np + 1) * D] = yk;
          }
          if (c.l == 0) prho[np] = rhok;
          wave_sync();
          double q = gk;
          for (int i = np; i >= 0; --i) {
            double si = sk, yi = yk;
            if (i != np) {
              if (i < LP) {
                si = lp[(2 * i) * D];
                yi = lp[(2 * i + 1) * D];
              } else {
                si = my[(size_t)(2 * i) * D];
                yi = my[(size_t)(2 * i + 1) * D];
              }
            }
            const double a = prho[i] * gsum<G>(si * q);
            if (c.l == 0) palpha[i] = a;
            q = fma(-a, yi, q);
          }
          wave_sync();
          for (int i = 0; i <= np; ++i) {
            double si = sk, yi = yk;
            if (i != np) {
              if (i < LP) {
                si = lp[(2 * i) * D];
                yi = lp[(2 * i + 1) * D];
              } else {
                si = my[(size_t)(2 * i) * D];
                yi = my[(size_t)(2 * i + 1) * D];
              }
            }
            const double bb = prho[i] * gsum<G>(yi * q);
            q = fma(si, palpha[i] - bb, q);
          }
          pk = -q;
          ls.start(fk, old_old, gsum<G>(gk * pk), &stp);
        }
      }
    }
    wave_sync();
  }
  if (status == 0) {
    const double gn = gmax<G>(fabs(gk));
    const double xn = gmax<G>(fabs(xk));
    if (kiter >= max_iter) status = 3;
    else if (gn != gn || fk != fk || xn != xn) status = 4;
  }
  double tr;
  const cd m = S::build_llh(c, xk, tr);
  S::emit(c, rho, b, mine, cd{m.re / tr, m.im / tr});
  if (mine) {
    if (c.l == 0) {
      if (nit_out) nit_out[b] = kiter;
      if (nfev_out) nfev_out[b] = nfev;
      if (fun_out) fun_out[b] = fk;
      if (status_out) status_out[b] = status;
    }
  }
}

#ifndef QT_BFGS_WAVES
#define QT_BFGS_WAVES 2  // waves per SIMD the BFGS kernel is compiled for; measured at B = 65 536 (15-iteration trials): 3 waves
                         // (<= 168 VGPRs, 18 spilled) 5.29 ms, 2 waves (205 VGPRs, no scratch) 5.39 ms -- instruction-bound either way
#endif
template <int NQ, bool ALDS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QT_BFGS_WAVES))) k_mle_bfgs(PovmView pv, const int64_t* __restrict__ counts, int B, int max_iter,
                                                  double gtol, EstOut rho, int32_t* __restrict__ nit_out,
                                                  int32_t* __restrict__ nfev_out, double* __restrict__ fun_out,
                                                  int32_t* __restrict__ status_out, const double* __restrict__ ws_x,
                                                  const double* __restrict__ ws_g, const double* __restrict__ ws_f,
                                                  const int32_t* __restrict__ ws_active, double* __restrict__ pairs) {
  using S = Small<NQ, ALDS>;
  constexpr int D = S::D;
  bool live;
  const int b = S::trial_index(B, &live);
  const bool mine = live && ws_active[b] != 0;
  if (!__syncthreads_or(mine)) return;  // nothing left to iterate in this workgroup
  extern __shared__ __attribute__((aligned(16))) double smem[];
  typename S::Ctx c;
  S::make_ctx(c, smem, pv);
  const int bb = live ? b : B - 1;
  S::load_image(c, pv.Aw);
  S::load_freq(c, counts + (size_t)bb * pv.M);
  // inactive trials of a live wave idle through the loop on dummy finite values
  const double xk = mine ? ws_x[(size_t)b * D + c.l] : (c.l < S::d ? 1.0 : 0.0);
  const double gk = mine ? ws_g[(size_t)b * D + c.l] : 0.0;
  const double fk = mine ? ws_f[b] : 0.0;
  // n = 3: two-loop form (the 64 x 64 inverse Hessian would cost 128 VGPRs per lane); n = 1, 2: H is 4 / 16 doubles
  // per lane and stays in registers
  if constexpr (NQ == 3)
    bfgs_iterate_2l<NQ, ALDS>(c, mine, xk, gk, fk, b, max_iter, gtol, rho, nit_out, nfev_out, fun_out, status_out, pairs);
  else
    bfgs_iterate<NQ, ALDS>(c, mine, xk, gk, fk, b, max_iter, gtol, rho, nit_out, nfev_out, fun_out, status_out);
}

constexpr int kFusedLdsPairs = 24;  // (s, y) pairs of k_mle_fused<3> kept in LDS: 24 KB per trial, 4 trials per workgroup

// a10 in ONE launch, for batches small enough that its 256-VGPR footprint (two waves per SIMD) is no
// handicap: start point, first evaluation and -- for the waves that still hold an open trial -- the BFGS
// loop.  Saves the second launch (2.5-4 us when nothing iterates, ~10 % of a 1000-trial step).
template <int NQ, bool ALDS>
__device__ __forceinline__ void mle_fused_body(const PovmView& pv, const int64_t* __restrict__ counts, int B, int init,
                                               int max_iter, double gtol, EstOut rho,
                                               int32_t* __restrict__ nit_out, int32_t* __restrict__ nfev_out,
                                               double* __restrict__ fun_out, int32_t* __restrict__ status_out,
                                               double* __restrict__ pairs) {
  using S = Small<NQ, ALDS>;
  constexpr int D = S::D, G = S::G, d = S::d;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  typename S::Ctx c;
  bool live;
  const int b = S::trial_index(B, &live);
  const int bb = live ? b : B - 1;
  typename S::Prefetch pf;
  S::prefetch_counts(pf, counts + (size_t)bb * pv.M, pv);
  S::make_ctx(c, smem, pv);
  QT_STAMP(0);
  const bool shots_ok = S::load_freq(c, counts + (size_t)bb * pv.M, &pf);
  QT_STAMP(1);
  int ok;
  double xk;
  typename S::StartPoint sp;
  if (init == 0) {
    S::load_image(c, pv.PinvT);
    double bl;
    const cd lin = S::lin_invert(c, bl);
    QT_STAMP(2);
    S::load_image(c, pv.Aw);
    sp.rho = S::template make_feasible<true>(c, lin, &xk, &ok, &sp.lscale);  // one wave per SIMD: speculative inverse (cholesky_param)
  } else {
    S::load_image(c, pv.Aw);
    sp.rho = cd{c.i == c.j ? 1.0 / d : 0.0, 0.0};
    sp.lscale = 1.0;
    xk = S::cholesky_param(c, sp.rho, ok);
  }
  QT_STAMP(8);
  double fk, gk;
  cd rho_l;
  S::nll_grad(c, xk, fk, gk, &rho_l, true, &sp);
  const double gnorm = gmax<G>(fabs(gk));
  QT_STAMP(9);
  const bool iterate = live && shots_ok && ok && (gnorm > gtol) && (0 < max_iter);
  S::emit(c, rho, b, live && !iterate, rho_l);  // (trials that iterate are written at the end of the loop)
  if (!iterate) {
    int status = 0;
    if (!shots_ok) status = 5;
    else if (!ok) status = 1;
    else {
      const double xn = gmax<G>(fabs(xk));
      if (0 >= max_iter) status = 3;
      else if (gnorm != gnorm || fk != fk || xn != xn) status = 4;
    }
    if (live) {
      if (c.l == 0) {
        if (nit_out) nit_out[b] = 0;
        if (nfev_out) nfev_out[b] = ok ? 1 : 0;
        if (fun_out) fun_out[b] = fk;
        if (status_out) status_out[b] = status;
      }
    }
  }
  QT_STAMP(10);
  if (!__any(iterate)) return;  // per wave
  if (!iterate) {                // finite dummies for the groups of this wave that are already done
    xk = (c.l < d) ? 1.0 : 0.0;
    gk = 0.0;
    fk = 0.0;
  }
  // n = 3: two-loop form, the first kFusedLdsPairs (s, y) pairs in LDS (the 64 x 64 inverse Hessian took 128 VGPRs per lane
  // and ~460 AGPR moves per iteration); n = 1, 2: the 4 / 16-entry Hessian rows stay in registers
  if constexpr (NQ == 3)
    bfgs_iterate_2l<NQ, ALDS, kFusedLdsPairs>(c, iterate, xk, gk, fk, b, max_iter, gtol, rho, nit_out, nfev_out, fun_out,
                                              status_out, pairs);
  else
    bfgs_iterate<NQ, ALDS>(c, iterate, xk, gk, fk, b, max_iter, gtol, rho, nit_out, nfev_out, fun_out, status_out);
}

// Two entry points over the same body: the fully mixed start (`init = 'mixed'`: every trial iterates, ~10x the duration)
// runs under its own kernel name, so that a profiler's per-kernel average of k_mle_fused is the average of the
// 'lin'-start launches (bench.py's timed steps) and not a mixture with the iterating side measurements.
template <int NQ, bool ALDS>
__global__ void __launch_bounds__(256) k_mle_fused(PovmView pv, const int64_t* __restrict__ counts, int B, int max_iter,
                                                   double gtol, EstOut rho, int32_t* __restrict__ nit_out,
                                                   int32_t* __restrict__ nfev_out, double* __restrict__ fun_out,
                                                   int32_t* __restrict__ status_out, double* __restrict__ pairs) {
  mle_fused_body<NQ, ALDS>(pv, counts, B, 0, max_iter, gtol, rho, nit_out, nfev_out, fun_out, status_out, pairs);
}
template <int NQ, bool ALDS>
__global__ void __launch_bounds__(256) k_mle_fused_mixed(PovmView pv, const int64_t* __restrict__ counts, int B,
                                                         int max_iter, double gtol, EstOut rho,
                                                         int32_t* __restrict__ nit_out, int32_t* __restrict__ nfev_out,
                                                         double* __restrict__ fun_out, int32_t* __restrict__ status_out,
                                                         double* __restrict__ pairs) {
  mle_fused_body<NQ, ALDS>(pv, counts, B, 1, max_iter, gtol, rho, nit_out, nfev_out, fun_out, status_out, pairs);
}

// Metropolis-Hastings chain on the Cholesky parameters (reference mhmc.py:80-119 with
// `normalized_update`, interval.py:735-750): one chain per lane group, the proposal increments and the
// uniforms drawn on the host in the reference's order.  Step t:
//   x' = (x + step * delta_t) / ||x + step * delta_t||,  alpha = exp(nll(x) - nll(x')),  accept iff u_t <= alpha.
// chain[c][t][:] = the state AFTER step t, accepted[c][t] = 0 / 1.
template <int NQ, bool ALDS>
__global__ void __launch_bounds__(256) k_mhmc_state(PovmView pv, const int64_t* __restrict__ counts, int C,
                                                    const double* __restrict__ x_init, const double* __restrict__ deltas,
                                                    const double* __restrict__ uniforms, int T_steps, double step,
                                                    double* __restrict__ chain, int32_t* __restrict__ accepted) {
  using S = Small<NQ, ALDS>;
  constexpr int D = S::D, G = S::G;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  typename S::Ctx c;
  S::make_ctx(c, smem, pv);
  bool live;
  const int b = S::trial_index(C, &live);
  const int bb = live ? b : C - 1;
  S::load_image(c, pv.Aw);
  S::load_freq(c, counts + (size_t)bb * pv.M);
  double x = x_init[(size_t)bb * D + c.l];
  double f, unused;
  S::nll_grad(c, x, f, unused, nullptr, false);
  const double* dl = deltas + (size_t)bb * T_steps * D + c.l;
  const double* un = uniforms + (size_t)bb * T_steps;
  for (int t = 0; t < T_steps; ++t) {
    const double xp = fma(step, dl[(size_t)t * D], x);
    const double nrm = sqrt(gsum<G>(xp * xp));
    const double xn = xp / nrm;
    double fn;
    S::nll_grad(c, xn, fn, unused, nullptr, false);
    const double alpha = exp(f - fn);
    const bool acc = un[t] <= alpha;  // false for a NaN alpha, like the reference's comparison
    if (acc) {
      x = xn;
      f = fn;
    }
    if (live) {
      chain[((size_t)b * T_steps + t) * D + c.l] = x;
      if (c.l == 0) accepted[(size_t)b * T_steps + t] = acc ? 1 : 0;
    }
  }
}

}  // namespace qt
