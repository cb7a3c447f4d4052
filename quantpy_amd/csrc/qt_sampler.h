// a4 / a12 / a16 host side: the multinomial sampler behind `experiment()` (reference state.py:109-114:
// `np.random.multinomial(n_s, p_s)` per POVM setting on NumPy's global legacy stream).
//
// The draws themselves are a third-party algorithm: NumPy's legacy `RandomState.multinomial` (numpy 2.2,
// numpy/random/mtrand.pyx) -> `legacy_random_binomial` (legacy-distributions.c) -> Kachitvichyanukul & Schmeiser's
// BTPE rejection sampler above n p = 30 and sequential inversion below, both fed by 53-bit doubles built from two
// MT19937 words.  The reference spends one Python call per setting and resample on it (0.9 ms per experiment at
// n = 3; 68 of the 70 ms of a 2000-resample bootstrap CI); this file restates the published
// algorithm so that a whole bootstrap's draws are one C loop on the SAME stream: it takes the 624-word MT19937 key
// and position of `np.random.get_state()`, consumes exactly the words NumPy would, and hands the advanced state
// back for `np.random.set_state()`.  tests/test_host_logic.py pins it bit for bit against numpy.random itself
// (counts and final generator state) over the (n, p) regimes the two branches cover.
//
// The binomial / multinomial routines are templates over the generator of uniform doubles and compile for host and
// device: with `Mt19937` on the host they ARE the bit-exact path above; with `Philox` (counter-based, below) one GPU
// thread draws one multinomial row independently of every other row (`k_multinomial_rows`, opt-in
// `sampler="device"`: same distribution, NOT the reference's stream).  Floating-point contraction is off in here:
// every product and sum rounds as NumPy's build rounds it.
#pragma once
#include <math.h>
#include <stdint.h>

namespace qt_sampler {

#pragma clang fp contract(off)

#if defined(__HIPCC__)
#define QT_SAMPLER_HD __host__ __device__
#else  // plain host compiler: tests/host/sampler_host.cpp under g++ -fsanitize=address,undefined
#define QT_SAMPLER_HD
#endif

struct Mt19937 {
  uint32_t* key;  // [624], updated in place
  int pos;        // 0..624 (624 = the block is used up)
  inline uint32_t next();
};

inline void mt_refill(Mt19937& g) {
  constexpr int N = 624, M = 397;
  constexpr uint32_t UPPER = 0x80000000u, LOWER = 0x7fffffffu, MAG = 0x9908b0dfu;
  uint32_t* k = g.key;
  for (int i = 0; i < N; ++i) {
    const uint32_t y = (k[i] & UPPER) | (k[(i + 1) % N] & LOWER);
    k[i] = k[(i + M) % N] ^ (y >> 1) ^ ((y & 1u) ? MAG : 0u);
  }
  g.pos = 0;
}

inline uint32_t mt_next(Mt19937& g) {
  if (g.pos >= 624) mt_refill(g);
  uint32_t y = g.key[g.pos++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

inline uint32_t Mt19937::next() { return mt_next(*this); }

// the legacy 53-bit double: 27 high bits of one word, 26 of the next
template <class G>
QT_SAMPLER_HD inline double uniform53(G& g) {
  const int32_t a = (int32_t)(g.next() >> 5), b = (int32_t)(g.next() >> 6);
  return (a * 67108864.0 + b) / 9007199254740992.0;
}

// Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11): a keyed
// bijection of a 128-bit counter, ten rounds of two 32x32 -> 64 multiplies.  A stream is (key = seed,
// counter = {block index, row low, row high, substream}); it hands out the four words of a block in order.
QT_SAMPLER_HD inline void philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
  for (int round = 0; round < 10; ++round) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0, out[1] = c1, out[2] = c2, out[3] = c3;
}

struct Philox {
  uint32_t key[2], ctr[4], word[4];
  int have;
  QT_SAMPLER_HD Philox(uint64_t seed, uint64_t row, uint32_t substream) : have(0) {
    key[0] = (uint32_t)seed, key[1] = (uint32_t)(seed >> 32);
    ctr[0] = 0, ctr[1] = (uint32_t)row, ctr[2] = (uint32_t)(row >> 32), ctr[3] = substream;
  }
  QT_SAMPLER_HD uint32_t next() {
    if (have == 0) {
      philox4x32_10(ctr, key, word);
      ++ctr[0];  // 2^32 blocks per row: no binomial comes near
      have = 4;
    }
    return word[4 - have--];
  }
};

// The set-up constants of one (n, p): NumPy caches them in RandomState._binomial; they are pure functions of
// (n, p), so a local cache gives the same draws.
struct BinomialSetup {
  bool valid = false;
  int64_t n = 0;
  double p = 0.0;
  // inversion
  double q, qn, np;
  int64_t bound;
  // BTPE
  double r, fm, xm, xl, xr, c, laml, lamr, p1, p2, p3, p4;
  int64_t m;
  bool btpe = false;
};

// n p <= 30: walk the mass function from 0 upwards; restart with a fresh uniform past `bound`
template <class G>
QT_SAMPLER_HD inline int64_t binomial_inversion(G& g, int64_t n, double p, BinomialSetup& s) {
  if (!s.valid || s.btpe || s.n != n || s.p != p) {
    s.valid = true, s.btpe = false, s.n = n, s.p = p;
    s.q = 1.0 - p;
    s.qn = exp(n * log(s.q));
    s.np = n * p;
    const double lim = s.np + 10.0 * sqrt(s.np * s.q + 1);
    s.bound = (int64_t)((double)n < lim ? (double)n : lim);
  }
  const double q = s.q, qn = s.qn;
  const int64_t bound = s.bound;
  int64_t x = 0;
  double px = qn, u = uniform53(g);
  while (u > px) {
    ++x;
    if (x > bound) {
      x = 0;
      px = qn;
      u = uniform53(g);
    } else {
      u -= px;
      px = ((n - x + 1) * p * px) / (x * q);
    }
  }
  return x;
}

// n p > 30 (p <= 0.5): BTPE -- triangle, parallelogram and two exponential tails as the envelope, then the
// squeeze / exact acceptance tests of the published algorithm
template <class G>
QT_SAMPLER_HD inline int64_t binomial_btpe(G& g, int64_t n, double p, BinomialSetup& s) {
  if (!s.valid || !s.btpe || s.n != n || s.p != p) {
    s.valid = true, s.btpe = true, s.n = n, s.p = p;
    s.r = p < 1.0 - p ? p : 1.0 - p;
    s.q = 1.0 - s.r;
    s.fm = n * s.r + s.r;
    s.m = (int64_t)floor(s.fm);
    s.p1 = floor(2.195 * sqrt(n * s.r * s.q) - 4.6 * s.q) + 0.5;
    s.xm = s.m + 0.5;
    s.xl = s.xm - s.p1;
    s.xr = s.xm + s.p1;
    s.c = 0.134 + 20.5 / (15.3 + s.m);
    double a = (s.fm - s.xl) / (s.fm - s.xl * s.r);
    s.laml = a * (1.0 + a / 2.0);
    a = (s.xr - s.fm) / (s.xr * s.q);
    s.lamr = a * (1.0 + a / 2.0);
    s.p2 = s.p1 * (1.0 + 2.0 * s.c);
    s.p3 = s.p2 + s.c / s.laml;
    s.p4 = s.p3 + s.c / s.lamr;
  }
  const double r = s.r, q = s.q, xm = s.xm, xl = s.xl, xr = s.xr, c = s.c, laml = s.laml, lamr = s.lamr;
  const double p1 = s.p1, p2 = s.p2, p3 = s.p3, p4 = s.p4;
  const int64_t m = s.m;
  const double nrq = n * r * q;
  int64_t y;
  for (;;) {
    const double u = uniform53(g) * p4;
    double v = uniform53(g);
    if (u <= p1) {  // triangle: accepted at once
      y = (int64_t)floor(xm - p1 * v + u);
      break;
    }
    if (u <= p2) {  // parallelogram
      const double x = xl + (u - p1) / c;
      v = v * c + 1.0 - fabs(m - x + 0.5) / p1;
      if (v > 1.0) continue;
      y = (int64_t)floor(x);
    } else if (u <= p3) {  // left tail
      y = (int64_t)floor(xl + log(v) / laml);
      if (y < 0 || v == 0.0) continue;
      v = v * (u - p2) * laml;
    } else {  // right tail
      y = (int64_t)floor(xr - log(v) / lamr);
      if (y > n || v == 0.0) continue;
      v = v * (u - p3) * lamr;
    }
    const int64_t k = y > m ? y - m : m - y;
    if (!(k > 20 && k < nrq / 2.0 - 1)) {  // near the mode or far out: the exact ratio f(y) / f(m) by recursion
      const double sq = r / q, a = sq * (n + 1);
      double f = 1.0;
      if (m < y) {
        for (int64_t i = m + 1; i <= y; ++i) f *= (a / i - sq);
      } else if (m > y) {
        for (int64_t i = y + 1; i <= m; ++i) f /= (a / i - sq);
      }
      if (v > f) continue;
      break;
    }
    // squeeze on log v, then the Stirling-corrected bound
    const double rho = (k / nrq) * ((k * (k / 3.0 + 0.625) + 0.16666666666666666) / nrq + 0.5);
    const double t = -k * k / (2 * nrq);
    const double lv = log(v);
    if (lv < t - rho) break;
    if (lv > t + rho) continue;
    const double x1 = y + 1, f1 = m + 1, z = n + 1 - m, w = n - y + 1;
    const double x2 = x1 * x1, f2 = f1 * f1, z2 = z * z, w2 = w * w;
    const double limit = xm * log(f1 / x1) + (n - m + 0.5) * log(z / w) + (y - m) * log(w * r / (x1 * q)) +
                         (13680. - (462. - (132. - (99. - 140. / f2) / f2) / f2) / f2) / f1 / 166320. +
                         (13680. - (462. - (132. - (99. - 140. / z2) / z2) / z2) / z2) / z / 166320. +
                         (13680. - (462. - (132. - (99. - 140. / x2) / x2) / x2) / x2) / x1 / 166320. +
                         (13680. - (462. - (132. - (99. - 140. / w2) / w2) / w2) / w2) / w / 166320.;
    if (lv > limit) continue;
    break;
  }
  return y;  // the caller only comes here with p <= 0.5: no reflection left to do
}

template <class G>
QT_SAMPLER_HD inline int64_t legacy_binomial(G& g, double p, int64_t n, BinomialSetup& s) {
  if (n == 0 || p == 0.0) return 0;  // no word consumed (RandomState.multinomial shares random_binomial's early exit)
  if (p <= 0.5) return p * n <= 30.0 ? binomial_inversion(g, n, p, s) : binomial_btpe(g, n, p, s);
  const double q = 1.0 - p;
  return n - (q * n <= 30.0 ? binomial_inversion(g, n, q, s) : binomial_btpe(g, n, q, s));
}

// one RandomState.multinomial(n, pvals) row: conditional binomials, first to last-but-one category
template <class G>
QT_SAMPLER_HD inline void legacy_multinomial(G& g, int64_t n, const double* pvals, int K, int64_t* out, BinomialSetup* cache) {
  double remaining_p = 1.0;
  int64_t dn = n;
  for (int j = 0; j < K; ++j) out[j] = 0;
  for (int j = 0; j < K - 1; ++j) {
    out[j] = legacy_binomial(g, pvals[j] / remaining_p, dn, cache[j]);
    dn -= out[j];
    if (dn <= 0) break;
    remaining_p -= pvals[j];
  }
  if (dn > 0) out[K - 1] = dn;
}

// One row of the opt-in device sampler: multinomial(n_s, p[0..K)) drawn from the Philox stream (seed, global row) by the
// conditional-binomial chain above, each conditional probability clamped into [0, 1] (rounding can leave it a hair above 1;
// a NaN must not reach the rejection loop).  Host and device: tests/test_gpu_device_sampler.py runs the host instantiation
// (tests/host/sampler_host.cpp) beside the kernel on the same streams -- same template, same uniforms, same counts.
QT_SAMPLER_HD inline void philox_multinomial_row(uint64_t seed, uint64_t global_row, int64_t n_s, const double* p, int K,
                                                  int64_t* o) {
  Philox g(seed, global_row, 0u);
  BinomialSetup setup;
  double remaining_p = 1.0;
  int64_t dn = n_s;
  int j = 0;
  for (; j < K - 1 && dn > 0; ++j) {
    const double pj = p[j];
    double pc = pj / remaining_p;
    pc = pc >= 0.0 ? (pc <= 1.0 ? pc : 1.0) : 0.0;
    const int64_t x = legacy_binomial(g, pc, dn, setup);
    o[j] = x;
    dn -= x;
    remaining_p -= pj;
  }
  for (; j < K - 1; ++j) o[j] = 0;
  o[K - 1] = dn > 0 ? dn : 0;
}

#if defined(__HIPCC__)
// Opt-in device sampler: row r (= resample r / period, setting r % period) is drawn by one thread from its own Philox
// stream (seed, first_row + r), so the counts of a row depend on nothing but (seed, global row index, n, p): any
// split of the rows over launches or ranks gives the same table.  out is [rows][K] int64, what the estimators read.
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) k_multinomial_rows(uint64_t seed, uint64_t first_row, long long rows, int period,
                                                          const int64_t* __restrict__ n, const double* __restrict__ pvals,
                                                          int K, int64_t* __restrict__ out) {
  // thread -> row: a wavefront takes ONE setting of 64 consecutive resamples (same n, same p: one branch structure
  // through the set-up and the choice inversion / BTPE), `period` wavefronts cover a block of 64 x period rows
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long span = 64LL * period;
  const long long r = (gid / span * 64 + (gid % span) % 64) * period + (gid % span) / 64;
  if (r >= rows) return;
  const int s = (int)((first_row + (uint64_t)r) % (uint64_t)period);
  philox_multinomial_row(seed, first_row + (uint64_t)r, n[s], pvals + (size_t)s * K, K, out + (size_t)r * K);
}

#endif  // __HIPCC__

}  // namespace qt_sampler
