// Host build of quantpy_amd/csrc/qt_sampler.h (the legacy-multinomial restatement that writes into NumPy's MT19937
// state) for tests/test_host_sanitizers.py: compiled with g++ -fsanitize=address,undefined and driven over the
// bit-exactness cases of tests/test_host_logic.py.  Test infrastructure, not product: the shipped instantiation is the
// one hipcc compiles into libqtomo.so (qt_legacy_multinomial), and this file mirrors its loop.
#include <stdint.h>

#include <vector>

#if defined(__GNUC__) && !defined(__clang__)
#pragma GCC diagnostic ignored "-Wunknown-pragmas"
#endif
#include "../../quantpy_amd/csrc/qt_sampler.h"

extern "C" {
// mt_key[624] / *mt_pos advanced in place; pvals[period][K]; out[rows][K]
int qt_host_legacy_multinomial(uint32_t* mt_key, int* mt_pos, long long rows, int period, const int64_t* n,
                               const double* pvals, int K, int64_t* out) {
  if (!mt_key || !mt_pos || !n || !pvals || rows < 0 || period < 1 || K < 1 || *mt_pos < 0 || *mt_pos > 624) return -1;
  qt_sampler::Mt19937 g{mt_key, *mt_pos};
  std::vector<qt_sampler::BinomialSetup> cache((size_t)period * K);
  for (long long r = 0; r < rows; ++r) {
    const int s = (int)(r % period);
    qt_sampler::legacy_multinomial(g, n[s], pvals + (size_t)s * K, K, out + (size_t)r * K, cache.data() + (size_t)s * K);
  }
  *mt_pos = g.pos;
  return 0;
}
// the rows of qt_device_multinomial on the host: same template, same Philox streams (checker of the device kernel)
void qt_host_philox_multinomial(uint64_t seed, uint64_t first_row, long long rows, int period, const int64_t* n,
                                const double* pvals, int K, int64_t* out) {
  for (long long r = 0; r < rows; ++r) {
    const int s = (int)((first_row + (uint64_t)r) % (uint64_t)period);
    qt_sampler::philox_multinomial_row(seed, first_row + (uint64_t)r, n[s], pvals + (size_t)s * K, K, out + (size_t)r * K);
  }
}
// the Philox block of the device sampler (known-answer vectors)
void qt_host_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { qt_sampler::philox4x32_10(ctr, key, out); }
}
