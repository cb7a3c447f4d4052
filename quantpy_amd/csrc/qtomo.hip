// libqtomo.so -- C ABI (include/qtomo.h) over the HIP kernels in qt_small.h / qt_ops.h /
// qt_process.h.  gfx950 only.  There is no CPU implementation behind these entry points.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <string>
#include <vector>

#include "../../include/qtomo.h"
#include "qt_large.h"
#include "qt_ops.h"
#include "qt_process.h"
#include "qt_process64.h"
#include "qt_process_wave16.h"
#include "qt_sampler.h"
#include "qt_small.h"

namespace {

#ifdef QT_PHASE_TIMING
int g_host_diag = 0;  // profile build: compile-time variant of k_lifp_gemm to launch (qt_debug_set_diag)
#endif
thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIPCHK(expr)                                                                          \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) return fail(QT_ERR_HIP, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&p, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

}  // namespace

struct qt_handle {
  int device = 0, nq = 0, d = 0, D = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr, ev_sync = nullptr;
  // Pinned host mailbox for the small transfers of host-pointer calls (counts of a few trials in, rho / nit / status
  // out): hipMemcpyAsync from / to pageable memory blocks the caller ~10 us per copy, a one-trial qt_mle_batch makes six.
  // Through pinned memory they are asynchronous; outputs are copied to the caller's arrays once the stream has been
  // waited for (drain_mailbox).  Transfers above kMailMax, or when the box is full, take the direct route.
  struct Pending {
    void* dst;
    const void* src;
    size_t bytes;
  };
  char* mail = nullptr;
  size_t mail_used = 0;
  std::vector<Pending> mail_pending;
  // POVM cache.  The dense operands A, A^T, A', A'^T ([M][D] each: 64 MB at n = 5) exist when `dense_ready`;
  // a product POVM at n >= 4 never reads them and builds them only on demand (ensure_dense), likewise the dense
  // left inverse (`pinv_ready`, compute_dense_pinv).
  bool povm_set = false, dense_ready = false, a_loaded = false, pinv_ready = false;
  int S = 0, K = 0, M = 0;
  DevBuf A, AT, Aw, AwT, Pinv, PinvT, Ns, aug, info;
  // packed row digits of the Kronecker assembly (k_povm_kron), cached per (S1, K1)
  DevBuf kron_dig;
  int kron_S1 = 0, kron_K1 = 0;
  // product-POVM (Kronecker) description, valid when prod.enabled
  DevBuf pr_T, pr_P1, pr_P1T, pr_wrow, pr_rmap, pr_rinv, pr_fwd, pr_bwd, pr_aug;
  qt::ProductView prod{};
  // staging for host-pointer calls
  DevBuf in0, in1, out0, out1, out2, out3, out4, out5, proc_aug, proc_ws;
  // MLE hand-off between k_mle_start and k_mle_bfgs
  DevBuf ws_x, ws_g, ws_f, ws_act;
  // BFGS (s, y) history of the n >= 4 kernels (max_iter x 2 D doubles per trial of a chunk)
  DevBuf hess;
  // radix-sort double buffer + temporary storage (qt_sort_f64)
  DevBuf sort_alt, sort_tmp;
  // process tomography
  qt::ProcessState proc;
  bool proc_set = false;
  bool proc_dense = false;  // qt_process_prefer_dense: qt_lifp_batch multiplies by the dense left inverse where it has a choice

  double ns_tot = 0.0;  // sum of the registered shots per setting
  bool check_shots = true;  // qt_set_option(QT_OPT_SHOTS_CHECK) / QTOMO_SKIP_SHOTS_CHECK=1 at qt_create
  int fused_max_waves = 1024;  // qt_set_option(QT_OPT_MLE_FUSED_MAX_WAVES): largest batch (in trial-waves) of k_mle_fused
  int lds_extra = 0;  // per-trial extra LDS doubles of the launch being prepared (k_mle_bfgs); 0 otherwise
  double ns_max = 0.0;  // largest registered shot number (product POVMs): the n >= 4 count cache holds 32-bit counts
  qt::PovmView view() const {
    return qt::PovmView{Aw.as<double>(), AwT.as<double>(), PinvT.as<double>(), M, prod, jtol2,
                        check_shots ? Ns.as<double>() : nullptr, S, K, ns_tot, lds_extra};
  }
  // Jacobi stopping rule off^2 <= jtol2 * ||A||_F^2.  Measured on the C2 batch: the last sweep takes off^2
  // from > 1e-9 to < 1e-28 in one go, so no looser threshold saves a sweep without costing accuracy.
  double jtol2 = 1e-28;
};

namespace {

// Every entry point runs on the handle's device and leaves the calling thread's current device as it found
// it: a caller that shares the process with PyTorch (one rank per GPU) must not see torch's current device
// move because an engine call happened to target another card.
struct DeviceScope {
  int prev = -1, want = -1;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int device) : want(device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != want) err = hipSetDevice(want);
  }
  ~DeviceScope() {
    if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
  }
  DeviceScope(const DeviceScope&) = delete;
  DeviceScope& operator=(const DeviceScope&) = delete;
};

#define QT_ENTER(h)                                                                                          \
  if (!(h)) return fail(QT_ERR_ARG, "null handle");                                                          \
  DeviceScope qt_scope_((h)->device);                                                                        \
  if (qt_scope_.err != hipSuccess) return fail(QT_ERR_HIP, "hipSetDevice(%d): %s", (h)->device, hipGetErrorString(qt_scope_.err)); \
  /* outputs a FAILED host-pointer call left parked in the mailbox must never be copied into that caller's (possibly */   \
  /* freed) arrays by the next call's drain: a successful call has drained before it returned */                          \
  (h)->mail_pending.clear();                                                                                               \
  (h)->mail_used = 0

inline int grid_for(size_t total, int block = 256, int cap = 8192) {
  size_t g = (total + block - 1) / block;
  if (g < 1) g = 1;
  if (g > (size_t)cap) g = cap;
  return (int)g;
}

constexpr size_t kMailCap = 1 << 20, kMailMax = 128 << 10;
inline char* mail_alloc(qt_handle_t* h, size_t bytes) {
  if (bytes == 0 || bytes > kMailMax) return nullptr;
  if (!h->mail && hipHostMalloc(reinterpret_cast<void**>(&h->mail), kMailCap, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    h->mail = nullptr;
    return nullptr;
  }
  const size_t at = (h->mail_used + 63) & ~(size_t)63;
  if (at + bytes > kMailCap) return nullptr;
  h->mail_used = at + bytes;
  return h->mail + at;
}
// after the stream has been waited for: hand the outputs parked in the mailbox to the caller's arrays
inline void drain_mailbox(qt_handle_t* h) {
  for (const auto& p : h->mail_pending) memcpy(p.dst, p.src, p.bytes);
  h->mail_pending.clear();
  h->mail_used = 0;
}

// Resolve an input array: device pointer as-is, or staged copy of a host array.
template <class T>
int stage_in(qt_handle_t* h, DevBuf& buf, const T* src, size_t count, int flags, const T** out) {
  if (flags & QT_DEVICE_PTR) {
    *out = src;
    return 0;
  }
  HIPCHK(buf.ensure(count * sizeof(T)));
  const void* from = src;
  if (char* m = mail_alloc(h, count * sizeof(T))) {
    memcpy(m, src, count * sizeof(T));
    from = m;
  }
  HIPCHK(hipMemcpyAsync(buf.p, from, count * sizeof(T), hipMemcpyHostToDevice, h->stream));
  *out = buf.as<T>();
  return 0;
}
template <class T>
int stage_out(qt_handle_t*, DevBuf& buf, T* dst, size_t count, int flags, T** out) {
  if (!dst) {
    *out = nullptr;
    return 0;
  }
  if (flags & QT_DEVICE_PTR) {
    *out = dst;
    return 0;
  }
  HIPCHK(buf.ensure(count * sizeof(T)));
  *out = buf.as<T>();
  return 0;
}
template <class T>
int fetch_out(qt_handle_t* h, const T* dev, T* dst, size_t count, int flags) {
  if (!dst || (flags & QT_DEVICE_PTR)) return 0;
  if (char* m = mail_alloc(h, count * sizeof(T))) {
    HIPCHK(hipMemcpyAsync(m, dev, count * sizeof(T), hipMemcpyDeviceToHost, h->stream));
    h->mail_pending.push_back({dst, m, count * sizeof(T)});
    return 0;
  }
  HIPCHK(hipMemcpyAsync(dst, dev, count * sizeof(T), hipMemcpyDeviceToHost, h->stream));
  return 0;
}
// Wait for the handle's stream.  hipStreamSynchronize / hipEventSynchronize park the thread and wake it through an
// interrupt: ~50 us of latency measured around a 330 us timed region (20 steps of bench.py) and on every host-pointer
// call.  Most waits here are shorter than a millisecond, so: record an event, poll it for up to ~2 ms, then block.
int wait_event_spin(hipEvent_t ev) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipEventQuery(ev);
    if (q == hipSuccess) return 0;
    if (q != hipErrorNotReady) return fail(QT_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(q));
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
  }
  HIPCHK(hipEventSynchronize(ev));
  return 0;
}
int wait_stream(qt_handle_t* h) {
  // poll the stream itself (no event packet to push through the queue first: an idle wait costs ~2 us instead of ~12)
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(h->stream);
    if (q == hipSuccess) {
      drain_mailbox(h);
      return 0;
    }
    if (q != hipErrorNotReady) return fail(QT_ERR_HIP, "hipStreamQuery: %s", hipGetErrorString(q));
    if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
  }
  HIPCHK(hipStreamSynchronize(h->stream));
  drain_mailbox(h);
  return 0;
}
// for the entry points that block on the stream themselves
#define QT_STREAM_SYNC(h)                       \
  do {                                          \
    HIPCHK(hipStreamSynchronize((h)->stream));  \
    drain_mailbox(h);                           \
  } while (0)
int finish(qt_handle_t* h, int flags) {
  HIPCHK(hipGetLastError());
  if (!(flags & QT_DEVICE_PTR)) return wait_stream(h);
  return 0;
}
int count_bad(const int32_t* status, int B, int flags) {
  if (!status || (flags & QT_DEVICE_PTR)) return 0;
  int bad = 0;
  for (int b = 0; b < B; ++b) bad += status[b] != 0;
  return bad;
}

constexpr size_t kLdsLimit = 160 * 1024;  // LDS per CU on gfx950; one workgroup may use all of it
// Measured (profiles/round1_v3_*): a lone wave per SIMD pays ~7-10 ns of issue per LDS read, more than
// for a global load that lands asynchronously, and the image costs occupancy at large batch: the L2
// streaming variant is faster in both regimes, so the image variant is kept but not selected.
#ifdef QT_PREFER_LDS_IMAGE
constexpr bool kPreferLdsImage = true;
#endif

// Launch KERNEL<NQ, ALDS> for the handle's n: the LDS-image variant when image + scratch fit in
// 160 KB, else the variant that streams the operand from L2.  ARGS is the parenthesised argument list.
template <class K>
int allow_big_lds(K kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)bytes);
  if (e != hipSuccess) return fail(QT_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize=%zu): %s", bytes, hipGetErrorString(e));
  return 0;
}

#define QT_LAUNCH_ONE(KERNEL, NQV, ALDSV, M_, B_, ARGS)                                                   \
  do {                                                                                                    \
    using S_ = qt::Small<NQV, ALDSV>;                                                                     \
    const size_t lds_ = S_::lds_bytes(M_, h->prod.enabled ? h->prod.R1 : 0, h->lds_extra);                \
    if (lds_ > kLdsLimit) return fail(QT_ERR_UNSUPPORTED, "POVM too large for the n<=3 kernels (%zu B of LDS)", lds_); \
    if (int r_ = allow_big_lds(KERNEL<NQV, ALDSV>, lds_)) return r_;                                      \
    const int grid_ = ((B_) + S_::TPB - 1) / S_::TPB;                                                     \
    hipLaunchKernelGGL((KERNEL<NQV, ALDSV>), dim3(grid_), dim3(S_::NT), lds_, h->stream, QT_UNPACK ARGS); \
  } while (0)
#define QT_UNPACK(...) __VA_ARGS__
#ifdef QT_PREFER_LDS_IMAGE  /* the measured-slower variant: instantiated only on request (a third of the build time) */
#define QT_LAUNCH_N(KERNEL, NQV, M_, B_, ARGS)                          \
  do {                                                                  \
    if (kPreferLdsImage && qt::Small<NQV, true>::lds_bytes(M_, h->prod.enabled ? h->prod.R1 : 0) <= kLdsLimit) \
      QT_LAUNCH_ONE(KERNEL, NQV, true, M_, B_, ARGS);                   \
    else                                                                \
      QT_LAUNCH_ONE(KERNEL, NQV, false, M_, B_, ARGS);                  \
  } while (0)
#else
#define QT_LAUNCH_N(KERNEL, NQV, M_, B_, ARGS) QT_LAUNCH_ONE(KERNEL, NQV, false, M_, B_, ARGS)
#endif
#define QT_LAUNCH_SMALL(KERNEL, M_, B_, ARGS)                                                               \
  switch (h->nq) {                                                                                          \
    case 1: QT_LAUNCH_N(KERNEL, 1, M_, B_, ARGS); break;                                                    \
    case 2: QT_LAUNCH_N(KERNEL, 2, M_, B_, ARGS); break;                                                    \
    case 3: QT_LAUNCH_N(KERNEL, 3, M_, B_, ARGS); break;                                                    \
    default: return fail(QT_ERR_UNSUPPORTED, "estimators support n_qubits 1..3 in this release (got %d)", h->nq); \
  }
// kernels without an operand image (Cholesky parametrisation)
#define QT_LAUNCH_SMALL_NOIMG(KERNEL, B_, ARGS)                                                             \
  switch (h->nq) {                                                                                          \
    case 1: { using S_ = qt::Small<1, false>; hipLaunchKernelGGL((KERNEL<1>), dim3(((B_) + S_::TPB - 1) / S_::TPB), dim3(S_::NT), S_::lds_bytes(0), h->stream, QT_UNPACK ARGS); } break; \
    case 2: { using S_ = qt::Small<2, false>; hipLaunchKernelGGL((KERNEL<2>), dim3(((B_) + S_::TPB - 1) / S_::TPB), dim3(S_::NT), S_::lds_bytes(0), h->stream, QT_UNPACK ARGS); } break; \
    case 3: { using S_ = qt::Small<3, false>; hipLaunchKernelGGL((KERNEL<3>), dim3(((B_) + S_::TPB - 1) / S_::TPB), dim3(S_::NT), S_::lds_bytes(0), h->stream, QT_UNPACK ARGS); } break; \
    default: return fail(QT_ERR_UNSUPPORTED, "estimators support n_qubits 1..3 in this release (got %d)", h->nq); \
  }

// ---- n = 4, 5: workgroup-per-trial kernels (qt_large.h) ---------------------------------------------
int compute_dense_pinv_fwd(qt_handle_t* h);
// n >= 4: what the launch about to be made reads besides the factorised tables.  A plain tensor (qt_set_povm) has its
// dense operands and left inverse already; a product POVM with unequal shots needs the dense left inverse for 'lin'.
int prepare_large(qt_handle_t* h, bool needs_lin) {
  if (h->prod.enabled && needs_lin && !h->prod.uniform && !h->pinv_ready) return compute_dense_pinv_fwd(h);
  return 0;
}
#define QT_LAUNCH_LARGE(KERNEL, B_, M_, R1_, ARGS) QT_LAUNCH_LARGE_X(KERNEL, B_, M_, R1_, 0, ARGS)
// h->lds_extra (-> PovmView::extra) = offset in doubles, while ARGS (h->view()) is evaluated, of a block at the end of
// the LDS allocation that holds a 32-bit copy of the trial's counts in R-order (Large::make_ctx): taken whenever it fits next to everything else and the registered
// shots fit 32 bits.
#define QT_LAUNCH_LARGE_X(KERNEL, B_, M_, R1_, XTRA_, ARGS)                                                 \
  do {                                                                                                      \
    if (h->nq == 4) {                                                                                       \
      size_t lds_ = qt::Large<4>::lds_bytes(M_, R1_, XTRA_);                                                \
      if (lds_ > kLdsLimit) return fail(QT_ERR_UNSUPPORTED, "POVM too large for LDS (%zu B)", lds_);         \
      const bool cache_ = false; /* n = 4: measured slower with the cache (0.122 vs 0.101 ms per 1024 'mle') */ \
      h->lds_extra = 0;                                                                                     \
      if (int r_ = allow_big_lds(KERNEL<4>, lds_)) return r_;                                               \
      hipLaunchKernelGGL((KERNEL<4>), dim3(B_), dim3(qt::Large<4>::NT), lds_, h->stream, QT_UNPACK ARGS);   \
      h->lds_extra = 0;                                                                                     \
    } else {                                                                                                \
      size_t lds_ = qt::Large<5>::lds_bytes(M_, R1_, XTRA_);                                                \
      if (lds_ > kLdsLimit) return fail(QT_ERR_UNSUPPORTED, "POVM too large for LDS (%zu B)", lds_);         \
      const bool cache_ = h->prod.enabled && h->ns_max < 4294967296.0 && lds_ + 4 * (size_t)(M_) + 8 <= kLdsLimit; \
      if (cache_) lds_ += 4 * (size_t)(M_) + 8;                                                             \
      h->lds_extra = cache_ ? (int)((lds_ - 4 * (size_t)(M_) - 8) / 8) : 0;                                 \
      if (int r_ = allow_big_lds(KERNEL<5>, lds_)) return r_;                                               \
      hipLaunchKernelGGL((KERNEL<5>), dim3(B_), dim3(qt::Large<5>::NT), lds_, h->stream, QT_UNPACK ARGS);   \
      h->lds_extra = 0;                                                                                     \
    }                                                                                                       \
  } while (0)

// In-place inverse of the n x n matrix in the left half of aug [n][2n]: one workgroup up to n = 127, the
// chip-wide variant (qt_ops.h) beyond -- the 256 x 256 complex Gram matrix of 2-qubit process tomography takes
// 12.5 ms in one workgroup and 1.8 ms as 2 x 256 small launches.
template <int CPLX>
void launch_gauss_jordan(qt_handle_t* h, int n, double* aug, int* info) {
  if (n < 128) {
    hipLaunchKernelGGL(qt::k_gauss_jordan<CPLX>, dim3(1), dim3(1024), 0, h->stream, n, aug, info);
    return;
  }
  hipLaunchKernelGGL(qt::k_gj_identity<CPLX>, dim3(grid_for((size_t)n * n)), dim3(256), 0, h->stream, n, aug, info);
  for (int k = 0; k < n; ++k) {
    hipLaunchKernelGGL(qt::k_gj_pivot<CPLX>, dim3(1), dim3(1024), 0, h->stream, n, aug, k, info);
    hipLaunchKernelGGL(qt::k_gj_eliminate<CPLX>, dim3(n), dim3(256), 0, h->stream, n, aug, k);
  }
}

template <int W>
void launch_transpose(qt_handle_t* h, const double* in, int R, int C, double* out) {
  constexpr int TS = 64 / W;
  hipLaunchKernelGGL(qt::k_transpose_tiled<W>, dim3((C + TS - 1) / TS, (R + TS - 1) / TS), dim3(256), 0, h->stream, in, R, C, out);
}

// out[cols][rows] = inv(A^T A) A^T of the complex A[rows][cols] (plain transposes, routines.py:69-71); all device
// pointers, enqueued on the handle's stream; the pivot report lands in h->info (0 = regular).
int enqueue_left_inverse_complex(qt_handle_t* h, const double* dA, int rows, int cols, double* dout) {
  DevBuf& aug = h->proc_aug;
  HIPCHK(aug.ensure((size_t)cols * 2 * cols * 2 * sizeof(double)));
  HIPCHK(h->info.ensure(sizeof(int)));
  double* g = aug.as<double>();
  dim3 gg((cols + 15) / 16, (cols + 15) / 16), gp((rows + 15) / 16, (cols + 15) / 16);
  hipLaunchKernelGGL(qt::k_gemm<1>, gg, dim3(64), 0, h->stream, cols, cols, rows, dA, cols, 1, dA, cols, 0, g, 2 * cols);
  launch_gauss_jordan<1>(h, cols, g, h->info.as<int>());
  hipLaunchKernelGGL(qt::k_gemm<1>, gp, dim3(64), 0, h->stream, cols, rows, cols, g + (size_t)cols * 2, 2 * cols, 0, dA, cols,
                     1, dout, rows);
  return 0;
}

int need_povm(qt_handle_t* h) {
  if (!h->povm_set) return fail(QT_ERR_STATE, "qt_set_povm has not been called on this handle");
  return 0;
}

}  // namespace

extern "C" {

int qt_version(void) { return 100; }

const char* qt_last_error(void) { return g_err.c_str(); }

int qt_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return fail(QT_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
  return n;
}

qt_handle_t* qt_create(int device, int n_qubits) {
  if (n_qubits < 1 || n_qubits > 5) {
    fail(QT_ERR_ARG, "n_qubits must be in 1..5 (got %d)", n_qubits);
    return nullptr;
  }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    fail(QT_ERR_HIP, "no usable HIP device (%s); this library has no CPU path", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    return nullptr;
  }
  if (device < 0 || device >= n) {
    fail(QT_ERR_ARG, "device %d out of range (have %d)", device, n);
    return nullptr;
  }
  DeviceScope scope(device);
  if (scope.err != hipSuccess) {
    fail(QT_ERR_HIP, "hipSetDevice: %s", hipGetErrorString(scope.err));
    return nullptr;
  }
  qt_handle_t* h = new qt_handle();
  h->device = device;
  h->nq = n_qubits;
  h->d = 1 << n_qubits;
  h->D = h->d * h->d;
  if ((e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreate(&h->ev0)) != hipSuccess || (e = hipEventCreate(&h->ev1)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&h->ev_sync, hipEventDisableTiming)) != hipSuccess) {
    fail(QT_ERR_HIP, "stream/event creation: %s", hipGetErrorString(e));
    delete h;
    return nullptr;
  }
  h->own_stream = true;
  if (const char* env = getenv("QTOMO_SKIP_SHOTS_CHECK")) h->check_shots = !(env[0] == '1');
  return h;
}

void qt_destroy(qt_handle_t* h) {
  if (!h) return;
  DeviceScope scope(h->device);
  (void)hipStreamSynchronize(h->stream);
  for (DevBuf* b : {&h->pr_T, &h->pr_P1, &h->pr_P1T, &h->pr_wrow, &h->pr_rmap, &h->pr_rinv, &h->pr_fwd, &h->pr_bwd, &h->pr_aug})
    b->release();
  for (DevBuf* b : {&h->A, &h->AT, &h->Aw, &h->AwT, &h->Pinv, &h->PinvT, &h->Ns, &h->aug, &h->info, &h->kron_dig, &h->in0, &h->in1,
                    &h->out0, &h->out1, &h->out2, &h->out3, &h->out4, &h->out5, &h->proc_aug, &h->proc_ws, &h->ws_x, &h->ws_g, &h->ws_f,
                    &h->ws_act, &h->hess, &h->sort_alt, &h->sort_tmp})
    b->release();
  h->proc.release();
  if (h->mail) (void)hipHostFree(h->mail);
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->ev_sync) (void)hipEventDestroy(h->ev_sync);
  if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int qt_sync(qt_handle_t* h) {
  QT_ENTER(h);
  return wait_stream(h);
}

int qt_set_stream(qt_handle_t* h, void* hip_stream) {
  QT_ENTER(h);
  QT_STREAM_SYNC(h);
  if (h->own_stream && h->stream) HIPCHK(hipStreamDestroy(h->stream));
  if (hip_stream) {
    h->stream = hip_stream == QT_STREAM_LEGACY ? hipStreamLegacy : static_cast<hipStream_t>(hip_stream);
    h->own_stream = false;
  } else {
    HIPCHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->own_stream = true;
  }
  return 0;
}

int qt_set_option(qt_handle_t* h, int option, double value) {
  QT_ENTER(h);
  switch (option) {
    case QT_OPT_SHOTS_CHECK: h->check_shots = value != 0.0; return 0;
    case QT_OPT_MLE_FUSED_MAX_WAVES:
      if (!(value >= 0.0 && value <= 1048576.0)) return fail(QT_ERR_ARG, "QT_OPT_MLE_FUSED_MAX_WAVES out of range");
      h->fused_max_waves = (int)value;
      return 0;
    default: return fail(QT_ERR_ARG, "unknown option %d", option);
  }
}

int qt_timer_begin(qt_handle_t* h) {
  QT_ENTER(h);
  HIPCHK(hipEventRecord(h->ev0, h->stream));
  return 0;
}

// the end event only (asynchronous): a caller that synchronises anyway reads the interval afterwards
int qt_timer_stop(qt_handle_t* h) {
  QT_ENTER(h);
  HIPCHK(hipEventRecord(h->ev1, h->stream));
  return 0;
}

int qt_timer_elapsed(qt_handle_t* h, double* elapsed_ms) {
  QT_ENTER(h);
  if (!elapsed_ms) return fail(QT_ERR_ARG, "null elapsed_ms");
  if (int r = wait_event_spin(h->ev1)) return r;
  float ms = 0.f;
  HIPCHK(hipEventElapsedTime(&ms, h->ev0, h->ev1));
  *elapsed_ms = ms;
  return 0;
}

int qt_timer_end(qt_handle_t* h, double* elapsed_ms) {
  if (int r = qt_timer_stop(h)) return r;
  return qt_timer_elapsed(h, elapsed_ms);
}

int qt_pauli_basis(qt_handle_t* h, double* out, int flags) {
  QT_ENTER(h);
  if (!out) return fail(QT_ERR_ARG, "null out");
  const size_t n = (size_t)h->D * h->D * 2;
  double* dout;
  if (int r = stage_out(h, h->out0, out, n, flags, &dout)) return r;
  hipLaunchKernelGGL(qt::k_pauli_basis, dim3(grid_for(n / 2)), dim3(256), 0, h->stream, h->nq, dout);
  if (int r = fetch_out(h, dout, out, n, flags)) return r;
  return finish(h, flags);
}

// Row digits for k_povm_kron: row = s K + k with s = sum_q s_q S1^(n-1-q), k likewise; byte q = s_q K1 + k_q.
static int ensure_kron_digits(qt_handle_t* h, int S1, int K1) {
  if (h->kron_S1 == S1 && h->kron_K1 == K1 && h->kron_dig.p) return 0;
  const int n = h->nq;
  long long S = 1, K = 1;
  for (int q = 0; q < n; ++q) {
    S *= S1;
    K *= K1;
  }
  std::vector<unsigned long long> dig((size_t)(S * K));
  for (long long s = 0; s < S; ++s)
    for (long long k = 0; k < K; ++k) {
      unsigned long long pack = 0;
      long long sr = s, kr = k;
      for (int q = n - 1; q >= 0; --q) {
        pack |= (unsigned long long)((sr % S1) * K1 + (kr % K1)) << (8 * q);
        sr /= S1;
        kr /= K1;
      }
      dig[(size_t)(s * K + k)] = pack;
    }
  h->kron_S1 = h->kron_K1 = 0;
  HIPCHK(h->kron_dig.ensure(dig.size() * sizeof(unsigned long long)));
  HIPCHK(hipMemcpyAsync(h->kron_dig.p, dig.data(), dig.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, h->stream));
  QT_STREAM_SYNC(h);  // `dig` goes out of scope
  h->kron_S1 = S1;
  h->kron_K1 = K1;
  return 0;
}

static int launch_povm_kron(qt_handle_t* h, const double* dtable, int S1, int K1, double* dout) {
  if (S1 * K1 > 255) return fail(QT_ERR_UNSUPPORTED, "one-qubit table with %d rows (> 255)", S1 * K1);
  if (int r = ensure_kron_digits(h, S1, K1)) return r;
  size_t total = (size_t)h->D;
  for (int q = 0; q < h->nq; ++q) total *= (size_t)S1 * K1;
  if ((total >> (2 * h->nq)) > ((size_t)1 << 31)) return fail(QT_ERR_UNSUPPORTED, "POVM tensor too large");
  // >> 256 workgroups, each lane a 16-byte store per pass
  hipLaunchKernelGGL(qt::k_povm_kron, dim3(grid_for(total / 2, 256, 4096)), dim3(256), 0, h->stream, h->nq, dtable, S1 * K1,
                     h->kron_dig.as<unsigned long long>(), total, dout);
  return 0;
}

int qt_povm_kron(qt_handle_t* h, const double* povm1, int S1, int K1, double* out, int flags) {
  QT_ENTER(h);
  if (!povm1 || !out || S1 < 1 || K1 < 1) return fail(QT_ERR_ARG, "bad povm_kron arguments");
  size_t S = 1, K = 1;
  for (int q = 0; q < h->nq; ++q) {
    S *= S1;
    K *= K1;
  }
  const size_t n = S * K * h->D;
  const double* din;
  double* dout;
  if (int r = stage_in(h, h->in0, povm1, (size_t)S1 * K1 * 4, flags, &din)) return r;
  if (int r = stage_out(h, h->out0, out, n, flags, &dout)) return r;
  if (int r = launch_povm_kron(h, din, S1, K1, dout)) return r;
  if (int r = fetch_out(h, dout, out, n, flags)) return r;
  return finish(h, flags);
}

// Dense operands A ([M][D]; the Kronecker power of the table for a product POVM), A^T, A', A'^T -- built when first
// needed: Born kernel / dense estimators at n <= 3, process set-up, the dense left inverse.
static int ensure_dense(qt_handle_t* h) {
  if (h->dense_ready) return 0;
  const size_t bytes = (size_t)h->M * h->D * sizeof(double);
  HIPCHK(h->A.ensure(bytes));
  HIPCHK(h->AT.ensure(bytes));
  HIPCHK(h->Aw.ensure(bytes));
  HIPCHK(h->AwT.ensure(bytes));
  if (!h->a_loaded) {
    if (!h->pr_T.p || h->kron_S1 * h->kron_K1 == 0) return fail(QT_ERR_STATE, "no POVM tensor to build the dense operands from");
    if (int r = launch_povm_kron(h, h->pr_T.as<double>(), h->kron_S1, h->kron_K1, h->A.as<double>())) return r;
    h->a_loaded = true;
  }
  hipLaunchKernelGGL(qt::k_povm_setup, dim3((h->D + 63) / 64, (h->M + 63) / 64), dim3(256), 0, h->stream, h->A.as<double>(),
                     h->Ns.as<double>(), h->ns_tot, h->K, h->M, h->D, h->AT.as<double>(), h->Aw.as<double>(),
                     h->AwT.as<double>());
  HIPCHK(hipGetLastError());
  h->dense_ready = true;
  return 0;
}

// Dense left inverse inv(A'^T A') A'^T of the cached weighted POVM (Gram GEMM, pivoted Gauss-Jordan, GEMM).
static int compute_dense_pinv(qt_handle_t* h) {
  if (int r = ensure_dense(h)) return r;
  const int D = h->D, M = h->M;
  const size_t bytes = (size_t)M * D * sizeof(double);
  HIPCHK(h->Pinv.ensure(bytes));
  HIPCHK(h->PinvT.ensure(bytes));
  HIPCHK(h->aug.ensure((size_t)D * 2 * D * sizeof(double)));
  double *dAw = h->Aw.as<double>(), *dAwT = h->AwT.as<double>();
  double *dP = h->Pinv.as<double>(), *dPT = h->PinvT.as<double>(), *aug = h->aug.as<double>();
  dim3 gg((D + 15) / 16, (D + 15) / 16);
  hipLaunchKernelGGL(qt::k_gemm<0>, gg, dim3(64), 0, h->stream, D, D, M, dAw, D, 1, dAw, D, 0, aug, 2 * D);
  launch_gauss_jordan<0>(h, D, aug, h->info.as<int>());
  dim3 gp((M + 15) / 16, (D + 15) / 16);
  hipLaunchKernelGGL(qt::k_gemm<0>, gp, dim3(64), 0, h->stream, D, M, D, aug + D, 2 * D, 0, dAwT, M, 0, dP, M);
  launch_transpose<1>(h, dP, D, M, dPT);
  int info = 0;
  HIPCHK(hipMemcpyAsync(&info, h->info.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipGetLastError());
  QT_STREAM_SYNC(h);
  if (info != 0) return fail(QT_ERR_SINGULAR, "A^T A is singular (no pivot in column %d): POVM not informationally complete", info - 1);
  h->pinv_ready = true;
  return 0;
}

}  // extern "C"
namespace {
int compute_dense_pinv_fwd(qt_handle_t* h) { return compute_dense_pinv(h); }
}  // namespace
extern "C" {

// Start of qt_set_povm / qt_set_povm_product: forget the previous POVM, record the shape.
static int begin_povm(qt_handle_t* h, int S, int K) {
  const size_t M = (size_t)S * K;
  if (M < (size_t)h->D) return fail(QT_ERR_SINGULAR, "POVM has %zu rows < D = %d: not informationally complete", M, h->D);
  h->povm_set = false;
  h->proc_set = false;
  h->dense_ready = h->a_loaded = h->pinv_ready = false;
  h->prod = qt::ProductView{};
  h->S = S;
  h->K = K;
  h->M = (int)M;
  HIPCHK(h->Ns.ensure(S * sizeof(double)));
  HIPCHK(h->info.ensure(sizeof(int)));
  return 0;
}

int qt_set_povm(qt_handle_t* h, const double* A, int S, int K, const double* Ns, int flags) {
  QT_ENTER(h);
  if (!A || !Ns || S < 1 || K < 1) return fail(QT_ERR_ARG, "bad set_povm arguments");
  if (int r = begin_povm(h, S, K)) return r;
  const hipMemcpyKind kind = (flags & QT_DEVICE_PTR) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  HIPCHK(h->A.ensure((size_t)S * K * h->D * sizeof(double)));
  HIPCHK(hipMemcpyAsync(h->A.p, A, (size_t)S * K * h->D * sizeof(double), kind, h->stream));
  HIPCHK(hipMemcpyAsync(h->Ns.p, Ns, S * sizeof(double), kind, h->stream));
  h->a_loaded = true;
  {
    std::vector<double> ns((size_t)S);
    if (flags & QT_DEVICE_PTR) HIPCHK(hipMemcpy(ns.data(), Ns, ns.size() * sizeof(double), hipMemcpyDeviceToHost));
    else memcpy(ns.data(), Ns, ns.size() * sizeof(double));
    h->ns_tot = 0.0;
    for (double v : ns) h->ns_tot += v;
  }
  if (int r = compute_dense_pinv(h)) return r;  // a plain tensor: the dense operands ARE the POVM
  h->povm_set = true;
  return 0;
}

int qt_set_povm_product(qt_handle_t* h, const double* povm1, int S1, int K1, const double* Ns, int flags) {
  QT_ENTER(h);
  if (!povm1 || !Ns || S1 < 1 || K1 < 1) return fail(QT_ERR_ARG, "bad set_povm_product arguments");
  const int n = h->nq, R1 = S1 * K1;
  long long S = 1, K = 1, M = 1;
  for (int q = 0; q < n; ++q) {
    S *= S1;
    K *= K1;
    M *= R1;
  }
  if (M > (1 << 15)) return fail(QT_ERR_UNSUPPORTED, "product POVM with %lld rows is too large", M);
  if (R1 > 255) return fail(QT_ERR_UNSUPPORTED, "one-qubit table with %d rows (> 255)", R1);
  if (int r = begin_povm(h, (int)S, (int)K)) return r;
  // host copies of the small inputs (table and shots) for the index tables
  std::vector<double> t1((size_t)R1 * 4), ns((size_t)S);
  if (flags & QT_DEVICE_PTR) {
    HIPCHK(hipMemcpy(t1.data(), povm1, t1.size() * sizeof(double), hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(ns.data(), Ns, ns.size() * sizeof(double), hipMemcpyDeviceToHost));
  } else {
    memcpy(t1.data(), povm1, t1.size() * sizeof(double));
    memcpy(ns.data(), Ns, ns.size() * sizeof(double));
  }
  HIPCHK(h->pr_T.ensure(t1.size() * sizeof(double)));
  HIPCHK(hipMemcpyAsync(h->pr_T.p, t1.data(), t1.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->Ns.p, ns.data(), ns.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  double tot = 0.0;
  bool uniform = true;
  for (long long s = 0; s < S; ++s) {
    tot += ns[s];
    if (ns[s] != ns[0]) uniform = false;
  }
  h->ns_tot = tot;
  if (int r = ensure_kron_digits(h, S1, K1)) return r;
  // The dense operands (a2 tensor + transposes; Born GEMM, dense fallbacks, process set-up) are cheap at n <= 3 and
  // built now; at n = 4, 5 (6 x 64 MB at n = 5) the factorised estimators never read them: on demand only.
  // The dense left inverse is needed by 'lin' when the shots differ between settings (n <= 3).
  if (n <= 3) {
    if (int r = uniform ? ensure_dense(h) : compute_dense_pinv(h)) return r;
  }
  // pinv of the one-qubit table, on the device: inv(T^T T) T^T  ([4][R1]) and its transpose
  HIPCHK(h->pr_P1.ensure((size_t)4 * R1 * sizeof(double)));
  HIPCHK(h->pr_P1T.ensure((size_t)4 * R1 * sizeof(double)));
  HIPCHK(h->pr_aug.ensure((size_t)4 * 8 * sizeof(double)));
  {
    double *T = h->pr_T.as<double>(), *g = h->pr_aug.as<double>(), *P1 = h->pr_P1.as<double>();
    hipLaunchKernelGGL(qt::k_gemm<0>, dim3(1, 1), dim3(64), 0, h->stream, 4, 4, R1, T, 4, 1, T, 4, 0, g, 8);
    hipLaunchKernelGGL(qt::k_gauss_jordan<0>, dim3(1), dim3(1024), 0, h->stream, 4, g, h->info.as<int>());
    hipLaunchKernelGGL(qt::k_gemm<0>, dim3((R1 + 15) / 16, 1), dim3(64), 0, h->stream, 4, R1, 4, g + 4, 8, 0, T, 4, 1, P1,
                       R1);
    launch_transpose<1>(h, P1, 4, R1, h->pr_P1T.as<double>());
  }
  // host-side index bookkeeping: R-order row map, shot weights, stage tables
  std::vector<int> rmap((size_t)M), fwd, bwd;
  std::vector<double> wrow((size_t)M);
  for (long long mr = 0; mr < M; ++mr) {  // mr = [r_1 .. r_n], r_q = s_q K1 + o_q
    long long rem = mr, s = 0, o = 0, sp = 1, op = 1;
    for (int q = n - 1; q >= 0; --q) {
      const int r = (int)(rem % R1);
      rem /= R1;
      s += (r / K1) * sp;
      o += (r % K1) * op;
      sp *= S1;
      op *= K1;
    }
    rmap[mr] = (int)(s * K + o);
    wrow[mr] = ns[s] / tot;
  }
  auto ipow = [](long long b, int e) {
    long long r = 1;
    for (int i = 0; i < e; ++i) r *= b;
    return r;
  };
  for (int q = 1; q <= n; ++q) {  // forward stage q: out[r_1..r_q][k_(q+1)..k_n]
    const long long Kq = ipow(4, n - q), n_out = ipow(R1, q) * Kq;
    for (long long o = 0; o < n_out; ++o) {
      const long long rpre = o / (R1 * Kq), rq = (o / Kq) % R1, krest = o % Kq;
      fwd.push_back((int)((rpre * 4 * Kq + krest) | (rq << 16)));
    }
  }
  for (int q = n; q >= 1; --q) {  // backward stage q: out[r_1..r_(q-1)][k_q..k_n]
    const long long Kq = ipow(4, n - q), n_out = ipow(R1, q - 1) * 4 * Kq;
    for (long long o = 0; o < n_out; ++o) {
      const long long rpre = o / (4 * Kq), kq = (o / Kq) % 4, krest = o % Kq;
      bwd.push_back((int)((rpre * R1 * Kq + krest) | (kq << 16)));
    }
  }
  std::vector<int> rinv((size_t)M);
  for (long long mr = 0; mr < M; ++mr) rinv[(size_t)rmap[mr]] = (int)mr;
  HIPCHK(h->pr_rinv.ensure(rinv.size() * sizeof(int)));
  HIPCHK(hipMemcpyAsync(h->pr_rinv.p, rinv.data(), rinv.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(h->pr_rmap.ensure(rmap.size() * sizeof(int)));
  HIPCHK(h->pr_wrow.ensure(wrow.size() * sizeof(double)));
  HIPCHK(h->pr_fwd.ensure(fwd.size() * sizeof(int)));
  HIPCHK(h->pr_bwd.ensure(bwd.size() * sizeof(int)));
  HIPCHK(hipMemcpyAsync(h->pr_rmap.p, rmap.data(), rmap.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->pr_wrow.p, wrow.data(), wrow.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->pr_fwd.p, fwd.data(), fwd.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  HIPCHK(hipMemcpyAsync(h->pr_bwd.p, bwd.data(), bwd.size() * sizeof(int), hipMemcpyHostToDevice, h->stream));
  int info = 0;
  HIPCHK(hipMemcpyAsync(&info, h->info.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPCHK(hipGetLastError());
  QT_STREAM_SYNC(h);
  if (info != 0) return fail(QT_ERR_SINGULAR, "the one-qubit table is not informationally complete");
  h->prod.T = h->pr_T.as<double>();
  h->prod.P1T = h->pr_P1T.as<double>();
  h->prod.wrowR = h->pr_wrow.as<double>();
  h->prod.rmap = h->pr_rmap.as<int>();
  h->prod.rinv = h->pr_rinv.as<int>();
  h->ns_max = 0.0;
  for (double v : ns) h->ns_max = v > h->ns_max ? v : h->ns_max;
  h->prod.fwd = h->pr_fwd.as<int>();
  h->prod.bwd = h->pr_bwd.as<int>();
  h->prod.R1 = R1;
  h->prod.uniform = uniform ? 1 : 0;
  h->prod.wuni = ns[0] / tot;
  h->prod.enabled = 1;
  h->povm_set = true;
  return 0;
}

int qt_get_left_inverse(qt_handle_t* h, double* out, int flags) {
  QT_ENTER(h);
  if (int r = need_povm(h)) return r;
  if (!out) return fail(QT_ERR_ARG, "null out");
  if (!h->pinv_ready)
    if (int r = compute_dense_pinv(h)) return r;
  const size_t bytes = (size_t)h->D * h->M * sizeof(double);
  HIPCHK(hipMemcpyAsync(out, h->Pinv.p, bytes, (flags & QT_DEVICE_PTR) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                        h->stream));
  return finish(h, flags);
}

int qt_born_probs(qt_handle_t* h, const double* bloch, int B, double* p, int flags) {
  QT_ENTER(h);
  if (int r = need_povm(h)) return r;
  if (B < 0 || (B > 0 && (!bloch || !p))) return fail(QT_ERR_ARG, "bad born_probs arguments");
  if (B == 0) return 0;
  const double* din;
  double* dout;
  if (int r = stage_in(h, h->in0, bloch, (size_t)B * h->D, flags, &din)) return r;
  if (int r = stage_out(h, h->out0, p, (size_t)B * h->M, flags, &dout)) return r;
  if (h->nq >= 4 && h->prod.enabled) {  // factorised contraction, one workgroup per state
    QT_LAUNCH_LARGE(qt::k_born_large, B, h->M, h->prod.R1, (h->view(), din, B, dout));
    if (int r = fetch_out(h, dout, p, (size_t)B * h->M, flags)) return r;
    return finish(h, flags);
  }
  if (int r = ensure_dense(h)) return r;
  const int gx = (h->M + 255) / 256;
  int Mp = (h->M + 15) & ~15;
  if ((Mp & 31) != 16) Mp += 16;  // LDS pitch: 16 mod 32 doubles (see k_born_mfma)
  const size_t at_bytes = (size_t)h->D * Mp * sizeof(double);
  if (h->D <= 64 && at_bytes <= 128 * 1024 && B >= 4096) {
    // batched: the matrix-core kernel, one persistent 16-wave workgroup per CU (A^T lives in its LDS)
    int grid = (B + 16 * 16 - 1) / (16 * 16);
    if (grid > 256) grid = 256;
    switch (h->D) {
      case 4:
        if (int r = allow_big_lds(qt::k_born_mfma<4>, at_bytes)) return r;
        hipLaunchKernelGGL(qt::k_born_mfma<4>, dim3(grid), dim3(1024), at_bytes, h->stream, h->AT.as<double>(), h->M, Mp, h->d,
                           din, B, dout);
        break;
      case 16:
        if (int r = allow_big_lds(qt::k_born_mfma<16>, at_bytes)) return r;
        hipLaunchKernelGGL(qt::k_born_mfma<16>, dim3(grid), dim3(1024), at_bytes, h->stream, h->AT.as<double>(), h->M, Mp, h->d,
                           din, B, dout);
        break;
      default:
        if (int r = allow_big_lds(qt::k_born_mfma<64>, at_bytes)) return r;
        hipLaunchKernelGGL(qt::k_born_mfma<64>, dim3(grid), dim3(1024), at_bytes, h->stream, h->AT.as<double>(), h->M, Mp, h->d,
                           din, B, dout);
        break;
    }
  } else if (h->D <= 256) {
    constexpr int TB = 8;
    int gy = (B + TB - 1) / TB;
    if (gy > 2048) gy = 2048;
    hipLaunchKernelGGL(qt::k_born<TB>, dim3(gx, gy), dim3(256), TB * h->D * sizeof(double), h->stream, h->AT.as<double>(),
                       h->M, h->D, h->d, din, B, dout);
  } else {
    constexpr int TB = 4;
    int gy = (B + TB - 1) / TB;
    if (gy > 2048) gy = 2048;
    hipLaunchKernelGGL(qt::k_born<TB>, dim3(gx, gy), dim3(256), TB * h->D * sizeof(double), h->stream, h->AT.as<double>(),
                       h->M, h->D, h->d, din, B, dout);
  }
  if (int r = fetch_out(h, dout, p, (size_t)B * h->M, flags)) return r;
  return finish(h, flags);
}

int qt_bloch_from_mat(qt_handle_t* h, const double* mat, int B, double* bloch, int flags) {
  QT_ENTER(h);
  if (B < 0 || (B > 0 && (!mat || !bloch))) return fail(QT_ERR_ARG, "bad bloch_from_mat arguments");
  if (B == 0) return 0;
  const double* din;
  double* dout;
  const size_t n = (size_t)B * h->D;
  if (int r = stage_in(h, h->in0, mat, n * 2, flags, &din)) return r;
  if (int r = stage_out(h, h->out0, bloch, n, flags, &dout)) return r;
  hipLaunchKernelGGL(qt::k_bloch_from_mat, dim3(grid_for(n)), dim3(256), 0, h->stream, h->nq, din, B, dout);
  if (int r = fetch_out(h, dout, bloch, n, flags)) return r;
  return finish(h, flags);
}

int qt_mat_from_bloch(qt_handle_t* h, const double* bloch, int B, double* mat, int flags) {
  QT_ENTER(h);
  if (B < 0 || (B > 0 && (!mat || !bloch))) return fail(QT_ERR_ARG, "bad mat_from_bloch arguments");
  if (B == 0) return 0;
  const double* din;
  double* dout;
  const size_t n = (size_t)B * h->D;
  if (int r = stage_in(h, h->in0, bloch, n, flags, &din)) return r;
  if (int r = stage_out(h, h->out0, mat, n * 2, flags, &dout)) return r;
  hipLaunchKernelGGL(qt::k_mat_from_bloch, dim3(grid_for(n)), dim3(256), 0, h->stream, h->nq, din, B, dout);
  if (int r = fetch_out(h, dout, mat, n * 2, flags)) return r;
  return finish(h, flags);
}

// a6 + a7 (+ a16 when `dist` is asked for): one body behind qt_lin_batch and qt_lin_dist_batch
static int lin_batch_impl(qt_handle_t* h, const int64_t* counts, int B, int physical, const double* centre, double* rho,
                          double* dist, double* bloch_out, int32_t* status, int flags) {
  QT_ENTER(h);
  if (int r = need_povm(h)) return r;
  if (B < 0 || (B > 0 && (!counts || (!rho && !dist) || (dist && !centre)))) return fail(QT_ERR_ARG, "bad lin_batch arguments");
  if (B == 0) return 0;
  const int64_t* dc;
  const double* dcen = nullptr;
  double *drho, *dbl, *ddist;
  int32_t* dst;
  const size_t nel = (size_t)B * h->D;
  if (int r = stage_in(h, h->in0, counts, (size_t)B * h->M, flags, &dc)) return r;
  if (dist)
    if (int r = stage_in(h, h->in1, centre, (size_t)h->D * 2, flags, &dcen)) return r;
  if (int r = stage_out(h, h->out0, rho, nel * 2, flags, &drho)) return r;
  if (int r = stage_out(h, h->out1, bloch_out, nel, flags, &dbl)) return r;
  if (int r = stage_out(h, h->out2, status, (size_t)B, flags, &dst)) return r;
  if (int r = stage_out(h, h->out3, dist, (size_t)B, flags, &ddist)) return r;
  const qt::EstOut eo{drho, dcen, ddist};
  if (h->nq >= 4) {
    if (int r = prepare_large(h, true)) return r;
    QT_LAUNCH_LARGE(qt::k_lin_large, B, h->M, h->prod.R1, (h->view(), dc, B, physical, eo, dbl, dst));
  } else {
    QT_LAUNCH_SMALL(qt::k_lin_batch, h->M, B, (h->view(), dc, B, physical, eo, dbl, dst));
  }
  if (int r = fetch_out(h, drho, rho, nel * 2, flags)) return r;
  if (int r = fetch_out(h, dbl, bloch_out, nel, flags)) return r;
  if (int r = fetch_out(h, dst, status, (size_t)B, flags)) return r;
  if (int r = fetch_out(h, ddist, dist, (size_t)B, flags)) return r;
  if (int r = finish(h, flags)) return r;
  return count_bad(status, B, flags);
}

int qt_lin_batch(qt_handle_t* h, const int64_t* counts, int B, int physical, double* rho, double* bloch_out,
                 int32_t* status, int flags) {
  if (B > 0 && !rho) return fail(QT_ERR_ARG, "bad lin_batch arguments");
  return lin_batch_impl(h, counts, B, physical, nullptr, rho, nullptr, bloch_out, status, flags);
}

int qt_lin_dist_batch(qt_handle_t* h, const int64_t* counts, int B, int physical, const double* centre, double* rho,
                      double* dist, int32_t* status, int flags) {
  if (B > 0 && (!dist || !centre)) return fail(QT_ERR_ARG, "bad lin_dist_batch arguments");
  return lin_batch_impl(h, counts, B, physical, centre, rho, dist, nullptr, status, flags);
}

int qt_chol_param(qt_handle_t* h, const double* rho, int B, double* x, int32_t* status, int flags) {
  QT_ENTER(h);
  if (B < 0 || (B > 0 && (!rho || !x))) return fail(QT_ERR_ARG, "bad chol_param arguments");
  if (B == 0) return 0;
  const double* din;
  double* dx;
  int32_t* dst;
  const size_t nel = (size_t)B * h->D;
  if (int r = stage_in(h, h->in0, rho, nel * 2, flags, &din)) return r;
  if (int r = stage_out(h, h->out0, x, nel, flags, &dx)) return r;
  if (int r = stage_out(h, h->out2, status, (size_t)B, flags, &dst)) return r;
  qt::PovmView pv{};
  if (h->nq >= 4) {
    QT_LAUNCH_LARGE(qt::k_chol_param_large, B, 0, 1, (pv, din, B, dx, dst));
  } else {
    QT_LAUNCH_SMALL_NOIMG(qt::k_chol_param, B, (pv, din, B, dx, dst));
  }
  if (int r = fetch_out(h, dx, x, nel, flags)) return r;
  if (int r = fetch_out(h, dst, status, (size_t)B, flags)) return r;
  if (int r = finish(h, flags)) return r;
  return count_bad(status, B, flags);
}

int qt_chol_unparam(qt_handle_t* h, const double* x, int B, double* LLh, int flags) {
  QT_ENTER(h);
  if (B < 0 || (B > 0 && (!x || !LLh))) return fail(QT_ERR_ARG, "bad chol_unparam arguments");
  if (B == 0) return 0;
  const double* din;
  double* dout;
  const size_t nel = (size_t)B * h->D;
  if (int r = stage_in(h, h->in0, x, nel, flags, &din)) return r;
  if (int r = stage_out(h, h->out0, LLh, nel * 2, flags, &dout)) return r;
  qt::PovmView pv{};
  if (h->nq >= 4) {
    QT_LAUNCH_LARGE(qt::k_chol_unparam_large, B, 0, 1, (pv, din, B, dout));
  } else {
    QT_LAUNCH_SMALL_NOIMG(qt::k_chol_unparam, B, (pv, din, B, dout));
  }
  if (int r = fetch_out(h, dout, LLh, nel * 2, flags)) return r;
  return finish(h, flags);
}

int qt_nll_batch(qt_handle_t* h, const double* x, const int64_t* counts, int B, double* f, double* grad, int flags) {
  QT_ENTER(h);
  if (int r = need_povm(h)) return r;
  if (B < 0 || (B > 0 && (!x || !counts || !f))) return fail(QT_ERR_ARG, "bad nll_batch arguments");
  if (B == 0) return 0;
  const double* dx;
  const int64_t* dc;
  double *df, *dg;
  const size_t nel = (size_t)B * h->D;
  if (int r = stage_in(h, h->in0, x, nel, flags, &dx)) return r;
  if (int r = stage_in(h, h->in1, counts, (size_t)B * h->M, flags, &dc)) return r;
  if (int r = stage_out(h, h->out0, f, (size_t)B, flags, &df)) return r;
  if (int r = stage_out(h, h->out1, grad, nel, flags, &dg)) return r;
  if (h->nq >= 4) {
    QT_LAUNCH_LARGE(qt::k_nll_large, B, h->M, h->prod.R1, (h->view(), dx, dc, B, df, dg));
  } else {
    QT_LAUNCH_SMALL(qt::k_nll_batch, h->M, B, (h->view(), dx, dc, B, df, dg));
  }
  if (int r = fetch_out(h, df, f, (size_t)B, flags)) return r;
  if (int r = fetch_out(h, dg, grad, nel, flags)) return r;
  return finish(h, flags);
}

int qt_mhmc_state(qt_handle_t* h, const int64_t* counts, int C, const double* x_init, const double* deltas,
                  const double* uniforms, int T, double step, double* chain, int32_t* accepted, int flags) {
  QT_ENTER(h);
  if (int r = need_povm(h)) return r;
  if (C < 0 || T < 0 || (C > 0 && T > 0 && (!counts || !x_init || !deltas || !uniforms || !chain || !accepted)))
    return fail(QT_ERR_ARG, "bad mhmc_state arguments");
  if (h->nq > 3) return fail(QT_ERR_UNSUPPORTED, "qt_mhmc_state supports n_qubits 1..3");
  if (C == 0 || T == 0) return 0;
  const int64_t* dc;
  const double *dx, *dd, *du;
  double* dch;
  int32_t* dacc;
  const size_t nel = (size_t)C * T * h->D;
  if (int r = stage_in(h, h->in0, counts, (size_t)C * h->M, flags, &dc)) return r;
  if (int r = stage_in(h, h->in1, x_init, (size_t)C * h->D, flags, &dx)) return r;
  if (int r = stage_in(h, h->out2, deltas, nel, flags, &dd)) return r;
  if (int r = stage_in(h, h->out3, uniforms, (size_t)C * T, flags, &du)) return r;
  if (int r = stage_out(h, h->out0, chain, nel, flags, &dch)) return r;
  if (int r = stage_out(h, h->out1, accepted, (size_t)C * T, flags, &dacc)) return r;
  QT_LAUNCH_SMALL(qt::k_mhmc_state, h->M, C, (h->view(), dc, C, dx, dd, du, T, step, dch, dacc));
  if (int r = fetch_out(h, dch, chain, nel, flags)) return r;
  if (int r = fetch_out(h, dacc, accepted, (size_t)C * T, flags)) return r;
  return finish(h, flags);
}

// a8-a10 (+ a16 when `dist` is asked for): one body behind qt_mle_batch and qt_mle_dist_batch
static int mle_batch_impl(qt_handle_t* h, const int64_t* counts, int B, int init, int max_iter, double tol,
                          const double* centre, double* rho, double* dist, int32_t* nit, int32_t* nfev, double* fun,
                          int32_t* status, int flags) {
  QT_ENTER(h);
  if (int r = need_povm(h)) return r;
  if (B < 0 || (B > 0 && (!counts || (!rho && !dist) || (dist && !centre)))) return fail(QT_ERR_ARG, "bad mle_batch arguments");
  if (init != QT_INIT_LIN && init != QT_INIT_MIXED) return fail(QT_ERR_ARG, "init must be QT_INIT_LIN or QT_INIT_MIXED");
  if (max_iter < 0) return fail(QT_ERR_ARG, "max_iter < 0");
  if (B == 0) return 0;
  const int64_t* dc;
  const double* dcen = nullptr;
  double *drho, *dfun, *ddist;
  int32_t *dnit, *dnfev, *dst;
  const size_t nel = (size_t)B * h->D;
  if (int r = stage_in(h, h->in0, counts, (size_t)B * h->M, flags, &dc)) return r;
  if (dist)
    if (int r = stage_in(h, h->in1, centre, (size_t)h->D * 2, flags, &dcen)) return r;
  if (int r = stage_out(h, h->out0, rho, nel * 2, flags, &drho)) return r;
  if (int r = stage_out(h, h->out5, dist, (size_t)B, flags, &ddist)) return r;
  const qt::EstOut eo{drho, dcen, ddist};
  if (int r = stage_out(h, h->out1, nit, (size_t)B, flags, &dnit)) return r;
  if (int r = stage_out(h, h->out2, nfev, (size_t)B, flags, &dnfev)) return r;
  if (int r = stage_out(h, h->out3, fun, (size_t)B, flags, &dfun)) return r;
  if (int r = stage_out(h, h->out4, status, (size_t)B, flags, &dst)) return r;
  if (h->nq >= 4) {
    if (int r = prepare_large(h, init == QT_INIT_LIN)) return r;
    // BFGS history: 2 D doubles per iteration and trial, processed in chunks of <= 4 GiB
    if (max_iter > 4096) return fail(QT_ERR_UNSUPPORTED, "max_iter > 4096 is not supported for n_qubits >= 4");
    const size_t per_trial = (size_t)(max_iter > 0 ? max_iter : 1) * 2 * h->D * sizeof(double);
    int chunk = (int)(((size_t)4 << 30) / per_trial);
    if (chunk < 1) chunk = 1;
    if (chunk > B) chunk = B;
    HIPCHK(h->hess.ensure((size_t)chunk * per_trial));
    HIPCHK(h->ws_x.ensure(nel * sizeof(double)));
    HIPCHK(h->ws_g.ensure(nel * sizeof(double)));
    HIPCHK(h->ws_f.ensure((size_t)B * sizeof(double)));
    HIPCHK(h->ws_act.ensure((size_t)B * sizeof(int32_t)));
    double *wx = h->ws_x.as<double>(), *wg = h->ws_g.as<double>(), *wf = h->ws_f.as<double>();
    int32_t* wact = h->ws_act.as<int32_t>();
    // start point + first evaluation of every trial; then the BFGS loop of those that iterate, chunk by chunk
    QT_LAUNCH_LARGE(qt::k_mle_large_start, B, h->M, h->prod.R1,
                    (h->view(), dc, B, init, max_iter, tol, eo, dnit, dnfev, dfun, dst, wx, wg, wf, wact));
    for (int b0 = 0; b0 < B; b0 += chunk) {
      const int nb = (B - b0 < chunk) ? B - b0 : chunk;
      QT_LAUNCH_LARGE_X(qt::k_mle_large_bfgs, nb, h->M, h->prod.R1, max_iter,
                        (h->view(), dc + (size_t)b0 * h->M, nb, max_iter, tol,
                         qt::EstOut{drho ? drho + (size_t)b0 * h->D * 2 : nullptr, dcen, ddist ? ddist + b0 : nullptr},
                         dnit ? dnit + b0 : nullptr, dnfev ? dnfev + b0 : nullptr, dfun ? dfun + b0 : nullptr,
                         dst ? dst + b0 : nullptr, wx + (size_t)b0 * h->D, wg + (size_t)b0 * h->D, wf + b0, wact + b0,
                         h->hess.as<double>()));
    }
  } else {
    // up to one resident wave per SIMD (1024 trial-waves) the single fused launch wins; beyond that the
    // 256-VGPR BFGS loop would cap occupancy for every trial, so the split pair is used
    const int waves = (B + (64 / h->D > 0 ? 64 / h->D : 1) - 1) / (64 / h->D > 0 ? 64 / h->D : 1);
    // n = 3 keeps rho_i / alpha_i of every BFGS iteration in the trial's LDS (16 bytes per iteration and trial):
    // the one-launch kernel (which also keeps 24 pairs there) up to 256 iterations, the split pair up to 2000
    if (h->nq == 3 && max_iter > 2000)
      return fail(QT_ERR_UNSUPPORTED, "max_iter > 2000 is not supported for n_qubits = 3 (LDS holds the two-loop scalars)");
    if (waves <= h->fused_max_waves && !(h->nq == 3 && max_iter > 256)) {
      if (h->nq == 3) {  // two-loop BFGS: line-search state, rho_i, alpha_i and the first pairs in LDS, later pairs in global
        const int mi = max_iter > 0 ? max_iter : 1;
        h->lds_extra = qt::LineSearch::SLOTS + 2 * mi + qt::kFusedLdsPairs * 2 * h->D;
        const int over = mi > qt::kFusedLdsPairs ? mi : 1;  // (indexed by pair number: rows below kFusedLdsPairs stay unused)
        HIPCHK(h->hess.ensure((size_t)B * over * 2 * h->D * sizeof(double)));
      }
      if (init == QT_INIT_LIN) {
        QT_LAUNCH_SMALL(qt::k_mle_fused, h->M, B,
                        (h->view(), dc, B, max_iter, tol, eo, dnit, dnfev, dfun, dst, h->hess.as<double>()));
      } else {
        QT_LAUNCH_SMALL(qt::k_mle_fused_mixed, h->M, B,
                        (h->view(), dc, B, max_iter, tol, eo, dnit, dnfev, dfun, dst, h->hess.as<double>()));
      }
      h->lds_extra = 0;
    } else {
      HIPCHK(h->ws_x.ensure(nel * sizeof(double)));
      HIPCHK(h->ws_g.ensure(nel * sizeof(double)));
      HIPCHK(h->ws_f.ensure((size_t)B * sizeof(double)));
      HIPCHK(h->ws_act.ensure((size_t)B * sizeof(int32_t)));
      double *wx = h->ws_x.as<double>(), *wg = h->ws_g.as<double>(), *wf = h->ws_f.as<double>();
      int32_t* wact = h->ws_act.as<int32_t>();
      QT_LAUNCH_SMALL(qt::k_mle_start, h->M, B,
                      (h->view(), dc, B, init, max_iter, tol, eo, dnit, dnfev, dfun, dst, wx, wg, wf, wact));
      // BFGS history of the trials that iterate: 2 D doubles per iteration and trial (two-loop recursion), in
      // chunks of <= 4 GiB; rho_i, alpha_i and the parked line-search state in LDS
      int chunk = B;
      if (h->nq == 3) {  // (n = 1, 2 keep the 4 / 16-entry Hessian rows in registers: no workspace)
        const size_t per_trial = (size_t)(max_iter > 0 ? max_iter : 1) * 2 * h->D * sizeof(double);
        chunk = (int)(((size_t)4 << 30) / per_trial);
        if (chunk < 1) chunk = 1;
        if (chunk > B) chunk = B;
        HIPCHK(h->hess.ensure((size_t)chunk * per_trial));
        h->lds_extra = qt::LineSearch::SLOTS + 2 * (max_iter > 0 ? max_iter : 1);
      }
      for (int b0 = 0; b0 < B; b0 += chunk) {
        const int nb = (B - b0 < chunk) ? B - b0 : chunk;
        QT_LAUNCH_SMALL(qt::k_mle_bfgs, h->M, nb,
                        (h->view(), dc + (size_t)b0 * h->M, nb, max_iter, tol,
                         qt::EstOut{drho ? drho + (size_t)b0 * h->D * 2 : nullptr, dcen, ddist ? ddist + b0 : nullptr},
                         dnit ? dnit + b0 : nullptr, dnfev ? dnfev + b0 : nullptr, dfun ? dfun + b0 : nullptr,
                         dst ? dst + b0 : nullptr, wx + (size_t)b0 * h->D, wg + (size_t)b0 * h->D, wf + b0, wact + b0,
                         h->hess.as<double>()));
      }
      h->lds_extra = 0;
    }
  }
  if (int r = fetch_out(h, drho, rho, nel * 2, flags)) return r;
  if (int r = fetch_out(h, dnit, nit, (size_t)B, flags)) return r;
  if (int r = fetch_out(h, dnfev, nfev, (size_t)B, flags)) return r;
  if (int r = fetch_out(h, dfun, fun, (size_t)B, flags)) return r;
  if (int r = fetch_out(h, dst, status, (size_t)B, flags)) return r;
  if (int r = fetch_out(h, ddist, dist, (size_t)B, flags)) return r;
  if (int r = finish(h, flags)) return r;
  return count_bad(status, B, flags);
}

int qt_mle_batch(qt_handle_t* h, const int64_t* counts, int B, int init, int max_iter, double tol, double* rho,
                 int32_t* nit, int32_t* nfev, double* fun, int32_t* status, int flags) {
  if (B > 0 && !rho) return fail(QT_ERR_ARG, "bad mle_batch arguments");
  return mle_batch_impl(h, counts, B, init, max_iter, tol, nullptr, rho, nullptr, nit, nfev, fun, status, flags);
}

int qt_mle_dist_batch(qt_handle_t* h, const int64_t* counts, int B, int init, int max_iter, double tol,
                      const double* centre, double* rho, double* dist, int32_t* nit, int32_t* nfev, double* fun,
                      int32_t* status, int flags) {
  if (B > 0 && (!dist || !centre)) return fail(QT_ERR_ARG, "bad mle_dist_batch arguments");
  return mle_batch_impl(h, counts, B, init, max_iter, tol, centre, rho, dist, nit, nfev, fun, status, flags);
}

int qt_hs_dist_batch(qt_handle_t* h, const double* rho, const double* centre, int B, double* dist, int flags) {
  QT_ENTER(h);
  if (B < 0 || (B > 0 && (!rho || !centre || !dist))) return fail(QT_ERR_ARG, "bad hs_dist arguments");
  if (B == 0) return 0;
  const double *dr, *dcn;
  double* dd;
  if (int r = stage_in(h, h->in0, rho, (size_t)B * h->D * 2, flags, &dr)) return r;
  if (int r = stage_in(h, h->in1, centre, (size_t)h->D * 2, flags, &dcn)) return r;
  if (int r = stage_out(h, h->out0, dist, (size_t)B, flags, &dd)) return r;
  hipLaunchKernelGGL(qt::k_hs_dist, dim3(B), dim3(64), 0, h->stream, h->d, dr, dcn, B, dd);
  if (int r = fetch_out(h, dd, dist, (size_t)B, flags)) return r;
  return finish(h, flags);
}

// the same for dim x dim matrices of any size (the Choi matrices of an n-qubit channel are 4^n x 4^n: 2n-qubit objects)
int qt_hs_dist_dim(qt_handle_t* h, int dim, const double* rho, const double* centre, int B, double* dist, int flags) {
  QT_ENTER(h);
  if (dim < 1 || dim > 4096 || B < 0 || (B > 0 && (!rho || !centre || !dist))) return fail(QT_ERR_ARG, "bad hs_dist arguments");
  if (B == 0) return 0;
  const size_t ne = (size_t)dim * dim;
  const double *dr, *dcn;
  double* dd;
  if (int r = stage_in(h, h->in0, rho, (size_t)B * ne * 2, flags, &dr)) return r;
  if (int r = stage_in(h, h->in1, centre, ne * 2, flags, &dcn)) return r;
  if (int r = stage_out(h, h->out0, dist, (size_t)B, flags, &dd)) return r;
  hipLaunchKernelGGL(qt::k_hs_dist, dim3(B), dim3(64), 0, h->stream, dim, dr, dcn, B, dd);
  if (int r = fetch_out(h, dd, dist, (size_t)B, flags)) return r;
  return finish(h, flags);
}

// ---- a16: interval.py:610-612 ------------------------------------------------------------------------
int qt_sort_f64(qt_handle_t* h, double* x, long long n, int flags) {
  QT_ENTER(h);
  if (n < 0 || (n > 0 && !x)) return fail(QT_ERR_ARG, "bad sort arguments");
  if (n > 0x7fffffffLL) return fail(QT_ERR_UNSUPPORTED, "qt_sort_f64 sorts at most 2^31 - 1 values");
  if (n == 0) return 0;
  const double* din;
  if (int r = stage_in(h, h->in0, (const double*)x, (size_t)n, flags, &din)) return r;
  double* dx = const_cast<double*>(din);
  if (n <= 8192) {  // one workgroup, bitonic network in LDS
    int np2 = 2;
    while (np2 < n) np2 <<= 1;
    hipLaunchKernelGGL(qt::k_sort_small, dim3(1), dim3(np2 / 2 < 1024 ? (np2 / 2 < 64 ? 64 : np2 / 2) : 1024), np2 * sizeof(double),
                       h->stream, dx, (int)n, np2);
    if (int r = fetch_out(h, (const double*)dx, x, (size_t)n, flags)) return r;
    return finish(h, flags);
  }
  HIPCHK(h->sort_alt.ensure((size_t)n * sizeof(double)));
  hipcub::DoubleBuffer<double> keys(dx, h->sort_alt.as<double>());
  size_t tmp_bytes = 0;
  HIPCHK(hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, keys, (int)n, 0, 64, h->stream));
  HIPCHK(h->sort_tmp.ensure(tmp_bytes));
  HIPCHK(hipcub::DeviceRadixSort::SortKeys(h->sort_tmp.p, tmp_bytes, keys, (int)n, 0, 64, h->stream));
  if (keys.Current() != dx)
    HIPCHK(hipMemcpyAsync(dx, keys.Current(), (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  if (int r = fetch_out(h, (const double*)dx, x, (size_t)n, flags)) return r;
  return finish(h, flags);
}

int qt_sorted_quantiles(qt_handle_t* h, const double* sorted, long long n, const double* conf_levels, int n_levels,
                        double* out, int flags) {
  QT_ENTER(h);
  if (n < 1 || n_levels < 0 || !sorted || (n_levels > 0 && (!conf_levels || !out)))
    return fail(QT_ERR_ARG, "bad sorted_quantiles arguments");
  if (n_levels == 0) return 0;
  const double *ds, *dq;
  double* dout;
  if (int r = stage_in(h, h->in0, sorted, (size_t)n, flags, &ds)) return r;
  if (int r = stage_in(h, h->in1, conf_levels, (size_t)n_levels, flags, &dq)) return r;
  if (int r = stage_out(h, h->out0, out, (size_t)n_levels, flags, &dout)) return r;
  hipLaunchKernelGGL(qt::k_interp_sorted, dim3((n_levels + 255) / 256), dim3(256), 0, h->stream, ds, n, dq, n_levels, dout);
  if (int r = fetch_out(h, dout, out, (size_t)n_levels, flags)) return r;
  return finish(h, flags);
}

// ---- a16 over several ranks: the order statistics interp1d needs from a sample whose sorted shards live on N ranks
// (kernels and the argument in qt_ops.h; the two all-gathers in between are quantpy_amd/distributed.py's) -----------------
int qt_select_splitters(qt_handle_t* h, const double* sorted, long long n, long long stride, int P, double* splitters,
                        int flags) {
  QT_ENTER(h);
  if (n < 0 || stride < 1 || P < 1 || (n > 0 && !sorted) || !splitters) return fail(QT_ERR_ARG, "bad select_splitters arguments");
  const double* ds;
  double* dout;
  if (int r = stage_in(h, h->in0, sorted, (size_t)(n > 0 ? n : 1), n > 0 ? flags : QT_DEVICE_PTR, &ds)) return r;
  if (int r = stage_out(h, h->out0, splitters, (size_t)P, flags, &dout)) return r;
  hipLaunchKernelGGL(qt::k_select_splitters, dim3((P + 255) / 256), dim3(256), 0, h->stream, ds, n, stride, P, dout);
  if (int r = fetch_out(h, dout, splitters, (size_t)P, flags)) return r;
  return finish(h, flags);
}

int qt_select_bracket(qt_handle_t* h, const double* splitters, int N, int P, const int64_t* sizes, long long stride,
                      long long n_total, const double* conf_levels, int L, uint64_t* lo_key, uint64_t* hi_key, int flags) {
  QT_ENTER(h);
  if (N < 1 || P < 1 || L < 1 || stride < 1 || n_total < 0 || !splitters || !sizes || !conf_levels || !lo_key || !hi_key)
    return fail(QT_ERR_ARG, "bad select_bracket arguments");
  const double *dspl, *dq;
  const int64_t* dsz;
  uint64_t *dlo, *dhi;
  if (int r = stage_in(h, h->in0, splitters, (size_t)N * P, flags, &dspl)) return r;
  if (int r = stage_in(h, h->in1, sizes, (size_t)N, flags, &dsz)) return r;
  if (int r = stage_in(h, h->out2, conf_levels, (size_t)L, flags, &dq)) return r;
  if (int r = stage_out(h, h->out0, lo_key, (size_t)L, flags, &dlo)) return r;
  if (int r = stage_out(h, h->out1, hi_key, (size_t)L, flags, &dhi)) return r;
  hipLaunchKernelGGL(qt::k_select_init, dim3((L + 255) / 256), dim3(256), 0, h->stream,
                     reinterpret_cast<unsigned long long*>(dlo), reinterpret_cast<unsigned long long*>(dhi), L);
  hipLaunchKernelGGL(qt::k_select_bracket, dim3((unsigned)(((size_t)N * P + 255) / 256)), dim3(256), 0, h->stream, dspl, N, P,
                     reinterpret_cast<const long long*>(dsz), stride, n_total, dq, L,
                     reinterpret_cast<unsigned long long*>(dlo), reinterpret_cast<unsigned long long*>(dhi));
  if (int r = fetch_out(h, (const uint64_t*)dlo, lo_key, (size_t)L, flags)) return r;
  if (int r = fetch_out(h, (const uint64_t*)dhi, hi_key, (size_t)L, flags)) return r;
  return finish(h, flags);
}

int qt_select_window(qt_handle_t* h, const double* sorted, long long n, const uint64_t* lo_key, const uint64_t* hi_key, int L,
                     int W, double* window, int flags) {
  QT_ENTER(h);
  if (n < 0 || L < 1 || W < 1 || (n > 0 && !sorted) || !lo_key || !hi_key || !window)
    return fail(QT_ERR_ARG, "bad select_window arguments");
  const double* ds;
  const uint64_t *dlo, *dhi;
  double* dwin;
  const size_t wn = (size_t)L * (2 + W);
  if (int r = stage_in(h, h->in0, sorted, (size_t)(n > 0 ? n : 1), n > 0 ? flags : QT_DEVICE_PTR, &ds)) return r;
  if (int r = stage_in(h, h->in1, lo_key, (size_t)L, flags, &dlo)) return r;
  if (int r = stage_in(h, h->out2, hi_key, (size_t)L, flags, &dhi)) return r;
  if (int r = stage_out(h, h->out0, window, wn, flags, &dwin)) return r;
  hipLaunchKernelGGL(qt::k_select_window, dim3(L), dim3(256), 0, h->stream, ds, n,
                     reinterpret_cast<const unsigned long long*>(dlo), reinterpret_cast<const unsigned long long*>(dhi), W, dwin);
  if (int r = fetch_out(h, dwin, window, wn, flags)) return r;
  return finish(h, flags);
}

int qt_select_finish(qt_handle_t* h, const double* windows, int N, int L, int W, long long n_total, const double* conf_levels,
                     double* out, int32_t* overflow, int flags) {
  QT_ENTER(h);
  if (N < 1 || L < 1 || W < 1 || n_total < 1 || !windows || !conf_levels || !out || !overflow)
    return fail(QT_ERR_ARG, "bad select_finish arguments");
  const double *dw, *dq;
  double* dout;
  int32_t* dfl;
  const size_t wn = (size_t)N * L * (2 + W);
  if (int r = stage_in(h, h->in0, windows, wn, flags, &dw)) return r;
  if (int r = stage_in(h, h->in1, conf_levels, (size_t)L, flags, &dq)) return r;
  if (int r = stage_out(h, h->out0, out, (size_t)L, flags, &dout)) return r;
  if (int r = stage_out(h, h->out1, overflow, 1, flags, &dfl)) return r;
  HIPCHK(hipMemsetAsync(dfl, 0, sizeof(int32_t), h->stream));
  size_t cap = (size_t)N * W;
  if (cap > 16000) cap = 16000;  // 128 KB of LDS; a larger union raises the overflow flag (heavy ties: take the merge path)
  const size_t lds = cap * sizeof(double) + ((size_t)N + 2) * sizeof(int);
  if (int r = allow_big_lds(qt::k_select_finish, lds)) return r;
  hipLaunchKernelGGL(qt::k_select_finish, dim3(L), dim3(1024), lds, h->stream, dw, N, L, W, (int)cap, n_total, dq, dout,
                     reinterpret_cast<int*>(dfl));
  if (int r = fetch_out(h, dout, out, (size_t)L, flags)) return r;
  if (int r = fetch_out(h, dfl, overflow, 1, flags)) return r;
  return finish(h, flags);
}

// R sorted runs, concatenated in `runs` (lengths: a HOST array, the launch geometry depends on them) -> out sorted.
// Pairwise merge-path passes, ceil(log2 R) of them, ping-ponging between out and a scratch buffer.
int qt_merge_sorted(qt_handle_t* h, const double* runs, const int64_t* run_lengths, int R, double* out, int flags) {
  QT_ENTER(h);
  if (R < 1 || !run_lengths || !out) return fail(QT_ERR_ARG, "bad merge_sorted arguments");
  long long n = 0;
  for (int r = 0; r < R; ++r) {
    if (run_lengths[r] < 0) return fail(QT_ERR_ARG, "negative run length");
    n += run_lengths[r];
  }
  if (n == 0) return 0;
  if (!runs) return fail(QT_ERR_ARG, "bad merge_sorted arguments");
  const double* din;
  double* dout;
  if (int r = stage_in(h, h->in0, runs, (size_t)n, flags, &din)) return r;
  if (int r = stage_out(h, h->out0, out, (size_t)n, flags, &dout)) return r;
  std::vector<long long> len(run_lengths, run_lengths + R);
  int passes = 0;
  for (int m = R; m > 1; m = (m + 1) / 2) ++passes;
  if (passes == 0) {
    if (din != dout) HIPCHK(hipMemcpyAsync(dout, din, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  } else {
    HIPCHK(h->sort_alt.ensure((size_t)n * sizeof(double)));
    double* alt = h->sort_alt.as<double>();
    // the last pass must land in dout: choose the first destination accordingly (the source of pass 0 is din, read-only)
    const double* src = din;
    double* dst = (passes & 1) ? dout : alt;
    if (dst == src) return fail(QT_ERR_ARG, "qt_merge_sorted: out must not alias runs");
    constexpr int TILE = 8;
    for (int p = 0; p < passes; ++p) {
      std::vector<long long> next;
      long long at = 0;
      for (size_t r = 0; r < len.size(); r += 2) {
        const long long na = len[r], nb = r + 1 < len.size() ? len[r + 1] : 0;
        if (na + nb > 0) {
          const long long threads = (na + nb + TILE - 1) / TILE;
          hipLaunchKernelGGL(qt::k_merge_runs<TILE>, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, h->stream, src + at,
                             na, src + at + na, nb, dst + at);
        }
        next.push_back(na + nb);
        at += na + nb;
      }
      len.swap(next);
      src = dst;
      dst = (dst == dout) ? alt : dout;
    }
  }
  if (int r = fetch_out(h, (const double*)dout, out, (size_t)n, flags)) return r;
  return finish(h, flags);
}

// ---- f2: stats.py:21-47 over a batch (MomentInterval, interval.py:59-110) ------------------------------------------------
int qt_moment_batch(qt_handle_t* h, const int64_t* counts, int B, int S, int K, const double* ns, const double* inv_matrix,
                    int rows, double n_trials, double* mean, double* var, int flags) {
  QT_ENTER(h);
  if (B < 0 || S < 1 || K < 1 || rows < 1 || !(n_trials > 0.0) || !ns || !inv_matrix || (B > 0 && (!counts || !mean || !var)))
    return fail(QT_ERR_ARG, "bad moment_batch arguments");
  const long long M = (long long)S * K;
  if (M > 8192) return fail(QT_ERR_UNSUPPORTED, "qt_moment_batch supports up to 8192 POVM rows (got %lld)", M);
  if (B == 0) return 0;
  const int64_t* dc;
  const double *dns, *dp;
  double *dmean, *dvar;
  if (int r = stage_in(h, h->in0, counts, (size_t)B * M, flags, &dc)) return r;
  if (int r = stage_in(h, h->in1, ns, (size_t)S, flags, &dns)) return r;
  if (int r = stage_in(h, h->out2, inv_matrix, (size_t)rows * M, flags, &dp)) return r;
  if (int r = stage_out(h, h->out0, mean, (size_t)B, flags, &dmean)) return r;
  if (int r = stage_out(h, h->out1, var, (size_t)B, flags, &dvar)) return r;
  // W = P^T P on the matrix cores ([M x rows] . [rows x M])
  HIPCHK(h->out3.ensure((size_t)M * M * sizeof(double)));
  double* dW = h->out3.as<double>();
  hipLaunchKernelGGL(qt::k_gemm<0>, dim3((unsigned)((M + 15) / 16), (unsigned)((M + 15) / 16)), dim3(64), 0, h->stream, (int)M, (int)M,
                     rows, dp, (int)M, 1, dp, (int)M, 0, dW, (int)M);
  if (M <= 1024) {
    constexpr int T = 4;
    const size_t lds = (size_t)2 * T * M * sizeof(double);
    if (int r = allow_big_lds(qt::k_moment_batch<T, 4>, lds)) return r;
    hipLaunchKernelGGL((qt::k_moment_batch<T, 4>), dim3((B + T - 1) / T), dim3(256), lds, h->stream, dc, B, S, K, dns, dW, n_trials,
                       dmean, dvar);
  } else {
    const size_t lds = (size_t)2 * M * sizeof(double);
    if (int r = allow_big_lds(qt::k_moment_batch<1, 32>, lds)) return r;
    hipLaunchKernelGGL((qt::k_moment_batch<1, 32>), dim3(B), dim3(256), lds, h->stream, dc, B, S, K, dns, dW, n_trials, dmean, dvar);
  }
  if (int r = fetch_out(h, dmean, mean, (size_t)B, flags)) return r;
  if (int r = fetch_out(h, dvar, var, (size_t)B, flags)) return r;
  return finish(h, flags);
}

// ---- a4 / a12 / a16 host side: state.py:109-114, the draws of experiment() (qt_sampler.h) ---------
static int check_pvals(int period, int K, const int64_t* n, const double* pvals);

int qt_legacy_multinomial(uint32_t* mt_key, int* mt_pos, long long rows, int period, const int64_t* n,
                          const double* pvals, int K, int64_t* out) {
  if (!mt_key || !mt_pos || !n || !pvals || (rows > 0 && !out) || rows < 0 || period < 1 || K < 1)
    return fail(QT_ERR_ARG, "bad legacy_multinomial arguments");
  if (*mt_pos < 0 || *mt_pos > 624) return fail(QT_ERR_ARG, "MT19937 position %d outside 0..624", *mt_pos);
  if (int r = check_pvals(period, K, n, pvals)) return r;
  qt_sampler::Mt19937 g{mt_key, *mt_pos};
  std::vector<qt_sampler::BinomialSetup> cache((size_t)period * K);
  for (long long r = 0; r < rows; ++r) {
    const int s = (int)(r % period);
    qt_sampler::legacy_multinomial(g, n[s], pvals + (size_t)s * K, K, out + (size_t)r * K, cache.data() + (size_t)s * K);
  }
  *mt_pos = g.pos;
  return 0;
}

static int check_pvals(int period, int K, const int64_t* n, const double* pvals) {
  for (int s = 0; s < period; ++s) {
    if (n[s] < 0) return fail(QT_ERR_ARG, "n < 0 in row %d", s);
    // RandomState.multinomial's own checks (mtrand.pyx): every pval in [0, 1], and the leading K - 1 may not exceed 1
    double head = 0.0, comp = 0.0;  // compensated sum, as NumPy's check has it
    for (int j = 0; j < K; ++j) {
      const double p = pvals[(size_t)s * K + j];
      if (!(p >= 0.0 && p <= 1.0)) return fail(QT_ERR_ARG, "pvals < 0, pvals > 1 or pvals contains NaNs");
      if (j == 0 && K > 1) head = p;
      if (j > 0 && j < K - 1) {
        const double y = p - comp, t = head + y;
        comp = (t - head) - y;
        head = t;
      }
    }
    if (head > 1.0 + 1e-12) return fail(QT_ERR_ARG, "sum(pvals[:-1]) > 1.0");
  }
  return 0;
}

int qt_device_multinomial(qt_handle_t* h, uint64_t seed, uint64_t first_row, long long rows, int period,
                          const int64_t* n, const double* pvals, int K, int64_t* out, int flags) {
  QT_ENTER(h);
  if (!n || !pvals || (rows > 0 && !out) || rows < 0 || period < 1 || K < 1)
    return fail(QT_ERR_ARG, "bad device_multinomial arguments");
  if (rows > (1LL << 40)) return fail(QT_ERR_UNSUPPORTED, "qt_device_multinomial draws at most 2^40 rows per call");
  if (!(flags & QT_DEVICE_PTR))
    if (int r = check_pvals(period, K, n, pvals)) return r;
  if (rows == 0) return 0;
  const int64_t* dn;
  const double* dp;
  int64_t* dout;
  if (int r = stage_in(h, h->in0, n, (size_t)period, flags, &dn)) return r;
  if (int r = stage_in(h, h->in1, pvals, (size_t)period * K, flags, &dp)) return r;
  if (int r = stage_out(h, h->out0, out, (size_t)rows * K, flags, &dout)) return r;
  // whole 64 x period blocks of rows per launch (a wavefront = one setting of 64 consecutive resamples); a launch covers at
  // most 2^30 threads -- HIP rejects grids of 2^32 threads and more -- so larger tables go out in chunks, each keyed by its
  // own first row: the table depends on (seed, global row) only
  const long long span = 64LL * period;
  const long long chunk_rows = (span >= (1LL << 30)) ? span : ((1LL << 30) / span) * span;
  for (long long r0 = 0; r0 < rows; r0 += chunk_rows) {
    const long long nr = rows - r0 < chunk_rows ? rows - r0 : chunk_rows;
    const long long threads = (nr + span - 1) / span * span;
    hipLaunchKernelGGL(qt_sampler::k_multinomial_rows, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, h->stream, seed,
                       first_row + (uint64_t)r0, nr, period, dn, dp, K, dout + (size_t)r0 * K);
  }
  if (int r = fetch_out(h, (const int64_t*)dout, out, (size_t)rows * K, flags)) return r;
  return finish(h, flags);
}

void qt_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { qt_sampler::philox4x32_10(ctr, key, out); }

// ---- a5 for arbitrary matrices: routines.py:69-71 -------------------------------------------------
int qt_left_inverse(qt_handle_t* h, const double* A, int rows, int cols, int is_complex, double* out, int flags) {
  QT_ENTER(h);
  if (!A || !out || rows < 1 || cols < 1) return fail(QT_ERR_ARG, "bad left_inverse arguments");
  if (rows < cols) return fail(QT_ERR_SINGULAR, "matrix has fewer rows (%d) than columns (%d)", rows, cols);
  const int W = is_complex ? 2 : 1;
  const size_t nel = (size_t)rows * cols * W;
  const double* dA;
  double* dout;
  if (int r = stage_in(h, h->in0, A, nel, flags, &dA)) return r;
  if (int r = stage_out(h, h->out0, out, nel, flags, &dout)) return r;
  DevBuf& aug = h->proc_aug;
  HIPCHK(aug.ensure((size_t)cols * 2 * cols * W * sizeof(double)));
  HIPCHK(h->info.ensure(sizeof(int)));
  double* g = aug.as<double>();
  dim3 gg((cols + 15) / 16, (cols + 15) / 16), gp((rows + 15) / 16, (cols + 15) / 16);
  if (is_complex) {
    hipLaunchKernelGGL(qt::k_gemm<1>, gg, dim3(64), 0, h->stream, cols, cols, rows, dA, cols, 1, dA, cols, 0, g, 2 * cols);
    launch_gauss_jordan<1>(h, cols, g, h->info.as<int>());
    hipLaunchKernelGGL(qt::k_gemm<1>, gp, dim3(64), 0, h->stream, cols, rows, cols, g + (size_t)cols * 2, 2 * cols, 0, dA,
                       cols, 1, dout, rows);
  } else {
    hipLaunchKernelGGL(qt::k_gemm<0>, gg, dim3(64), 0, h->stream, cols, cols, rows, dA, cols, 1, dA, cols, 0, g, 2 * cols);
    launch_gauss_jordan<0>(h, cols, g, h->info.as<int>());
    hipLaunchKernelGGL(qt::k_gemm<0>, gp, dim3(64), 0, h->stream, cols, rows, cols, g + cols, 2 * cols, 0, dA, cols, 1,
                       dout, rows);
  }
  int info = 0;
  HIPCHK(hipMemcpyAsync(&info, h->info.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  if (int r = fetch_out(h, dout, out, nel, flags)) return r;
  HIPCHK(hipGetLastError());
  QT_STREAM_SYNC(h);
  if (info != 0) return fail(QT_ERR_SINGULAR, "A^T A is singular (no pivot in column %d)", info - 1);
  return 0;
}

// ---- process tomography: qt_process.h -------------------------------------------------------------
int qt_process_setup(qt_handle_t* h, const double* in_states, int flags) {
  QT_ENTER(h);
  if (int r = need_povm(h)) return r;
  if (!in_states) return fail(QT_ERR_ARG, "null in_states");
  if (h->nq > 3) return fail(QT_ERR_UNSUPPORTED, "process tomography supports n_qubits 1..3 in this release");
  h->proc_set = false;
  if (int r = ensure_dense(h)) return r;
  const int d = h->d, D = h->D, M = h->M;
  const size_t C2 = (size_t)D * D, R = (size_t)D * M;
  qt::ProcessState& ps = h->proc;
  ps.release();
  if (h->nq == 3) {
    // Kronecker-factored design matrix (qt_process64.h): L = (V_S (x) V_P) Pi^T, L^+ = Pi (V_S^+ (x) V_P^+)
    HIPCHK(hipMalloc(&ps.in_states, (size_t)D * D * 2 * sizeof(double)));
    HIPCHK(hipMalloc(&ps.emats, (size_t)M * D * 2 * sizeof(double)));
    HIPCHK(hipMalloc(&ps.vs_pinv, (size_t)D * D * 2 * sizeof(double)));
    HIPCHK(hipMalloc(&ps.vp_pinv, (size_t)D * M * 2 * sizeof(double)));
    HIPCHK(hipMalloc(&ps.vp_pinvT, (size_t)M * D * 2 * sizeof(double)));
    HIPCHK(hipMemcpyAsync(ps.in_states, in_states, (size_t)D * D * 2 * sizeof(double),
                          (flags & QT_DEVICE_PTR) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(qt::k_mat_from_bloch, dim3(grid_for((size_t)M * D)), dim3(256), 0, h->stream, h->nq, h->Aw.as<double>(),
                       M, (double*)ps.emats);
    int info[2] = {0, 0};
    if (int r = enqueue_left_inverse_complex(h, (const double*)ps.in_states, D, D, (double*)ps.vs_pinv)) return r;
    HIPCHK(hipMemcpyAsync(&info[0], h->info.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (int r = enqueue_left_inverse_complex(h, (const double*)ps.emats, M, D, (double*)ps.vp_pinv)) return r;
    HIPCHK(hipMemcpyAsync(&info[1], h->info.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    launch_transpose<2>(h, (const double*)ps.vp_pinv, D, M, (double*)ps.vp_pinvT);
    if (M % 4 == 0) {  // the operand of k_lifp64 (the matrix-core path of qt_lifp_batch)
      HIPCHK(hipMalloc(&ps.vp_perm, (size_t)4 * M * 32 * sizeof(double)));
      hipLaunchKernelGGL(qt::k_vp_perm, dim3(grid_for((size_t)4 * M * 32)), dim3(256), 0, h->stream, (const double*)ps.vp_pinvT, M,
                         D, d, 4, (double*)ps.vp_perm);
    }
    HIPCHK(hipGetLastError());
    QT_STREAM_SYNC(h);
    if (info[0] != 0) return fail(QT_ERR_SINGULAR, "input states do not span the operator space (column %d)", info[0] - 1);
    if (info[1] != 0) return fail(QT_ERR_SINGULAR, "POVM is not informationally complete (column %d)", info[1] - 1);
    ps.factored = true;
    h->proc_set = true;
    return 0;
  }
  HIPCHK(hipMalloc(&ps.in_states, (size_t)D * D * 2 * sizeof(double)));
  HIPCHK(hipMalloc(&ps.emats, (size_t)M * D * 2 * sizeof(double)));
  HIPCHK(hipMalloc(&ps.lifp, R * C2 * 2 * sizeof(double)));
  HIPCHK(hipMalloc(&ps.pinv, R * C2 * 2 * sizeof(double)));
  HIPCHK(hipMalloc(&ps.pinvT, R * C2 * 2 * sizeof(double)));
  HIPCHK(hipMalloc(&ps.aug, C2 * 2 * C2 * 2 * sizeof(double)));
  HIPCHK(h->info.ensure(sizeof(int)));
  HIPCHK(hipMemcpyAsync(ps.in_states, in_states, (size_t)D * D * 2 * sizeof(double),
                        (flags & QT_DEVICE_PTR) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
  double *lifp = (double*)ps.lifp, *pinv = (double*)ps.pinv, *pinvT = (double*)ps.pinvT, *aug = (double*)ps.aug;
  // E_m = sum_k A'[m][k] P_k  (process.py:204: Qobj(povm_bloch).matrix)
  hipLaunchKernelGGL(qt::k_mat_from_bloch, dim3(grid_for((size_t)M * D)), dim3(256), 0, h->stream, h->nq, h->Aw.as<double>(),
                     M, (double*)ps.emats);
  hipLaunchKernelGGL(qt::k_lifp_rows, dim3(grid_for(R * C2)), dim3(256), 0, h->stream, d, M, (const double*)ps.in_states,
                     (const double*)ps.emats, lifp);
  const int c2 = (int)C2, rr = (int)R;
  dim3 gg((c2 + 15) / 16, (c2 + 15) / 16), gp((rr + 15) / 16, (c2 + 15) / 16);
  hipLaunchKernelGGL(qt::k_gemm<1>, gg, dim3(64), 0, h->stream, c2, c2, rr, lifp, c2, 1, lifp, c2, 0, aug, 2 * c2);
  launch_gauss_jordan<1>(h, c2, aug, h->info.as<int>());
  hipLaunchKernelGGL(qt::k_gemm<1>, gp, dim3(64), 0, h->stream, c2, rr, c2, aug + (size_t)c2 * 2, 2 * c2, 0, lifp, c2, 1,
                     pinv, rr);
  launch_transpose<2>(h, pinv, c2, rr, pinvT);
  if (h->nq == 2) {  // the batched GEMM of qt_lifp_batch reads the left inverse in row-major Choi order
    HIPCHK(hipMalloc(&ps.pinvR, R * C2 * 2 * sizeof(double)));
    hipLaunchKernelGGL(qt::k_choi_order_rows, dim3(grid_for(R * C2)), dim3(256), 0, h->stream, (const double*)pinvT, R, D,
                       (double*)ps.pinvR);
  }
  int info = 0, finfo[2] = {0, 0};
  HIPCHK(hipMemcpyAsync(&info, h->info.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  const bool factors = h->nq == 2 && M % 4 == 0 && (size_t)M * 32 * sizeof(double) <= 64 * 1024;
  if (factors) {
    // the Kronecker factors of the left inverse (qt_process.h, k_lifp16): what qt_lifp_batch multiplies by at n = 2; the
    // dense operators above stay for qt_process_get_operators, 'pgdb' and the process chain
    HIPCHK(hipMalloc(&ps.vs_pinv, (size_t)D * D * 2 * sizeof(double)));
    HIPCHK(hipMalloc(&ps.vp_pinv, (size_t)D * M * 2 * sizeof(double)));
    HIPCHK(hipMalloc(&ps.vp_pinvT, (size_t)M * D * 2 * sizeof(double)));
    HIPCHK(hipMalloc(&ps.vp_perm, (size_t)M * 32 * sizeof(double)));
    if (int r = enqueue_left_inverse_complex(h, (const double*)ps.in_states, D, D, (double*)ps.vs_pinv)) return r;
    HIPCHK(hipMemcpyAsync(&finfo[0], h->info.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    if (int r = enqueue_left_inverse_complex(h, (const double*)ps.emats, M, D, (double*)ps.vp_pinv)) return r;
    HIPCHK(hipMemcpyAsync(&finfo[1], h->info.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    launch_transpose<2>(h, (const double*)ps.vp_pinv, D, M, (double*)ps.vp_pinvT);
    hipLaunchKernelGGL(qt::k_vp_perm, dim3(grid_for((size_t)M * 32)), dim3(256), 0, h->stream, (const double*)ps.vp_pinvT, M, D, d,
                       1, (double*)ps.vp_perm);
  }
  HIPCHK(hipGetLastError());
  QT_STREAM_SYNC(h);
  if (info != 0) return fail(QT_ERR_SINGULAR, "process design matrix is rank deficient (column %d): input states x POVM not complete", info - 1);
  if (factors && (finfo[0] != 0 || finfo[1] != 0)) {  // (cannot happen when the Kronecker product itself has full rank)
    (void)hipFree(ps.vp_perm);
    ps.vp_perm = nullptr;
  }
  h->proc_set = true;
  return 0;
}

int qt_process_get_operators(qt_handle_t* h, double* lifp_oper, double* lifp_oper_inv, int flags) {
  QT_ENTER(h);
  if (!h->proc_set) return fail(QT_ERR_STATE, "qt_process_setup has not been called");
  if (h->proc.factored)
    return fail(QT_ERR_UNSUPPORTED, "at n = 3 the design matrix is kept Kronecker-factored (qt_process_get_factors); its dense "
                                    "form would be 2 x 906 MB");
  const size_t bytes = (size_t)h->D * h->M * h->D * h->D * 2 * sizeof(double);
  const hipMemcpyKind kind = (flags & QT_DEVICE_PTR) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (lifp_oper) HIPCHK(hipMemcpyAsync(lifp_oper, h->proc.lifp, bytes, kind, h->stream));
  if (lifp_oper_inv) HIPCHK(hipMemcpyAsync(lifp_oper_inv, h->proc.pinv, bytes, kind, h->stream));
  return finish(h, flags);
}

int qt_process_prefer_dense(qt_handle_t* h, int on) {
  QT_ENTER(h);
  h->proc_dense = on != 0;
  return 0;
}

int qt_process_get_factors(qt_handle_t* h, double* vs_pinv, double* vp_pinv, int flags) {
  QT_ENTER(h);
  if (!h->proc_set) return fail(QT_ERR_STATE, "qt_process_setup has not been called");
  if (!h->proc.vs_pinv || !h->proc.vp_pinv)
    return fail(QT_ERR_UNSUPPORTED, "this set-up keeps the dense operator only (n = 1, or a POVM with M % 4 != 0 at n = 2): "
                                    "qt_process_get_operators");
  const hipMemcpyKind kind = (flags & QT_DEVICE_PTR) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (vs_pinv) HIPCHK(hipMemcpyAsync(vs_pinv, h->proc.vs_pinv, (size_t)h->D * h->D * 2 * sizeof(double), kind, h->stream));
  if (vp_pinv) HIPCHK(hipMemcpyAsync(vp_pinv, h->proc.vp_pinv, (size_t)h->D * h->M * 2 * sizeof(double), kind, h->stream));
  return finish(h, flags);
}

int qt_lifp_batch(qt_handle_t* h, const int64_t* counts, int B, int cptp, double* choi, int32_t* iters, int32_t* status,
                  int flags) {
  QT_ENTER(h);
  if (!h->proc_set) return fail(QT_ERR_STATE, "qt_process_setup has not been called");
  if (B < 0 || (B > 0 && (!counts || !choi))) return fail(QT_ERR_ARG, "bad lifp_batch arguments");
  if (B == 0) return 0;
  const int D = h->D, M = h->M;
  const int64_t* dc;
  double* dchoi;
  int32_t *dit, *dst;
  if (int r = stage_in(h, h->in0, counts, (size_t)B * D * M, flags, &dc)) return r;
  if (int r = stage_out(h, h->out0, choi, (size_t)B * D * D * 2, flags, &dchoi)) return r;
  if (int r = stage_out(h, h->out1, iters, (size_t)B, flags, &dit)) return r;
  if (int r = stage_out(h, h->out2, status, (size_t)B, flags, &dst)) return r;
  if (h->proc.factored) {  // n = 3: X = V_S^+ F V_P^+^T, two small products per process (qt_process64.h)
    if ((size_t)B * D > (size_t)1 << 26) return fail(QT_ERR_ARG, "batch too large");
    const int R = D * M;  // 13824: a multiple of 64, the pitch k_lifp_freq pads to
    double* raw = dchoi;
    if (cptp) {
      HIPCHK(h->ws_f.ensure((size_t)B * D * D * 2 * sizeof(double)));
      raw = h->ws_f.as<double>();
    }
    if (h->proc.vp_perm) {  // M % 4 == 0: both products of a process in one kernel on the matrix cores
      hipLaunchKernelGGL(qt::k_lifp64, dim3(4 * B), dim3(256), 0, h->stream, dc, B, M, (const double*)h->proc.vp_perm,
                         (const double*)h->proc.vs_pinv, raw, cptp ? (int32_t*)nullptr : dst, cptp ? (int32_t*)nullptr : dit);
    } else {
      HIPCHK(h->ws_x.ensure(((size_t)B * R + 192) * sizeof(double)));
      HIPCHK(h->ws_g.ensure((size_t)B * D * D * 2 * sizeof(double)));
      double *F = h->ws_x.as<double>(), *T = h->ws_g.as<double>();
      hipLaunchKernelGGL(qt::k_lifp_freq, dim3((B * D + 15) / 16), dim3(256), 0, h->stream, dc, B * D, M, D, R, F);
      // T[(b, s)][beta] = sum_m F[(b, s)][m] V_P^+[beta][m]: real x complex = a real GEMM with 2 D interleaved columns
      for (int b0 = 0; b0 < B; b0 += 8192) {  // (grid.y <= 65535 row tiles)
        const int nb = B - b0 < 8192 ? B - b0 : 8192;
        hipLaunchKernelGGL(qt::k_gemm<0>, dim3(2 * D / 16, (nb * D + 15) / 16), dim3(64), 0, h->stream, nb * D, 2 * D, M,
                           F + (size_t)b0 * R, M, 0, (const double*)h->proc.vp_pinvT, 2 * D, 0, T + (size_t)b0 * D * D * 2, 2 * D);
      }
      hipLaunchKernelGGL(qt::k_lifp_kron_finish, dim3(B), dim3(256), 0, h->stream, (const double*)T,
                         (const double*)h->proc.vs_pinv, B, raw, cptp ? (int32_t*)nullptr : dst, cptp ? (int32_t*)nullptr : dit);
    }
    if (cptp) {
      if (int r = allow_big_lds(qt::k_cptp_project64, qt::Proc64::kLdsBytes)) return r;
      HIPCHK(h->proc_ws.ensure((size_t)B * qt::Proc64::kWsComplex * 2 * sizeof(double)));  // Dykstra's p, q, y, x + the clip's input
      hipLaunchKernelGGL(qt::k_cptp_project64, dim3(B), dim3(qt::Proc64::NT), qt::Proc64::kLdsBytes, h->stream, (const double*)raw,
                         B, 0, 1000, 1e-12, dchoi, dit, dst, h->proc_ws.as<double>());
    }
    if (int r = fetch_out(h, dchoi, choi, (size_t)B * D * D * 2, flags)) return r;
    if (int r = fetch_out(h, dit, iters, (size_t)B, flags)) return r;
    if (int r = fetch_out(h, dst, status, (size_t)B, flags)) return r;
    if (int r = finish(h, flags)) return r;
    return count_bad(status, B, flags);
  }
  const size_t dyn = (size_t)D * M * sizeof(double);
  if (dyn > 32 * 1024) return fail(QT_ERR_UNSUPPORTED, "POVM has too many rows for the process kernel");
  const int R = D * M, Rp = (R + 63) / 64 * 64;
  const size_t gemm_lds = ((size_t)Rp * 16 + 4 * 256) * sizeof(double);
  const size_t gemm_lds2 = ((size_t)Rp * 32 + 4 * 512) * sizeof(double);  // two column tiles per workgroup
  if (D == 4) {
    hipLaunchKernelGGL(qt::k_lifp_batch<4>, dim3(B), dim3(qt::ProcWG<4>::NT), dyn, h->stream, dc, B, M,
                       (const double*)h->proc.pinvT, cptp, dchoi, dit, dst);
  } else if (h->proc.vp_perm && !h->proc_dense) {
    // n = 2 through the Kronecker factors of the left inverse: one wavefront per process (qt_process.h, k_lifp16)
    double* raw = dchoi;
    if (cptp) {
      HIPCHK(h->ws_g.ensure((size_t)B * 256 * 2 * sizeof(double)));
      raw = h->ws_g.as<double>();
    }
    int grid = (B + 3) / 4;
    if (grid > 768) grid = 768;  // three resident workgroups per CU (166 VGPRs): the wavefronts stride over the batch with the
                                 // next process's counts in flight, and a workgroup stages V_P^+ once for all its processes
    const size_t lds = (size_t)M * 32 * sizeof(double);
    int32_t *st = cptp ? (int32_t*)nullptr : dst, *it0 = cptp ? (int32_t*)nullptr : dit;
    if (M == 36)
      hipLaunchKernelGGL(qt::k_lifp16<9>, dim3(grid), dim3(256), lds, h->stream, dc, B, M, (const double*)h->proc.vp_perm,
                         (const double*)h->proc.vs_pinv, raw, st, it0);
    else
      hipLaunchKernelGGL(qt::k_lifp16<0>, dim3(grid), dim3(256), lds, h->stream, dc, B, M, (const double*)h->proc.vp_perm,
                         (const double*)h->proc.vs_pinv, raw, st, it0);
    if (cptp)
      hipLaunchKernelGGL(qt::k_cptp_wave16, dim3((B + 3) / 4), dim3(256), 0, h->stream, (const double*)raw, B, 0, 1000, 1e-12,
                         dchoi, dit, dst);
  } else if (B >= 256 && gemm_lds <= 152 * 1024) {
    // many processes: frequencies, then one FP64 MFMA GEMM over the batch, then (cptp) the projection kernel
    constexpr int NE = 256;
    HIPCHK(h->ws_x.ensure(((size_t)B * Rp + 192) * sizeof(double)));  // [B][Rp] + the zeros k_lifp_freq appends
    double* F = h->ws_x.as<double>();
    double* raw = dchoi;
    if (cptp) {
      HIPCHK(h->ws_g.ensure((size_t)B * NE * 2 * sizeof(double)));
      raw = h->ws_g.as<double>();
    }
    hipLaunchKernelGGL(qt::k_lifp_freq, dim3((B * D + 15) / 16), dim3(256), 0, h->stream, dc, B * D, M, D, Rp, F);
    // 4 groups of 16 processes per workgroup pass (x 2 halves of K).  A workgroup keeps its operand slice for up
    // to 4 passes once there are enough blocks to fill the chip anyway (measured: B = 1024 best with 1-2 passes,
    // 26 M/s; B = 8192 with 4, 38 M/s against 35 M/s with 1)
    const int nblocks = (B + 63) / 64;
    const int passes = nblocks >= 64 ? 4 : (nblocks >= 32 ? 2 : 1);
    const int row_blocks = (nblocks + passes - 1) / passes;
#ifdef QT_PHASE_TIMING
#define QT_GEMM_VARIANT(V)                                                                                                \
  case V:                                                                                                                 \
    if (int r = allow_big_lds(qt::k_lifp_gemm<16, 2, V>, gemm_lds2)) return r;                                             \
    hipLaunchKernelGGL((qt::k_lifp_gemm<16, 2, V>), dim3(2 * NE / 32, row_blocks), dim3(512), gemm_lds2, h->stream, F, B, R, Rp, \
                       (const double*)h->proc.pinvR, raw, cptp ? (int32_t*)nullptr : dst, cptp ? (int32_t*)nullptr : dit);  \
    break;
    if (gemm_lds2 <= kLdsLimit && g_host_diag != 0) {
      switch (g_host_diag) {
        QT_GEMM_VARIANT(1) QT_GEMM_VARIANT(2) QT_GEMM_VARIANT(3) QT_GEMM_VARIANT(4) QT_GEMM_VARIANT(7) QT_GEMM_VARIANT(8)
        QT_GEMM_VARIANT(15)
        default: return fail(QT_ERR_ARG, "no such diagnostic variant");
      }
    } else
#endif
    if (gemm_lds2 <= kLdsLimit) {  // two column tiles per workgroup: half the re-reads of F (R <= 576)
      if (int r = allow_big_lds(qt::k_lifp_gemm<16, 2>, gemm_lds2)) return r;
      hipLaunchKernelGGL((qt::k_lifp_gemm<16, 2>), dim3(2 * NE / 32, row_blocks), dim3(512), gemm_lds2, h->stream, F, B, R, Rp,
                         (const double*)h->proc.pinvR, raw, cptp ? (int32_t*)nullptr : dst, cptp ? (int32_t*)nullptr : dit);
    } else {
      if (int r = allow_big_lds(qt::k_lifp_gemm<16, 1>, gemm_lds)) return r;
      hipLaunchKernelGGL((qt::k_lifp_gemm<16, 1>), dim3(2 * NE / 16, row_blocks), dim3(512), gemm_lds, h->stream, F, B, R, Rp,
                         (const double*)h->proc.pinvR, raw, cptp ? (int32_t*)nullptr : dst, cptp ? (int32_t*)nullptr : dit);
    }
    if (cptp)
      hipLaunchKernelGGL(qt::k_cptp_wave16, dim3((B + 3) / 4), dim3(256), 0, h->stream, (const double*)raw, B, 0, 1000, 1e-12,
                         dchoi, dit, dst);
  } else {
    hipLaunchKernelGGL(qt::k_lifp_batch<16>, dim3(B), dim3(qt::ProcWG<16>::NT), dyn, h->stream, dc, B, M,
                       (const double*)h->proc.pinvT, cptp, dchoi, dit, dst);
  }
  if (int r = fetch_out(h, dchoi, choi, (size_t)B * D * D * 2, flags)) return r;
  if (int r = fetch_out(h, dit, iters, (size_t)B, flags)) return r;
  if (int r = fetch_out(h, dst, status, (size_t)B, flags)) return r;
  if (int r = finish(h, flags)) return r;
  return count_bad(status, B, flags);
}

int qt_pgdb_batch(qt_handle_t* h, const int64_t* counts, int B, int n_iter, double tol, int stop_rule, double* choi,
                  int32_t* iters, int32_t* status, int flags) {
  QT_ENTER(h);
  if (!h->proc_set) return fail(QT_ERR_STATE, "qt_process_setup has not been called");
  if (B < 0 || (B > 0 && (!counts || !choi))) return fail(QT_ERR_ARG, "bad pgdb_batch arguments");
  if (stop_rule != 0 && stop_rule != 1) return fail(QT_ERR_ARG, "stop_rule must be 0 (reference) or 1 (converged)");
  if (n_iter < 0) return fail(QT_ERR_ARG, "n_iter must be >= 0");
  if (B == 0) return 0;
  const int D = h->D, M = h->M;
  const int64_t* dc;
  double* dchoi;
  int32_t *dit, *dst;
  if (int r = stage_in(h, h->in0, counts, (size_t)B * D * M, flags, &dc)) return r;
  if (int r = stage_out(h, h->out0, choi, (size_t)B * D * D * 2, flags, &dchoi)) return r;
  if (int r = stage_out(h, h->out1, iters, (size_t)B, flags, &dit)) return r;
  if (int r = stage_out(h, h->out2, status, (size_t)B, flags, &dst)) return r;
  if (h->proc.factored) {  // n = 3: three launches per iteration over the batch, loop state on the device (qt_process64.h)
    using S = qt::Pgdb64;
    const size_t ne2 = (size_t)D * D * 2;
    HIPCHK(h->ws_x.ensure((size_t)B * S::ws_doubles(M) * sizeof(double)));
    HIPCHK(h->ws_g.ensure((size_t)B * ne2 * sizeof(double)));  // trial points c - g / mu
    HIPCHK(h->ws_f.ensure((size_t)B * ne2 * sizeof(double)));  // their CPTP projections
    HIPCHK(h->proc_ws.ensure((size_t)B * qt::Proc64::kWsComplex * 2 * sizeof(double)));
    HIPCHK(h->ws_act.ensure(((size_t)B * 4 + 4) * sizeof(int32_t)));
    int32_t* state = h->ws_act.as<int32_t>();
    int32_t* n_active = state + (size_t)B * 4;
    if (int r = allow_big_lds(qt::k_cptp_project64, qt::Proc64::kLdsBytes)) return r;
    if (int r = allow_big_lds(qt::k_pgdb64_grad, S::kLdsBytes)) return r;
    if (int r = allow_big_lds(qt::k_pgdb64_step, S::kLdsBytes)) return r;
    const double *vs = (const double*)h->proc.in_states, *vp = (const double*)h->proc.emats;
    hipLaunchKernelGGL(qt::k_pgdb64_init, dim3(B), dim3(256), 0, h->stream, B, dchoi, state, dit, dst, n_active);
    for (int it = 0; it < n_iter; ++it) {
      hipLaunchKernelGGL(qt::k_pgdb64_grad, dim3(B), dim3(S::NT), S::kLdsBytes, h->stream, dc, B, M, vs, vp, (const double*)dchoi,
                         (const int32_t*)state, h->ws_x.as<double>(), h->ws_g.as<double>());
      hipLaunchKernelGGL(qt::k_cptp_project64, dim3(B), dim3(qt::Proc64::NT), qt::Proc64::kLdsBytes, h->stream,
                         (const double*)h->ws_g.as<double>(), B, 0, 1000, 1e-12, h->ws_f.as<double>(), (int32_t*)nullptr,
                         (int32_t*)nullptr, h->proc_ws.as<double>());
      hipLaunchKernelGGL(qt::k_pgdb64_step, dim3(B), dim3(S::NT), S::kLdsBytes, h->stream, dc, B, M, vs, vp,
                         (const double*)h->ws_f.as<double>(), n_iter, tol, stop_rule, dchoi, state, h->ws_x.as<double>(), dit, dst,
                         n_active);
      int left = 0;
      HIPCHK(hipMemcpyAsync(&left, n_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      QT_STREAM_SYNC(h);
      if (left <= 0) break;
    }
    HIPCHK(hipGetLastError());
    if (int r = fetch_out(h, dchoi, choi, (size_t)B * D * D * 2, flags)) return r;
    if (int r = fetch_out(h, dit, iters, (size_t)B, flags)) return r;
    if (int r = fetch_out(h, dst, status, (size_t)B, flags)) return r;
    if (int r = finish(h, flags)) return r;
    return count_bad(status, B, flags);
  }
  const size_t dyn = (size_t)4 * D * M * sizeof(double);
  if (dyn > 32 * 1024) return fail(QT_ERR_UNSUPPORTED, "POVM has too many rows for the process kernel");
  if (D == 4)
    hipLaunchKernelGGL(qt::k_pgdb_batch<4>, dim3(B), dim3(qt::ProcWG<4>::NT), dyn, h->stream, dc, B, M,
                       (const double*)h->proc.lifp, n_iter, tol, stop_rule, dchoi, dit, dst);
  else
    hipLaunchKernelGGL(qt::k_pgdb_batch<16>, dim3(B), dim3(qt::ProcWG<16>::NT), dyn, h->stream, dc, B, M,
                       (const double*)h->proc.lifp, n_iter, tol, stop_rule, dchoi, dit, dst);
  if (int r = fetch_out(h, dchoi, choi, (size_t)B * D * D * 2, flags)) return r;
  if (int r = fetch_out(h, dit, iters, (size_t)B, flags)) return r;
  if (int r = fetch_out(h, dst, status, (size_t)B, flags)) return r;
  if (int r = finish(h, flags)) return r;
  return count_bad(status, B, flags);
}

int qt_pgdb_pieces(qt_handle_t* h, const int64_t* counts, int B, const double* choi_in, double* probas, double* grad,
                   double* projected, int flags) {
  QT_ENTER(h);
  if (!h->proc_set) return fail(QT_ERR_STATE, "qt_process_setup has not been called");
  if (B < 0 || (B > 0 && (!counts || !choi_in))) return fail(QT_ERR_ARG, "bad pgdb_pieces arguments");
  if (!h->proc.factored) return fail(QT_ERR_UNSUPPORTED, "qt_pgdb_pieces inspects the factored (n = 3) iteration");
  if (B == 0) return 0;
  using S = qt::Pgdb64;
  const int D = h->D, M = h->M, R = D * M;
  const size_t ne2 = (size_t)D * D * 2, wsd = S::ws_doubles(M);
  const int64_t* dc;
  const double* dcur;
  if (int r = stage_in(h, h->in0, counts, (size_t)B * R, flags, &dc)) return r;
  if (int r = stage_in(h, h->in1, choi_in, (size_t)B * ne2, flags, &dcur)) return r;
  HIPCHK(h->ws_x.ensure((size_t)B * wsd * sizeof(double)));
  HIPCHK(h->ws_g.ensure((size_t)B * ne2 * sizeof(double)));
  HIPCHK(h->ws_f.ensure((size_t)B * ne2 * sizeof(double)));
  HIPCHK(h->proc_ws.ensure((size_t)B * qt::Proc64::kWsComplex * 2 * sizeof(double)));
  HIPCHK(h->ws_act.ensure(((size_t)B * 4 + 4) * sizeof(int32_t)));
  int32_t* state = h->ws_act.as<int32_t>();
  HIPCHK(hipMemsetAsync(state, 0, ((size_t)B * 4 + 4) * sizeof(int32_t), h->stream));
  if (int r = allow_big_lds(qt::k_cptp_project64, qt::Proc64::kLdsBytes)) return r;
  if (int r = allow_big_lds(qt::k_pgdb64_grad, S::kLdsBytes)) return r;
  hipLaunchKernelGGL(qt::k_pgdb64_grad, dim3(B), dim3(S::NT), S::kLdsBytes, h->stream, dc, B, M, (const double*)h->proc.in_states,
                     (const double*)h->proc.emats, dcur, (const int32_t*)state, h->ws_x.as<double>(), h->ws_g.as<double>());
  hipLaunchKernelGGL(qt::k_cptp_project64, dim3(B), dim3(qt::Proc64::NT), qt::Proc64::kLdsBytes, h->stream,
                     (const double*)h->ws_g.as<double>(), B, 0, 1000, 1e-12, h->ws_f.as<double>(), (int32_t*)nullptr,
                     (int32_t*)nullptr, h->proc_ws.as<double>());
  HIPCHK(hipGetLastError());
  const hipMemcpyKind kind = (flags & QT_DEVICE_PTR) ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  const double* ws = h->ws_x.as<double>();
  if (probas)
    HIPCHK(hipMemcpy2DAsync(probas, (size_t)R * sizeof(double), ws + (size_t)2 * D * M, wsd * sizeof(double),
                            (size_t)R * sizeof(double), B, kind, h->stream));
  if (grad)
    HIPCHK(hipMemcpy2DAsync(grad, ne2 * sizeof(double), ws + (size_t)2 * D * M + 3 * (size_t)R, wsd * sizeof(double),
                            ne2 * sizeof(double), B, kind, h->stream));
  if (projected) HIPCHK(hipMemcpyAsync(projected, h->ws_f.p, (size_t)B * ne2 * sizeof(double), kind, h->stream));
  return finish(h, flags);
}

int qt_mhmc_process(qt_handle_t* h, const int64_t* counts, int C, const double* choi_init, const double* deltas,
                    const double* uniforms, int T, double step, double* chain, int32_t* accepted, int flags) {
  QT_ENTER(h);
  if (!h->proc_set) return fail(QT_ERR_STATE, "qt_process_setup has not been called");
  if (C < 0 || T < 0 || (C > 0 && T > 0 && (!counts || !choi_init || !deltas || !uniforms || !chain || !accepted)))
    return fail(QT_ERR_ARG, "bad mhmc_process arguments");
  if (C == 0 || T == 0) return 0;
  const int D = h->D, M = h->M;
  const size_t ne = (size_t)D * D;
  const int64_t* dc;
  const double *dx, *dd, *du;
  double* dch;
  int32_t* dacc;
  if (int r = stage_in(h, h->in0, counts, (size_t)C * D * M, flags, &dc)) return r;
  if (int r = stage_in(h, h->in1, choi_init, (size_t)C * ne * 2, flags, &dx)) return r;
  if (int r = stage_in(h, h->out2, deltas, (size_t)C * T * ne, flags, &dd)) return r;
  if (int r = stage_in(h, h->out3, uniforms, (size_t)C * T, flags, &du)) return r;
  if (int r = stage_out(h, h->out0, chain, (size_t)C * T * ne * 2, flags, &dch)) return r;
  if (int r = stage_out(h, h->out1, accepted, (size_t)C * T, flags, &dacc)) return r;
  if (h->proc.factored) {  // n = 3: three launches per step, the chain's state stays on the device (qt_process64.h)
    using S = qt::Pgdb64;
    const int nt = qt::Fwd64::tiles(M);
    HIPCHK(h->ws_x.ensure(((size_t)C * S::ws_doubles(M) + C + (size_t)C * nt) * sizeof(double)));
    HIPCHK(h->ws_g.ensure((size_t)C * ne * 2 * sizeof(double)));     // proposals before the projection
    HIPCHK(h->ws_f.ensure((size_t)C * ne * 2 * sizeof(double)));     // ... and after it
    HIPCHK(h->hess.ensure((size_t)C * ne * 2 * sizeof(double)));     // the chains' current points
    HIPCHK(h->proc_ws.ensure((size_t)C * qt::Proc64::kWsComplex * 2 * sizeof(double)));
    if (int r = allow_big_lds(qt::k_cptp_project64, qt::Proc64::kLdsBytes)) return r;
    if (int r = allow_big_lds(qt::k_fwd64_nll, qt::Fwd64::kLdsBytes)) return r;
    double *ws = h->ws_x.as<double>(), *fcur = ws + (size_t)C * S::ws_doubles(M), *fpart = fcur + C, *x = h->hess.as<double>();
    const double *vs = (const double*)h->proc.in_states, *vp = (const double*)h->proc.emats;
    HIPCHK(hipMemcpyAsync(x, dx, (size_t)C * ne * 2 * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    // the model and the NLL of a point: C x ceil(M / 16) workgroups on the matrix cores (k_fwd64_nll), then the accept test
    hipLaunchKernelGGL(qt::k_fwd64_nll, dim3(C * nt), dim3(256), qt::Fwd64::kLdsBytes, h->stream, dc, C, M, vs, vp,
                       (const double*)x, (const double*)nullptr, 0, fpart);
    hipLaunchKernelGGL(qt::k_mhmc64_decide, dim3(C), dim3(256), 0, h->stream, C, nt, T, -1, (const double*)fpart,
                       (const double*)nullptr, du, x, fcur, dch, dacc);
    for (int t = 0; t < T; ++t) {
      hipLaunchKernelGGL(qt::k_mhmc64_propose, dim3(C), dim3(256), 0, h->stream, C, T, t, step, (const double*)x, dd,
                         h->ws_g.as<double>());
      hipLaunchKernelGGL(qt::k_cptp_project64, dim3(C), dim3(qt::Proc64::NT), qt::Proc64::kLdsBytes, h->stream,
                         (const double*)h->ws_g.as<double>(), C, 0, 1000, 1e-12, h->ws_f.as<double>(), (int32_t*)nullptr,
                         (int32_t*)nullptr, h->proc_ws.as<double>());
      hipLaunchKernelGGL(qt::k_fwd64_nll, dim3(C * nt), dim3(256), qt::Fwd64::kLdsBytes, h->stream, dc, C, M, vs, vp,
                         (const double*)x, (const double*)h->ws_f.as<double>(), 1, fpart);
      hipLaunchKernelGGL(qt::k_mhmc64_decide, dim3(C), dim3(256), 0, h->stream, C, nt, T, t, (const double*)fpart,
                         (const double*)h->ws_f.as<double>(), du, x, fcur, dch, dacc);
    }
    HIPCHK(hipGetLastError());
    if (int r = fetch_out(h, dch, chain, (size_t)C * T * ne * 2, flags)) return r;
    if (int r = fetch_out(h, dacc, accepted, (size_t)C * T, flags)) return r;
    return finish(h, flags);
  }
  const size_t dyn = (size_t)2 * D * M * sizeof(double);
  if (dyn > 32 * 1024) return fail(QT_ERR_UNSUPPORTED, "POVM has too many rows for the process kernel");
  if (D == 4)
    hipLaunchKernelGGL(qt::k_mhmc_process<4>, dim3(C), dim3(qt::ProcWG<4>::NT), dyn, h->stream, dc, C, M,
                       (const double*)h->proc.lifp, dx, dd, du, T, step, dch, dacc);
  else
    hipLaunchKernelGGL(qt::k_mhmc_process<16>, dim3(C), dim3(qt::ProcWG<16>::NT), dyn, h->stream, dc, C, M,
                       (const double*)h->proc.lifp, dx, dd, du, T, step, dch, dacc);
  if (int r = fetch_out(h, dch, chain, (size_t)C * T * ne * 2, flags)) return r;
  if (int r = fetch_out(h, dacc, accepted, (size_t)C * T, flags)) return r;
  return finish(h, flags);
}

int qt_cptp_project_batch(qt_handle_t* h, const double* choi_in, int B, int mode, int n_iter, double tol, double* choi_out,
                          int32_t* iters, int flags) {
  QT_ENTER(h);
  if (B < 0 || (B > 0 && (!choi_in || !choi_out))) return fail(QT_ERR_ARG, "bad cptp_project arguments");
  if (mode < 0 || mode > 2) return fail(QT_ERR_ARG, "mode must be 0 (CPTP), 1 (TP) or 2 (CP)");
  if (h->nq > 3) return fail(QT_ERR_UNSUPPORTED, "process tomography supports n_qubits 1..3 in this release");
  if (B == 0) return 0;
  const int D = h->D;
  const double* din;
  double* dout;
  int32_t* dit;
  if (int r = stage_in(h, h->in0, choi_in, (size_t)B * D * D * 2, flags, &din)) return r;
  if (int r = stage_out(h, h->out0, choi_out, (size_t)B * D * D * 2, flags, &dout)) return r;
  if (int r = stage_out(h, h->out1, iters, (size_t)B, flags, &dit)) return r;
  if (D == 64) {
    if (int r = allow_big_lds(qt::k_cptp_project64, qt::Proc64::kLdsBytes)) return r;
    if (mode != 1) HIPCHK(h->proc_ws.ensure((size_t)B * qt::Proc64::kWsComplex * 2 * sizeof(double)));  // Dykstra's p, q, y, x + the clip's input
    hipLaunchKernelGGL(qt::k_cptp_project64, dim3(B), dim3(qt::Proc64::NT), qt::Proc64::kLdsBytes, h->stream, din, B, mode, n_iter,
                       tol, dout, dit, (int32_t*)nullptr, h->proc_ws.as<double>());
  } else if (D == 4)
    hipLaunchKernelGGL(qt::k_cptp_project<4>, dim3(B), dim3(qt::ProcWG<4>::NT), 0, h->stream, din, B, mode, n_iter, tol, dout,
                       dit);
  else
    hipLaunchKernelGGL(qt::k_cptp_wave16, dim3((B + 3) / 4), dim3(256), 0, h->stream, din, B, mode, n_iter, tol, dout, dit,
                       (int32_t*)nullptr);
  if (int r = fetch_out(h, dout, choi_out, (size_t)B * D * D * 2, flags)) return r;
  if (int r = fetch_out(h, dit, iters, (size_t)B, flags)) return r;
  return finish(h, flags);
}

}  // extern "C"

#ifdef QT_PHASE_TIMING
extern "C" int qt_debug_set_diag(int v) {  // which compile-time variant of k_lifp_gemm<16, 2> the next qt_lifp_batch launches
  g_host_diag = v;
  return 0;
}
// profile build only (scripts/phase_timing.py): where the kernels drop their phase stamps
extern "C" int qt_debug_set_prof(void* device_ptr) {
  long long* p = static_cast<long long*>(device_ptr);
  return hipMemcpyToSymbol(HIP_SYMBOL(qt::g_qt_prof), &p, sizeof(p)) == hipSuccess ? 0 : -3;
}
#endif
