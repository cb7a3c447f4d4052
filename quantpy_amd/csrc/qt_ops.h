// Size-generic kernels (any n <= 5): operator assembly, POVM weighting, Gram / left inverse,
// Born probabilities, matrix <-> Bloch conversion, Hilbert-Schmidt distance.
// All are HBM/L2-streaming kernels: coalesced along the fastest axis, LDS tiles where an
// operand is re-used.  Reference lines are cited per kernel (paths into /root/reference/quantpy).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "qt_small.h"  // gsum / dpp reductions

namespace qt {

// ---- a1: routines.py:14-19 ----------------------------------------------------------------
// out[k][r][c] = (c == r ^ x(k)) ? (-i)^ny(k) (-1)^popc(r & z(k)) : 0.  One thread per complex element, one
// 16-byte store per lane (a wavefront writes 1 KB contiguous); d is a power of two, so the three indices are
// shifts and masks of the flat element number (the first version divided 64-bit numbers by a run-time d).
__global__ void __launch_bounds__(256) k_pauli_basis(int nq, double* __restrict__ out) {
  const unsigned d = 1u << nq, dm = d - 1;
  const size_t total = (size_t)1 << (4 * nq);
  double2* o2 = reinterpret_cast<double2*>(out);
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const unsigned c = (unsigned)e & dm, r = (unsigned)(e >> nq) & dm, k = (unsigned)(e >> (2 * nq));
    unsigned xm = 0, zm = 0, ny = 0;
    for (int b = 0; b < nq; ++b) {
      const unsigned dig = (k >> (2 * b)) & 3;
      xm |= (unsigned)(dig == 1 || dig == 2) << b;
      zm |= (unsigned)(dig >= 2) << b;
      ny += dig == 2;
    }
    double re = 0.0, im = 0.0;
    if (c == (r ^ xm)) {
      const double sg = (__popc(r & zm) & 1) ? -1.0 : 1.0;
      switch (ny & 3) {
        case 0: re = sg; break;
        case 1: im = -sg; break;
        case 2: re = -sg; break;
        default: im = sg; break;
      }
    }
    o2[e] = double2{re, im};
  }
}

// ---- a2: measurements.py:88-93 --------------------------------------------------------------
// out[s][k][j] = prod_q povm1[s_q][k_q][j_q], digits most-significant first, multiplied left to
// right exactly as repeated np.kron does ((a*b)*c...): bit-identical entries.
// HBM-write bound (8 S K D bytes, 63.7 MB at n = 5).  A lane produces TWO neighbouring entries (they share every
// factor but the last) and stores them as one 16-byte word, so a wavefront instruction writes 1 KB contiguous.
// No division anywhere: the row's n digits r_q = s_q K1 + k_q come packed (8 bits each) from a table the host
// builds once per (S1, K1, n), the Pauli digits j_q are bit fields of the column, the one-qubit table sits in LDS.
__global__ void __launch_bounds__(256) k_povm_kron(int nq, const double* __restrict__ p1, int R1,
                                                   const unsigned long long* __restrict__ rowdig, size_t total,
                                                   double* __restrict__ out) {
  __shared__ double tab[256 * 4];
  for (int e = threadIdx.x; e < R1 * 4; e += blockDim.x) tab[e] = p1[e];
  __syncthreads();
  const unsigned dmask = (1u << (2 * nq)) - 1;
  double2* o2 = reinterpret_cast<double2*>(out);
  const size_t npairs = total >> 1;
  for (size_t pidx = blockIdx.x * (size_t)blockDim.x + threadIdx.x; pidx < npairs; pidx += (size_t)gridDim.x * blockDim.x) {
    const size_t e = pidx << 1;
    const unsigned j = (unsigned)e & dmask;
    const unsigned long long dig = rowdig[e >> (2 * nq)];
    double v = 0.0;
    for (int q = 0; q + 1 < nq; ++q) {
      const unsigned r = (unsigned)(dig >> (8 * q)) & 255u, jq = (j >> (2 * (nq - 1 - q))) & 3u;
      const double f = tab[r * 4 + jq];
      v = (q == 0) ? f : v * f;
    }
    const unsigned rl = (unsigned)(dig >> (8 * (nq - 1))) & 255u, jl = j & 3u;  // jl is 0 or 2
    const double f0 = tab[rl * 4 + jl], f1 = tab[rl * 4 + jl + 1];
    o2[pidx] = (nq == 1) ? double2{f0, f1} : double2{v * f0, v * f1};
  }
}

// ---- state.py:194-197 and the transposes the dense kernels read: from A [M][D] (row m = s K + k) produce
// A^T [D][M], A' = A * Ns[s] / sum(Ns) [M][D] and A'^T [D][M] in ONE pass: a 64 x 64 tile goes through LDS
// (pitch 65: the transposed read is conflict-free), so every global access is a full 512-byte row segment.
// `tot` = sum(Ns), summed once on the host (the first version had every thread re-add Ns[0..S)).
__global__ void __launch_bounds__(256) k_povm_setup(const double* __restrict__ A, const double* __restrict__ Ns, double tot,
                                                    int K, int M, int D, double* __restrict__ AT, double* __restrict__ Aw,
                                                    double* __restrict__ AwT) {
  __shared__ double ta[64][65], tw[64][65];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int m0 = blockIdx.y * 64, k0 = blockIdx.x * 64;
  for (int r = ty; r < 64; r += 4) {
    const int m = m0 + r, k = k0 + tx;
    if (m < M && k < D) {
      const double a = A[(size_t)m * D + k];
      const double w = a * Ns[m / K] / tot;
      ta[r][tx] = a;
      tw[r][tx] = w;
      Aw[(size_t)m * D + k] = w;
    }
  }
  __syncthreads();
  for (int r = ty; r < 64; r += 4) {
    const int k = k0 + r, m = m0 + tx;
    if (k < D && m < M) {
      AT[(size_t)k * M + m] = ta[tx][r];
      AwT[(size_t)k * M + m] = tw[tx][r];
    }
  }
}

// out[C][R] = in[R][C]^T through an LDS tile (64 x 64 real, 32 x 32 complex; odd pitch: the transposed read is
// conflict-free).  W = 1 real, W = 2 complex (interleaved).
template <int W>
__global__ void __launch_bounds__(256) k_transpose_tiled(const double* __restrict__ in, int R, int C, double* __restrict__ out) {
  constexpr int TS = 64 / W, STEP = 256 / TS;
  __shared__ double tile[TS][TS * W + 1];
  const int tx = threadIdx.x % TS, ty = threadIdx.x / TS;
  const int r0 = blockIdx.y * TS, c0 = blockIdx.x * TS;
  for (int r = ty; r < TS; r += STEP) {
    const int rr = r0 + r, cc = c0 + tx;
    if (rr < R && cc < C) {
#pragma unroll
      for (int w = 0; w < W; ++w) tile[r][tx * W + w] = in[((size_t)rr * C + cc) * W + w];
    }
  }
  __syncthreads();
  for (int r = ty; r < TS; r += STEP) {
    const int cc = c0 + r, rr = r0 + tx;
    if (cc < C && rr < R) {
#pragma unroll
      for (int w = 0; w < W; ++w) out[((size_t)cc * R + rr) * W + w] = tile[tx][r * W + w];
    }
  }
}

// ---- GEMM  C[MxN] = op(A) op(B) on the FP64 matrix cores, real (CPLX = 0) or complex (CPLX = 1) ----
// The one true GEMM of the path: the Gram matrix A^T A and the product inv(A^T A) A^T of the left
// inverse (routines.py:69-71) -- 256 x 256 x 576 complex for the C3 process design matrix
// (process.py:208-210), 64 x 64 x 216 real for the C2 POVM.  Row-major operands with explicit leading
// dimensions; TA / TB select PLAIN transposes (never conjugated: routines.py:71 uses A.T, and the
// design matrix is complex).  One wavefront owns one 16 x 16 tile of C and walks K four at a time with
// v_mfma_f64_16x16x4_f64: lane l feeds A[row0 + l%16][k0 + l/16] and B[k0 + l/16][col0 + l%16] and
// holds C[row0 + l/16 + 4 r][col0 + l%16], r = 0..3.  Complex = four real products into four
// accumulator tiles (rr, ii, ri, ir).  Edges are zero-padded by predicated loads.
typedef double v4f64 __attribute__((ext_vector_type(4)));

template <int CPLX>
__global__ void __launch_bounds__(64) k_gemm(int M, int N, int K, const double* __restrict__ A, int lda, int ta,
                                             const double* __restrict__ B, int ldb, int tb, double* __restrict__ C,
                                             int ldc) {
  constexpr int W = CPLX ? 2 : 1;
  const int lane = threadIdx.x;
  const int row0 = blockIdx.y * 16, col0 = blockIdx.x * 16;
  const int ar = row0 + (lane & 15), bc = col0 + (lane & 15), kk = lane >> 4;
  v4f64 acc_rr = {0.0, 0.0, 0.0, 0.0}, acc_ii = acc_rr, acc_ri = acc_rr, acc_ir = acc_rr;
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + kk;
    double a_re = 0.0, a_im = 0.0, b_re = 0.0, b_im = 0.0;
    if (ar < M && k < K) {
      const size_t idx = ta ? ((size_t)k * lda + ar) : ((size_t)ar * lda + k);
      a_re = A[idx * W];
      if (CPLX) a_im = A[idx * W + 1];
    }
    if (k < K && bc < N) {
      const size_t idx = tb ? ((size_t)bc * ldb + k) : ((size_t)k * ldb + bc);
      b_re = B[idx * W];
      if (CPLX) b_im = B[idx * W + 1];
    }
    acc_rr = __builtin_amdgcn_mfma_f64_16x16x4f64(a_re, b_re, acc_rr, 0, 0, 0);
    if (CPLX) {
      acc_ii = __builtin_amdgcn_mfma_f64_16x16x4f64(a_im, b_im, acc_ii, 0, 0, 0);
      acc_ri = __builtin_amdgcn_mfma_f64_16x16x4f64(a_re, b_im, acc_ri, 0, 0, 0);
      acc_ir = __builtin_amdgcn_mfma_f64_16x16x4f64(a_im, b_re, acc_ir, 0, 0, 0);
    }
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = row0 + (lane >> 4) + 4 * r, col = col0 + (lane & 15);
    if (row < M && col < N) {
      const size_t idx = (size_t)row * ldc + col;
      if (CPLX) {
        C[idx * W] = acc_rr[r] - acc_ii[r];
        C[idx * W + 1] = acc_ri[r] + acc_ir[r];
      } else {
        C[idx] = acc_rr[r];
      }
    }
  }
}

// ---- in-place inverse by Gauss-Jordan with partial (row) pivoting ---------------------------
// One 1024-thread workgroup; `aug` is the n x 2n augmented matrix [G | I] in global memory
// (row-major, complex interleaved when CPLX).  On exit the right half holds inv(G).
// info = 0 ok, k+1 if no usable pivot in column k.  (scipy.linalg.inv = LAPACK getrf/getri with
// the same pivoting rule; routines.py:71.)
template <int CPLX>
__global__ void __launch_bounds__(1024) k_gauss_jordan(int n, double* __restrict__ aug, int* __restrict__ info) {
  constexpr int W = CPLX ? 2 : 1;
  __shared__ double s_val[1024];
  __shared__ int s_idx[1024];
  __shared__ double s_piv[2];
  __shared__ int s_prow;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n2 = 2 * n;
  for (int e = tid; e < n * n; e += nt) {  // right half = identity
    const int r = e / n, c = e % n;
    aug[((size_t)r * n2 + n + c) * W] = (r == c) ? 1.0 : 0.0;
    if (CPLX) aug[((size_t)r * n2 + n + c) * W + 1] = 0.0;
  }
  if (tid == 0) *info = 0;
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    // pivot: max |a_ik| over i >= k  (LAPACK izamax uses |re| + |im| for complex)
    double best = -1.0;
    int bi = k;
    for (int i = k + tid; i < n; i += nt) {
      const double* p = aug + ((size_t)i * n2 + k) * W;
      const double v = CPLX ? fabs(p[0]) + fabs(p[W - 1]) : fabs(p[0]);
      if (v > best) {
        best = v;
        bi = i;
      }
    }
    s_val[tid] = best;
    s_idx[tid] = bi;
    __syncthreads();
    for (int s = nt / 2; s > 0; s >>= 1) {
      if (tid < s) {
        const double o = s_val[tid + s];
        const int oi = s_idx[tid + s];
        if (o > s_val[tid] || (o == s_val[tid] && oi < s_idx[tid])) {
          s_val[tid] = o;
          s_idx[tid] = oi;
        }
      }
      __syncthreads();
    }
    if (tid == 0) {
      s_prow = s_idx[0];
      if (!(s_val[0] > 0.0) && *info == 0) *info = k + 1;
    }
    __syncthreads();
    const int pr = s_prow;
    if (pr != k) {
      for (int c = tid; c < n2 * W; c += nt) {
        const double t = aug[(size_t)k * n2 * W + c];
        aug[(size_t)k * n2 * W + c] = aug[(size_t)pr * n2 * W + c];
        aug[(size_t)pr * n2 * W + c] = t;
      }
    }
    __syncthreads();
    if (tid == 0) {
      s_piv[0] = aug[((size_t)k * n2 + k) * W];
      if (CPLX) s_piv[1] = aug[((size_t)k * n2 + k) * W + 1];
    }
    __syncthreads();
    {  // scale the pivot row by 1 / pivot
      const double pr_ = s_piv[0], pi_ = CPLX ? s_piv[1] : 0.0;
      const double den = pr_ * pr_ + pi_ * pi_;
      for (int c = tid; c < n2; c += nt) {
        double* p = aug + ((size_t)k * n2 + c) * W;
        if (CPLX) {
          const double xr = p[0], xi = p[W - 1];
          p[0] = (xr * pr_ + xi * pi_) / den;
          p[W - 1] = (xi * pr_ - xr * pi_) / den;
        } else {
          p[0] = p[0] / pr_;
        }
      }
    }
    __syncthreads();
    // eliminate column k from every other row; columns < k of the left half are already e_j
    const int c0 = k, ncols = n2 - k;
    const size_t work = (size_t)n * ncols;
    for (size_t e = tid; e < work; e += nt) {
      const int i = (int)(e / ncols), c = c0 + (int)(e % ncols);
      if (i == k || c == k) continue;
      const double* fk = aug + ((size_t)i * n2 + k) * W;
      const double* rk = aug + ((size_t)k * n2 + c) * W;
      double* p = aug + ((size_t)i * n2 + c) * W;
      if (CPLX) {
        const double fr = fk[0], fi = fk[W - 1], rr = rk[0], ri = rk[W - 1];
        p[0] -= fr * rr - fi * ri;
        p[W - 1] -= fr * ri + fi * rr;
      } else {
        p[0] -= fk[0] * rk[0];
      }
    }
    __syncthreads();
    for (int i = tid; i < n; i += nt) {  // now clear column k itself
      if (i == k) continue;
      aug[((size_t)i * n2 + k) * W] = 0.0;
      if (CPLX) aug[((size_t)i * n2 + k) * W + 1] = 0.0;
    }
    __syncthreads();
  }
}

// The same elimination for large n (the 256 x 256 complex Gram matrix of 2-qubit process tomography), spread over
// the chip: one workgroup cannot keep 256 x 512 complex elements moving (12.5 ms: 48 us per pivot step, every
// element through L2 by 1024 threads).  Per pivot step two launches on the handle's stream -- k_gj_pivot (one
// workgroup: pivot search with the same rule, row swap, pivot row scaled) and k_gj_eliminate (one workgroup per
// row) -- preceded by k_gj_identity.  Same arithmetic per element as k_gauss_jordan, so the same inverse.
template <int CPLX>
__global__ void k_gj_identity(int n, double* __restrict__ aug, int* __restrict__ info) {
  constexpr int W = CPLX ? 2 : 1;
  const int n2 = 2 * n;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n * n; e += gridDim.x * blockDim.x) {
    const int r = e / n, c = e % n;
    aug[((size_t)r * n2 + n + c) * W] = (r == c) ? 1.0 : 0.0;
    if (CPLX) aug[((size_t)r * n2 + n + c) * W + 1] = 0.0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) *info = 0;
}

template <int CPLX>
__global__ void __launch_bounds__(1024) k_gj_pivot(int n, double* __restrict__ aug, int k, int* __restrict__ info) {
  constexpr int W = CPLX ? 2 : 1;
  __shared__ double s_val[1024];
  __shared__ int s_idx[1024];
  __shared__ double s_piv[2];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n2 = 2 * n;
  double best = -1.0;  // max |a_ik| over i >= k  (LAPACK izamax: |re| + |im| for complex), lowest row on ties
  int bi = k;
  for (int i = k + tid; i < n; i += nt) {
    const double* p = aug + ((size_t)i * n2 + k) * W;
    const double v = CPLX ? fabs(p[0]) + fabs(p[W - 1]) : fabs(p[0]);
    if (v > best) {
      best = v;
      bi = i;
    }
  }
  s_val[tid] = best;
  s_idx[tid] = bi;
  __syncthreads();
  for (int s = nt / 2; s > 0; s >>= 1) {
    if (tid < s) {
      const double o = s_val[tid + s];
      const int oi = s_idx[tid + s];
      if (o > s_val[tid] || (o == s_val[tid] && oi < s_idx[tid])) {
        s_val[tid] = o;
        s_idx[tid] = oi;
      }
    }
    __syncthreads();
  }
  const int pr = s_idx[0];
  if (tid == 0 && !(s_val[0] > 0.0) && *info == 0) *info = k + 1;
  if (tid == 0) {
    s_piv[0] = aug[((size_t)pr * n2 + k) * W];
    if (CPLX) s_piv[1] = aug[((size_t)pr * n2 + k) * W + 1];
  }
  __syncthreads();
  const double pr_ = s_piv[0], pi_ = CPLX ? s_piv[1] : 0.0;
  const double den = pr_ * pr_ + pi_ * pi_;
  for (int c = tid; c < n2; c += nt) {  // rows k and pr change places; the new row k is divided by the pivot
    double* pk = aug + ((size_t)k * n2 + c) * W;
    double* pp = aug + ((size_t)pr * n2 + c) * W;
    const double kr = pk[0], ki = CPLX ? pk[W - 1] : 0.0;
    const double xr = pp[0], xi = CPLX ? pp[W - 1] : 0.0;
    if (pr != k) {
      pp[0] = kr;
      if (CPLX) pp[W - 1] = ki;
    }
    if (CPLX) {
      pk[0] = (xr * pr_ + xi * pi_) / den;
      pk[W - 1] = (xi * pr_ - xr * pi_) / den;
    } else {
      pk[0] = xr / pr_;
    }
  }
}

template <int CPLX>
__global__ void __launch_bounds__(256) k_gj_eliminate(int n, double* __restrict__ aug, int k) {
  constexpr int W = CPLX ? 2 : 1;
  const int i = blockIdx.x, n2 = 2 * n;
  if (i == k) return;
  double* row = aug + (size_t)i * n2 * W;
  const double* rk = aug + (size_t)k * n2 * W;
  const double fr = row[(size_t)k * W], fi = CPLX ? row[(size_t)k * W + 1] : 0.0;
  __syncthreads();  // every thread holds the factor before column k is cleared
  for (int c = k + 1 + threadIdx.x; c < n2; c += blockDim.x) {  // columns < k of the left half are already e_j
    if (CPLX) {
      const double rr = rk[c * W], ri = rk[c * W + 1];
      row[c * W] -= fr * rr - fi * ri;
      row[c * W + 1] -= fr * ri + fi * rr;
    } else {
      row[c] -= fr * rk[c];
    }
  }
  if (threadIdx.x == 0) {
    row[(size_t)k * W] = 0.0;
    if (CPLX) row[(size_t)k * W + 1] = 0.0;
  }
}

// ---- a4: state.py:109-110  p[b][m] = clip(d * sum_k A[m][k] bloch[b][k], 0, 1) ---------------
// 256 rows m per block, TB trials per block pass; the Bloch tile is staged in LDS and each A
// element (read once, coalesced over m from the [D][M] layout) feeds TB accumulators.
template <int TB>
__global__ void __launch_bounds__(256) k_born(const double* __restrict__ AT, int M, int D, int dscale,
                                              const double* __restrict__ bloch, int B, double* __restrict__ p) {
  extern __shared__ double sbl[];  // [TB][D]
  const int m = blockIdx.x * 256 + threadIdx.x;
  for (int b0 = blockIdx.y * TB; b0 < B; b0 += gridDim.y * TB) {
    __syncthreads();
    for (int e = threadIdx.x; e < TB * D; e += 256) {
      const int t = e / D, k = e % D;
      sbl[e] = (b0 + t < B) ? bloch[(size_t)(b0 + t) * D + k] : 0.0;
    }
    __syncthreads();
    if (m < M) {
      double acc[TB];
#pragma unroll
      for (int t = 0; t < TB; ++t) acc[t] = 0.0;
      for (int k = 0; k < D; ++k) {
        const double a = AT[(size_t)k * M + m];
#pragma unroll
        for (int t = 0; t < TB; ++t) acc[t] += a * sbl[t * D + k];
      }
#pragma unroll
      for (int t = 0; t < TB; ++t)
        if (b0 + t < B) {
          double v = acc[t] * dscale;
          v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
          p[(size_t)(b0 + t) * M + m] = v;
        }
    }
  }
}

// ---- a4 on the FP64 matrix cores: P[B x M] = d * Bloch[B x D] * A^T[D x M], clipped to [0, 1] ------------
// Batched over states the Born rule is a GEMM with a small fixed right-hand side.  A^T (padded to 16-column
// tiles) is staged once per workgroup in LDS (112 KB at n = 3: one workgroup of 8 waves per CU); a wavefront
// takes 16 states at a time, holds their Bloch rows as the MFMA A-operand (D / 4 registers), and walks the
// M / 16 column tiles with v_mfma_f64_16x16x4_f64, one ds_read_b64 per MFMA.  What is left is the stream:
// 8 (D + M) bytes per state.  (Operand layout as in k_gemm above.)
//
// 16 wavefronts per workgroup (4 per SIMD, 71 VGPRs): the other waves of a SIMD cover the operand latencies
// (an 8-wave version with software prefetch and double-buffered operand tiles measured 59 us where this
// one takes 51).  LDS pitch Mp = 16 mod 32 doubles: the four rows of one B-operand read (128 bytes each)
// alternate between the two halves of the 64 banks.
template <int DD>
__global__ void __launch_bounds__(1024) k_born_mfma(const double* __restrict__ AT, int M, int Mp, int dscale,
                                                      const double* __restrict__ bloch, int B, double* __restrict__ p) {
  extern __shared__ double s_at[];
  for (int e = threadIdx.x; e < DD * Mp; e += blockDim.x) {
    const int k = e / Mp, m = e % Mp;
    s_at[e] = m < M ? AT[(size_t)k * M + m] : 0.0;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int r16 = lane & 15, kq = lane >> 4;
  const int ngroups = (B + 15) / 16, ntiles = (M + 15) / 16;
  for (int g = blockIdx.x * nwave + wave; g < ngroups; g += gridDim.x * nwave) {
    const int row = g * 16 + r16;
    double a[DD / 4];
#pragma unroll
    for (int sidx = 0; sidx < DD / 4; ++sidx) a[sidx] = row < B ? bloch[(size_t)row * DD + 4 * sidx + kq] : 0.0;
    for (int ct = 0; ct < ntiles; ++ct) {
      const int c0 = ct * 16;
      double bt[DD / 4];
#pragma unroll
      for (int sidx = 0; sidx < DD / 4; ++sidx) bt[sidx] = s_at[(4 * sidx + kq) * Mp + c0 + r16];
      v4f64 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int sidx = 0; sidx < DD / 4; ++sidx) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[sidx], bt[sidx], acc, 0, 0, 0);
      const int col = c0 + r16;
      if (col < M) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int orow = g * 16 + kq + 4 * r;
          if (orow < B) {
            double v = acc[r] * dscale;
            v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            p[(size_t)orow * M + col] = v;
          }
        }
      }
    }
  }
}

// ---- a3 (any n): qobj.py:126-135 / :109-118, one thread per output element --------------------
__global__ void k_bloch_from_mat(int nq, const double* __restrict__ mat, int B, double* __restrict__ bloch) {
  const int d = 1 << nq;
  const size_t D = (size_t)d * d;
  const size_t total = (size_t)B * D;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(e % D);
    const double* m = mat + (e / D) * D * 2;
    int xm = 0, zm = 0, ny = 0;
    for (int b = 0; b < nq; ++b) {
      const int dig = (k >> (2 * b)) & 3;
      if (dig == 1 || dig == 2) xm |= 1 << b;
      if (dig == 2 || dig == 3) zm |= 1 << b;
      if (dig == 2) ++ny;
    }
    double sr = 0.0, si = 0.0;
    for (int r = 0; r < d; ++r) {
      const double* el = m + ((size_t)r * d + (r ^ xm)) * 2;
      const double sg = (__popc(r & zm) & 1) ? -1.0 : 1.0;
      sr += sg * el[0];
      si -= sg * el[1];
    }
    double v;
    switch (ny & 3) {
      case 0: v = sr; break;
      case 1: v = si; break;
      case 2: v = -sr; break;
      default: v = -si; break;
    }
    bloch[e] = v / d;
  }
}

__global__ void k_mat_from_bloch(int nq, const double* __restrict__ bloch, int B, double* __restrict__ mat) {
  const int d = 1 << nq;
  const size_t D = (size_t)d * d;
  const size_t total = (size_t)B * D;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    const int el = (int)(e % D);
    const int i = el / d, j = el % d;
    const double* v = bloch + (e / D) * D;
    const int x = i ^ j;
    double sr = 0.0, si = 0.0;
    for (int z = 0; z < d; ++z) {
      int k = 0;
      for (int b = 0; b < nq; ++b) {
        const int xb = (x >> b) & 1, zb = (z >> b) & 1;
        const int dig = xb ? (zb ? 2 : 1) : (zb ? 3 : 0);
        k |= dig << (2 * b);
      }
      const double sg = (__popc(i & z) & 1) ? -1.0 : 1.0;
      const double val = sg * v[k];
      switch (__popc(x & z) & 3) {
        case 0: sr += val; break;
        case 1: si -= val; break;
        case 2: sr -= val; break;
        default: si += val; break;
      }
    }
    mat[2 * e] = sr;
    mat[2 * e + 1] = si;
  }
}

// ---- a16: geometry.py:16-20, one wavefront per trial -----------------------------------------
__global__ void __launch_bounds__(64) k_hs_dist(int d, const double* __restrict__ rho, const double* __restrict__ centre,
                                                int B, double* __restrict__ dist) {
  const int b = blockIdx.x;
  if (b >= B) return;
  const double* r = rho + (size_t)b * d * d * 2;
  double sr = 0.0, si = 0.0;
  for (int e = threadIdx.x; e < d * d; e += 64) {
    const int i = e / d, j = e % d;
    const int et = j * d + i;
    const cd dl{r[2 * e] - centre[2 * e], r[2 * e + 1] - centre[2 * e + 1]};
    const cd dt{r[2 * et] - centre[2 * et], r[2 * et + 1] - centre[2 * et + 1]};
    const cd t = hs_term(dl, dt);  // (Delta Delta)_ii summed = sum_ij Delta_ij Delta_ji
    sr += t.re;
    si += t.im;
  }
  // the reduction tree of the estimators' fused distance (Small::hs_to_centre): for d <= 8 -- one element per lane --
  // the two-pass and the one-pass distance are the same bits
  sr = gsum<64>(sr);
  si = gsum<64>(si);
  if (threadIdx.x == 0) {
    const double v = sqrt(hypot(sr, si)) / sqrt(2.0);
    dist[b] = v < 1e-15 ? 0.0 : v;
  }
}

// ---- a16: interval.py:610  dist.sort() for small n (<= 8192: every bootstrap the reference's defaults produce) ----
// One workgroup, bitonic network in LDS on order-preserving 64-bit keys (sign-flipped IEEE bits: the order of a radix
// sort, NaN last like np.sort).  A device radix sort is five launches for any n; this is one, ~10 us at n = 2048.
__global__ void __launch_bounds__(1024) k_sort_small(double* __restrict__ x, int n, int npow2) {
  extern __shared__ unsigned long long keys[];
  for (int e = threadIdx.x; e < npow2; e += blockDim.x) {
    unsigned long long k = ~0ull;  // padding sorts behind everything
    if (e < n) {
      const unsigned long long b = (unsigned long long)__double_as_longlong(x[e]);
      k = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    }
    keys[e] = k;
  }
  __syncthreads();
  for (int size = 2; size <= npow2; size <<= 1) {
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
      for (int t = threadIdx.x; t < npow2 / 2; t += blockDim.x) {
        const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
        const bool up = (lo & size) == 0;
        const unsigned long long a = keys[lo], b = keys[hi];
        if ((a > b) == up) {
          keys[lo] = b;
          keys[hi] = a;
        }
      }
      __syncthreads();
    }
  }
  for (int e = threadIdx.x; e < n; e += blockDim.x) {
    const unsigned long long k = keys[e];
    const unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    x[e] = __longlong_as_double((long long)b);
  }
}

// ---- a16: interval.py:610-612  interp1d(linspace(0, 1, n), sorted_dist)(conf_levels) -------------
// scipy's linear interp1d: hi = searchsorted(x, q) clipped to [1, n-1], lo = hi - 1,
// y = (y_hi - y_lo) / (x_hi - x_lo) * (q - x_lo) + y_lo, with x_i = i * (1 / (n - 1)) and x_(n-1) = 1 exactly
// (np.linspace).  q outside [0, 1] gives NaN (interp1d raises; the Python layer checks first).
// scipy's interp1d(kind='linear') on real 1-D data IS numpy.interp (interp1d._call_linear_np; also in the scipy 1.9.3 the
// reference pins), so these are numpy.interp's semantics (numpy/_core/src/multiarray/compiled_base.c, arr_interp), on the
// grid x_i = i / (n - 1) of np.linspace(0, 1, n) (x_(n-1) = 1 exactly): the cell is j with x_j <= x < x_(j+1); a query ON a
// grid point (or at the right end) returns y_j itself -- rounds 2's kernel took the cell to the LEFT of a grid point and
// evaluated the line at its right end, one ulp off y_j in ~1 of 3 such cases (found by the property sweep of
// tests/test_sharded_quantiles.py: levels that are grid points, e.g. 0.5 with n = 2001) -- otherwise
// slope = (y_(j+1) - y_j) / (x_(j+1) - x_j), y = slope (x - x_j) + y_j, with numpy's retry from the other end if that is NaN.
struct InterpCell {
  long long j;   // cell: grid(j) <= x < grid(j + 1), or n - 1 for x = 1
  double xj, xj1;
  bool exact;    // x == grid(j) (or j == n - 1): the result is y_j
};
__device__ __forceinline__ InterpCell interp_cell(long long n, double x) {  // n >= 2, 0 <= x <= 1
  const double step = 1.0 / (double)(n - 1);
  auto grid = [&](long long i) { return i == n - 1 ? 1.0 : (double)i * step; };
  long long j = (long long)(x * (double)(n - 1));  // close to the answer; fix up against the actual grid values
  if (j > n - 1) j = n - 1;
  while (j > 0 && grid(j) > x) --j;
  while (j < n - 1 && grid(j + 1) <= x) ++j;
  const double xj = grid(j);
  return InterpCell{j, xj, j < n - 1 ? grid(j + 1) : 1.0, j == n - 1 || xj == x};
}
// No contraction: an fma in place of the last multiply-add changes the last bit, and the bootstrap quantiles are compared
// with interp1d bit for bit.
__device__ __forceinline__ double interp_value(const InterpCell& c, double x, double yj, double yj1) {
#pragma clang fp contract(off)
  if (c.exact) return yj;
  const double slope = (yj1 - yj) / (c.xj1 - c.xj);
  double res = slope * (x - c.xj);
  res = res + yj;
  if (res != res) {  // "If we get nan in one direction, try the other"
    res = slope * (x - c.xj1);
    res = res + yj1;
    if (res != res && yj == yj1) res = yj;
  }
  return res;
}

__global__ void k_interp_sorted(const double* __restrict__ y, long long n, const double* __restrict__ q, int nq,
                                double* __restrict__ out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nq) return;
  const double x = q[t];
  if (!(x >= 0.0 && x <= 1.0) || n < 1) {
    out[t] = __builtin_nan("");
    return;
  }
  if (n == 1) {
    out[t] = y[0];
    return;
  }
  const InterpCell c = interp_cell(n, x);
  out[t] = interp_value(c, x, y[c.j], y[c.j < n - 1 ? c.j + 1 : c.j]);
}

// ---- a16 over several ranks: interval.py:610-612 when the sorted sample is spread over N ranks -------------------------
// Every rank holds a SORTED shard of the n_total bootstrap distances.  interp1d at a confidence level needs the two order
// statistics k0 = cell(q).j and k1 = min(k0 + 1, n - 1) of the union, not the union: they are found exactly in two small exchanges
// (quantpy_amd/distributed.py) instead of an all-gather of everything and a full sort on every rank --
//   1. every rank publishes P splitters, shard[j * s] (k_select_splitters);  all-gather [N][P]
//   2. with cnt_r(v) = number of rank r's splitters <= v, the number of sample values <= v lies between
//      low(v) = sum_r ((cnt_r - 1) s + 1 | 0) and up(v) = sum_r min(n_r, cnt_r s).  lo = the largest splitter with
//      up <= k0 (the k0-th value is > lo), hi = the smallest with low >= k1 + 1 (the k1-th value is <= hi)
//      (k_select_bracket; the same on every rank)
//   3. every rank publishes how many of its values are <= lo and its values in (lo, hi] -- at most (2 N + 3) s of them
//      (k_select_window);  all-gather [N][L][2 + W]
//   4. the (k - sum below)-th smallest of the windows' union is the k-th of the sample (k_select_finish), then scipy's
//      interpolation formula on the two values.
// All comparisons are on the order-preserving 64-bit keys of the radix sort (NaN last, as np.sort has it).
__device__ __forceinline__ unsigned long long sort_key(double v) {
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
// number of keys <= k (UPPER = true) or < k (UPPER = false) in the sorted doubles a[0..n)
template <bool UPPER, class Ptr>
__device__ __forceinline__ long long count_sorted(Ptr a, long long n, unsigned long long k) {
  long long lo = 0, hi = n;
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;
    const unsigned long long m = sort_key(a[mid]);
    if (UPPER ? (m <= k) : (m < k)) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// splitters[j] = sorted[j * stride] for j * stride < n, +inf-most key (NaN pattern ~0) behind; also resets the bracket
// accumulators of the next step (lo_key[L] = 0 "none", hi_key[L] = ~0 "none") so that no separate memset is needed
__global__ void k_select_splitters(const double* __restrict__ sorted, long long n, long long stride, int P,
                                   double* __restrict__ splitters) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= P) return;
  const long long at = (long long)j * stride;
  splitters[j] = at < n ? sorted[at] : __longlong_as_double(0x7fffffffffffffffLL);  // key ~0: behind everything
}

__global__ void k_select_init(unsigned long long* __restrict__ lo_key, unsigned long long* __restrict__ hi_key, int L) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < L) {
    lo_key[t] = 0ull;
    hi_key[t] = ~0ull;
  }
}

// spl[N][P] (row r = rank r's splitters, valid: ceil(n_r / stride)), sizes[N], conf levels q[L] -> lo_key[L], hi_key[L]
__global__ void __launch_bounds__(256) k_select_bracket(const double* __restrict__ spl, int N, int P,
                                                        const long long* __restrict__ sizes, long long stride,
                                                        long long n_total, const double* __restrict__ q, int L,
                                                        unsigned long long* __restrict__ lo_key,
                                                        unsigned long long* __restrict__ hi_key) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long long)N * P) return;
  const int r0 = (int)(idx / P), j0 = (int)(idx % P);
  if ((long long)j0 * stride >= sizes[r0]) return;  // padding
  const unsigned long long kv = sort_key(spl[idx]);
  long long up = 0, low = 0;
  for (int r = 0; r < N; ++r) {
    const long long nr = sizes[r], pr = (nr + stride - 1) / stride;
    const long long cnt = count_sorted<true>(spl + (size_t)r * P, pr, kv);
    const long long u = cnt * stride;
    up += u < nr ? u : nr;
    low += cnt > 0 ? (cnt - 1) * stride + 1 : 0;
  }
  for (int l = 0; l < L; ++l) {
    const double x = q[l];
    if (!(x >= 0.0 && x <= 1.0) || n_total < 2) continue;
    const long long k0 = interp_cell(n_total, x).j, k1 = k0 < n_total - 1 ? k0 + 1 : k0;
    if (up <= k0) atomicMax(lo_key + l, kv);
    if (low >= k1 + 1) atomicMin(hi_key + l, kv);
  }
}

// One workgroup per level: win[l][0] = number of this rank's values <= lo, win[l][1] = w = number in (lo, hi],
// win[l][2 .. 2 + min(w, W)) = those values.  w > W (heavy ties; otherwise excluded by the bound of step 3) is seen by
// k_select_finish on every rank, which raises its overflow flag.
__global__ void __launch_bounds__(256) k_select_window(const double* __restrict__ sorted, long long n,
                                                       const unsigned long long* __restrict__ lo_key,
                                                       const unsigned long long* __restrict__ hi_key, int W,
                                                       double* __restrict__ win) {
  const int l = blockIdx.x;
  const unsigned long long lk = lo_key[l], hk = hi_key[l];
  const long long below = lk == 0ull ? 0 : count_sorted<true>(sorted, n, lk);
  const long long upto = hk == ~0ull ? n : count_sorted<true>(sorted, n, hk);
  const long long w = upto - below;
  double* out = win + (size_t)l * (2 + W);
  if (threadIdx.x == 0) {
    out[0] = (double)below;
    out[1] = (double)w;
  }
  for (long long e = threadIdx.x; e < w && e < W; e += blockDim.x) out[2 + e] = sorted[below + e];
}

// One workgroup per level: gathered windows gw[N][L][2 + W] -> out[l] = interp1d(linspace(0, 1, n_total), sample)(q[l]).
// The valid window entries of all ranks are compacted into LDS (at most `cap` of them: the host sizes the allocation
// from the bound (3 N + 2) s); candidate c is the t-th smallest of the union iff #(< c) <= t < #(<= c), each count a sum
// of N binary searches over the ranks' (sorted) windows.
__global__ void __launch_bounds__(1024) k_select_finish(const double* __restrict__ gw, int N, int L, int W, int cap,
                                                        long long n_total, const double* __restrict__ q,
                                                        double* __restrict__ out, int* __restrict__ flag) {
  extern __shared__ double cand[];  // [cap] values, then N + 1 offsets (as doubles' worth of ints behind)
  int* off = reinterpret_cast<int*>(cand + cap);
  __shared__ double picked[2];
  __shared__ long long s_below;
  const int l = blockIdx.x;
  const double x = q[l];
  if (!(x >= 0.0 && x <= 1.0) || n_total < 1) {
    if (threadIdx.x == 0) out[l] = __builtin_nan("");
    return;
  }
  if (threadIdx.x == 0) {
    long long below = 0;
    int tot = 0;
    for (int r = 0; r < N; ++r) {
      const double* w = gw + ((size_t)r * L + l) * (2 + W);
      below += (long long)w[0];
      off[r] = tot;
      if (w[1] > (double)W) flag[0] = 1;  // a clipped window: the caller takes the merge path
      tot += w[1] > (double)W ? W : (int)w[1];
    }
    off[N] = tot;
    s_below = below;
    if (tot > cap) flag[0] = 2;
    picked[0] = picked[1] = __builtin_nan("");
  }
  __syncthreads();
  const int tot = off[N] > cap ? cap : off[N];
  for (int r = 0; r < N; ++r) {
    const double* w = gw + ((size_t)r * L + l) * (2 + W) + 2;
    const int o = off[r], cnt = off[r + 1] - o;
    for (int e = threadIdx.x; e < cnt && o + e < cap; e += blockDim.x) cand[o + e] = w[e];
  }
  __syncthreads();
  long long k0 = 0, k1 = 0;
  InterpCell cell{0, 0.0, 1.0, true};
  if (n_total >= 2) {
    cell = interp_cell(n_total, x);
    k0 = cell.j;
    k1 = k0 < n_total - 1 ? k0 + 1 : k0;
  }
  const long long t0 = k0 - s_below, t1 = k1 - s_below;
  for (int e = threadIdx.x; e < tot; e += blockDim.x) {
    const unsigned long long kv = sort_key(cand[e]);
    long long lt = 0, le = 0;
    for (int r = 0; r < N; ++r) {
      const int o = off[r], cnt = (off[r + 1] > cap ? cap : off[r + 1]) - o;
      if (cnt <= 0) continue;
      lt += count_sorted<false>(cand + o, cnt, kv);
      le += count_sorted<true>(cand + o, cnt, kv);
    }
    if (lt <= t0 && t0 < le) picked[0] = cand[e];  // (ties: equal values, any writer)
    if (lt <= t1 && t1 < le) picked[1] = cand[e];
  }
  __syncthreads();
  if (threadIdx.x == 0) out[l] = interp_value(cell, x, picked[0], picked[1]);  // (n_total = 1: the cell is exact)
}

// ---- f2: stats.py:21-47 l2_first_moment / l2_second_moment over a batch of trials (MomentInterval, interval.py:59-110) --
// Mean and variance of ||P (f - p)||^2 for multinomial frequencies f[S][K] (N shots per setting) with the weights
// W[(a,i),(b,j)] = sum_d P[d][(a,i)] P[d][(b,j)] (interval.py:88: einsum('aij,akl->ijkl')).  With
//   U[a][(b,j)] = sum_i f_ai W[(a,i),(b,j)],  Q_ab = sum_j U[a][(b,j)] f_bj,  t = sum_ai W[(a,i),(a,i)] f_ai
// the reference's fourteen einsums collect to (W symmetric)
//   E  = (t - tr Q) / N
//   E2 = ((tr Q - t)^2 + 2 sum_ab Q_ab^2 - 4 sum_a sum_bj U[a][(b,j)]^2 f_bj + 2 f^T (W o W) f) / N^2,   Var = E2 - E^2.
// One workgroup takes T trials and walks the settings a: thread c owns columns c, c + 256, ... of W (so a wavefront reads
// 512 contiguous bytes of each row, and every row once per workgroup: W is read B / T times from L2 in all), keeps
// U[a][c] for its columns in registers while the K rows of setting a go by, squares / multiplies on the spot, and leaves
// U[a][c] f_c in LDS for the segmented sums Q_ab (K consecutive columns) that threads b < S square and add up.
template <int T, int NC>
__global__ void __launch_bounds__(256) k_moment_batch(const int64_t* __restrict__ counts, int B, int S, int K,
                                                      const double* __restrict__ ns, const double* __restrict__ W,
                                                      double n_trials, double* __restrict__ mean,
                                                      double* __restrict__ var) {
  extern __shared__ double sm[];
  const int M = S * K, tid = threadIdx.x;
  double* f = sm;               // [T][M]
  double* prod = sm + T * M;    // [T][M]
  __shared__ double red[4][5 * T];
  const int b0 = blockIdx.x * T;
  for (int e = tid; e < T * M; e += 256) {
    const int t = e / M, r = e % M, b = b0 + t;
    f[e] = b < B ? (double)counts[(size_t)b * M + r] / ns[r / K] : 0.0;
  }
  __syncthreads();
  double uuf[T], wwf[T], tdiag[T], q2[T], trq[T], ww[NC][T];
#pragma unroll
  for (int t = 0; t < T; ++t) uuf[t] = wwf[t] = tdiag[t] = q2[t] = trq[t] = 0.0;
#pragma unroll
  for (int n = 0; n < NC; ++n)
#pragma unroll
    for (int t = 0; t < T; ++t) ww[n][t] = 0.0;
  for (int a = 0; a < S; ++a) {
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      const int c = tid + 256 * n;
      if (c < M) {
        double u[T];
#pragma unroll
        for (int t = 0; t < T; ++t) u[t] = 0.0;
        for (int i = 0; i < K; ++i) {
          const int r = a * K + i;
          const double w = W[(size_t)r * M + c], w2 = w * w;
#pragma unroll
          for (int t = 0; t < T; ++t) {
            const double fr = f[t * M + r];
            u[t] = fma(fr, w, u[t]);
            ww[n][t] = fma(fr, w2, ww[n][t]);
          }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
          const double fc = f[t * M + c], uf = u[t] * fc;
          uuf[t] = fma(u[t], uf, uuf[t]);
          prod[t * M + c] = uf;
        }
      }
    }
    __syncthreads();
    for (int e = tid; e < T * S; e += 256) {  // Q_ab for every b of every trial
      const int t = e / S, b = e % S;
      const double* p = prod + t * M + b * K;
      double q = 0.0;
      for (int j = 0; j < K; ++j) q += p[j];
#pragma unroll
      for (int tt = 0; tt < T; ++tt)
        if (tt == t) {
          q2[tt] = fma(q, q, q2[tt]);
          if (b == a) trq[tt] += q;
        }
    }
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int c = tid + 256 * n;
    if (c < M) {
      const double wd = W[(size_t)c * M + c];
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const double fc = f[t * M + c];
        wwf[t] = fma(ww[n][t], fc, wwf[t]);
        tdiag[t] = fma(wd, fc, tdiag[t]);
      }
    }
  }
  // five sums per trial over the workgroup
#pragma unroll
  for (int t = 0; t < T; ++t) {
    double v[5] = {uuf[t], wwf[t], tdiag[t], q2[t], trq[t]};
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      v[k] = gsum<64>(v[k]);
      if ((tid & 63) == 0) red[tid >> 6][5 * t + k] = v[k];
    }
  }
  __syncthreads();
  if (tid < T && b0 + tid < B) {
    double v[5];
    for (int k = 0; k < 5; ++k) v[k] = (red[0][5 * tid + k] + red[1][5 * tid + k]) + (red[2][5 * tid + k] + red[3][5 * tid + k]);
    const double s_uuf = v[0], s_wwf = v[1], t_d = v[2], s_q2 = v[3], tr_q = v[4];
    const double first = (t_d - tr_q) / n_trials;
    const double second = ((tr_q - t_d) * (tr_q - t_d) + 2.0 * s_q2 - 4.0 * s_uuf + 2.0 * s_wwf) / (n_trials * n_trials);
    mean[b0 + tid] = first;
    var[b0 + tid] = second - first * first;
  }
}

// ---- merge of two adjacent sorted runs (gather of sorted shards -> the sorted sample; np.sort's order, NaN last) ------
// in[0..na) and in[na..na+nb) sorted -> out[0..na+nb): every thread owns TILE consecutive outputs, finds its start on
// the merge path by a binary search over the diagonal, then merges sequentially.  Stable (ties: first run first).
template <int TILE>
__global__ void __launch_bounds__(256) k_merge_runs(const double* __restrict__ a, long long na,
                                                    const double* __restrict__ b, long long nb,
                                                    double* __restrict__ out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long diag = t * TILE, total = na + nb;
  if (diag >= total) return;
  long long lo = diag > nb ? diag - nb : 0, hi = diag < na ? diag : na;  // number taken from a
  while (lo < hi) {
    const long long mid = (lo + hi) >> 1;  // take mid from a, diag - mid from b: valid iff a[mid] > b[diag - mid - 1] fails
    if (sort_key(a[mid]) <= sort_key(b[diag - mid - 1])) lo = mid + 1;
    else hi = mid;
  }
  long long ia = lo, ib = diag - lo;
#pragma unroll
  for (int e = 0; e < TILE; ++e) {
    if (diag + e >= total) break;
    const bool take_a = ib >= nb || (ia < na && sort_key(a[ia]) <= sort_key(b[ib]));
    out[diag + e] = take_a ? a[ia++] : b[ib++];
  }
}

}  // namespace qt
