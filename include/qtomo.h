/* qtomo.h -- C ABI of libqtomo.so, the MI355X (gfx950) tomography-reconstruction engine.
 *
 * The reference (nordmtr/quantpy) is pure Python and has no FFI; this boundary sits directly
 * beneath the reference methods cited on each entry point (paths are into the reference tree,
 * SURVEY.md section 8a/8b).  INTEGRATION.md shows the ctypes binding a quantpy maintainer would
 * add; quantpy_amd/_capi.py is that binding for this repository's drop-in classes.
 *
 * Conventions
 *   - n qubits, d = 2^n, D = 4^n; a POVM is a tensor A[S][K][D] of Bloch rows (M = S*K rows).
 *   - all arrays are caller-owned and C-contiguous; complex = interleaved double[2] (re, im).
 *   - pointers are HOST pointers unless `flags & QT_DEVICE_PTR`, in which case every array
 *     argument of that call is a device pointer valid on the handle's device and the call is
 *     asynchronous on the handle's stream (qt_sync to wait).
 *   - every function returns 0 on success and a negative qt_status on argument / runtime errors
 *     (qt_last_error() gives the text).  Batched estimators additionally fill a per-trial
 *     `int32 status[B]` with qt_trial_status and, for host-pointer calls, return the number of
 *     trials whose status is non-zero.
 *   - a handle is bound to one device and one stream and is not thread-safe; use one handle per
 *     host thread.  No exception crosses the ABI.  There is no CPU fallback: without a usable
 *     HIP device qt_create fails.
 */
#ifndef QTOMO_H
#define QTOMO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qt_handle qt_handle_t;

enum qt_status {
  QT_OK = 0,
  QT_ERR_ARG = -1,      /* bad argument (null pointer, size, n_qubits out of range) */
  QT_ERR_STATE = -2,    /* call order: e.g. estimator before qt_set_povm */
  QT_ERR_HIP = -3,      /* HIP runtime error */
  QT_ERR_SINGULAR = -4, /* A^T A is singular: the POVM is not informationally complete */
  QT_ERR_UNSUPPORTED = -5
};

enum qt_trial_status {
  QT_TRIAL_OK = 0,
  QT_TRIAL_NOT_PD = 1,      /* Cholesky pivot <= 0: scipy.linalg.cholesky raises LinAlgError   */
  QT_TRIAL_LINESEARCH = 2,  /* both line searches failed: scipy BFGS warnflag 2                 */
  QT_TRIAL_MAXITER = 3,     /* iteration cap reached: scipy BFGS warnflag 1                     */
  QT_TRIAL_NAN = 4,         /* NaN in value, gradient or parameters: scipy BFGS warnflag 3      */
  QT_TRIAL_SHOTS = 5        /* the trial's per-setting totals are not proportional to the Ns registered with
                               qt_set_povm: the reference would weight this trial differently
                               (state.py:138-141, 194-197); the returned estimate is not meaningful  */
};

enum qt_flags { QT_HOST_PTR = 0, QT_DEVICE_PTR = 1 };
enum qt_init { QT_INIT_LIN = 0, QT_INIT_MIXED = 1 };

/* ---- library / handle ------------------------------------------------------------------ */
int qt_version(void);
const char* qt_last_error(void);
/* number of HIP devices visible, or a negative qt_status */
int qt_device_count(void);
/* n_qubits in 1..5.  State estimators: any POVM at every n (n = 4, 5: a product POVM registered with
 * qt_set_povm_product contracts qubit by qubit; a plain tensor takes the streaming dense path).  Process
 * tomography: n <= 3 (n = 3 through the Kronecker factors of the design matrix, see qt_process_setup). */
qt_handle_t* qt_create(int device, int n_qubits);
void qt_destroy(qt_handle_t* h);
int qt_sync(qt_handle_t* h);
/* Run the handle's work on an existing hipStream_t (e.g. torch's current stream); NULL = a private
 * non-blocking stream (the default after qt_create); QT_STREAM_LEGACY = the legacy default ("null")
 * stream, which is what a framework means by stream 0.
 * ORDERING CONTRACT of device-pointer calls (flags & QT_DEVICE_PTR): they are enqueued on the handle's
 * stream and return at once.  Inputs must have been produced on that stream, or by work the caller has
 * already ordered before it (event / synchronize); outputs may be consumed on that stream, after
 * qt_sync, or after the caller's own event on it.  A caller whose producers run on another stream
 * (torch's current stream, say) either binds the handle to that stream with qt_set_stream or orders
 * the two streams itself.  Every call leaves the calling thread's current HIP device unchanged. */
#define QT_STREAM_LEGACY ((void*)1)
int qt_set_stream(qt_handle_t* h, void* hip_stream);
/* Per-handle switches.  QT_OPT_SHOTS_CHECK (default 1): compare every trial's per-setting totals with the Ns of
 * qt_set_povm (QT_TRIAL_SHOTS).  QT_OPT_MLE_FUSED_MAX_WAVES (default 1024): batches of up to this many trial-wavefronts
 * (n <= 3) run the MLE as one launch with the BFGS inverse Hessian in registers; larger ones as a start launch plus a
 * BFGS launch in two-loop form (0 = always the latter).  Both forms compute scipy's iterates. */
enum qt_option { QT_OPT_SHOTS_CHECK = 1, QT_OPT_MLE_FUSED_MAX_WAVES = 2 };
int qt_set_option(qt_handle_t* h, int option, double value);
/* hipEvent timers on the handle's stream: begin, ..., end -> elapsed milliseconds */
int qt_timer_begin(qt_handle_t* h);
int qt_timer_end(qt_handle_t* h, double* elapsed_ms);
/* the two halves of qt_timer_end: record the end event (asynchronous) / wait for it and read the interval */
int qt_timer_stop(qt_handle_t* h);
int qt_timer_elapsed(qt_handle_t* h, double* elapsed_ms);

/* ---- a1: quantpy/routines.py:14-19 generate_pauli ---------------------------------------- */
/* out[D][d][d][2]: P_k = P_k1 (x) ... (x) P_kn, k = sum_j k_j 4^(n-1-j) */
int qt_pauli_basis(qt_handle_t* h, double* out, int flags);

/* ---- a2: quantpy/measurements.py:88-93 generate_measurement_matrix (array / string path) - */
/* povm1[S1][K1][4] one-qubit table -> out[S1^n][K1^n][D], all three axes Kronecker'd */
int qt_povm_kron(qt_handle_t* h, const double* povm1, int S1, int K1, double* out, int flags);

/* ---- a5 + cache: quantpy/routines.py:69-71 _left_inv; state.py:194-197, 222-225 ----------- */
/* Stores A[S][K][D], the shot-weighted A' = A * Ns[s] / sum(Ns) reshaped (M, D), and the left
 * inverse inv(A'^T A') A'^T (plain transpose).  Ns[S] = shots per setting (as float64, like the
 * reference's n_measurements).  Every later estimator call on this handle uses this POVM and
 * assumes each trial's per-setting totals equal Ns. */
int qt_set_povm(qt_handle_t* h, const double* A, int S, int K, const double* Ns, int flags);
/* The same for a POVM given as a one-qubit table povm1[S1][K1][4] to be tensored n times
 * (measurements.py:88-93: every built-in POVM and every array whose last axis is 4).  Builds the
 * full tensor on the device and, in addition, keeps the factorised form: the estimators then
 * contract qubit by qubit instead of over the dense M x D operand.  Ns[S1^n]. */
int qt_set_povm_product(qt_handle_t* h, const double* povm1, int S1, int K1, const double* Ns, int flags);
/* out[D][M]: the cached left inverse (for inspection / tests) */
int qt_get_left_inverse(qt_handle_t* h, double* out, int flags);

/* the same left inverse for an arbitrary rows x cols matrix, real (is_complex = 0) or complex
 * (interleaved; still the PLAIN transpose, as routines.py:71 writes it): out[cols][rows] */
int qt_left_inverse(qt_handle_t* h, const double* A, int rows, int cols, int is_complex, double* out, int flags);

/* ---- a4: quantpy/tomography/state.py:109-110 (probabilities only; sampling stays on host) -- */
/* p[B][S][K] = clip(d * sum_k A[s][k'][k] bloch[b][k], 0, 1) */
int qt_born_probs(qt_handle_t* h, const double* bloch, int B, double* p, int flags);

/* ---- a3: quantpy/qobj.py:109-135 + geometry.py:59-70 --------------------------------------- */
/* bloch[b][k] = Re Tr(P_k M_b^dagger) / d ;  M_b = sum_k bloch[b][k] P_k */
int qt_bloch_from_mat(qt_handle_t* h, const double* mat, int B, double* bloch, int flags);
int qt_mat_from_bloch(qt_handle_t* h, const double* bloch, int B, double* mat, int flags);

/* ---- a4 / a12 / a16 host side: state.py:109-114, the draws behind experiment() ----------------- */
/* rows x `np.random.multinomial(n[s], pvals[s])` with s = row % period -- NumPy's legacy sampler (MT19937 doubles,
 * conditional binomials by BTPE / inversion) restated bit for bit on the generator state the caller passes:
 * mt_key[624] and *mt_pos are those of `np.random.get_state()`, advanced in place by exactly the words NumPy
 * would have consumed, so `np.random.set_state()` afterwards leaves the global stream where the reference's
 * loop of Python calls leaves it.  pvals[period][K], out[rows][K].  Host memory only; needs no handle and no
 * GPU.  QT_ERR_ARG (with NumPy's message) when a pvals row fails RandomState.multinomial's own checks. */
int qt_legacy_multinomial(uint32_t* mt_key, int* mt_pos, long long rows, int period, const int64_t* n,
                          const double* pvals, int K, int64_t* out);

/* The same draws for callers that do NOT need the reference's stream (opt-in; state.py:109-114 only fixes the
 * distribution): row r is a multinomial(n[s], pvals[s]), s = (first_row + r) % period, made on the GPU by one thread
 * from its own Philox4x32-10 stream keyed by (seed, first_row + r) through the same conditional BTPE / inversion
 * binomials -- so the table depends on (seed, global row index) only, however the rows are split over calls or ranks,
 * and with QT_DEVICE_PTR the counts never leave HBM on their way to qt_lin/mle_batch.  n[period], pvals[period][K],
 * out[rows][K].  Host-pointer calls check pvals like RandomState.multinomial does (QT_ERR_ARG); device-pointer calls
 * clamp each conditional probability into [0, 1] (a NaN counts as 0). */
int qt_device_multinomial(qt_handle_t* h, uint64_t seed, uint64_t first_row, long long rows, int period,
                          const int64_t* n, const double* pvals, int K, int64_t* out, int flags);
/* The generator block behind it, on the host, for known-answer tests: out[4] = Philox4x32-10(ctr[4], key[2]). */
void qt_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out);

/* ---- a6 + a7: state.py:191-202 _point_estimate_lin, :267-273 _make_feasible ---------------- */
/* counts[B][S][K] int64 -> rho[B][d][d][2].  physical != 0: eigenvalues clipped at 1e-15 and
 * the trace renormalised.  bloch_out (nullable): the linear-inversion Bloch vector [B][D]. */
int qt_lin_batch(qt_handle_t* h, const int64_t* counts, int B, int physical, double* rho, double* bloch_out,
                 int32_t* status, int flags);

/* ---- a6/a7 + a16 in one pass: the body of the bootstrap loop, quantpy/tomography/interval.py:600-609 -------
 * qt_lin_batch plus dist[b] = hs_dst(estimate_b, centre) (geometry.py:5-20, as qt_hs_dist_batch computes it).
 * centre[d][d][2]; rho is NULLABLE here: a bootstrap needs one double per resample, not the matrices. */
int qt_lin_dist_batch(qt_handle_t* h, const int64_t* counts, int B, int physical, const double* centre, double* rho,
                      double* dist, int32_t* status, int flags);

/* ---- a8: quantpy/routines.py:84-101 Cholesky parametrisation ------------------------------- */
/* x[B][D] = [diag L | Re L_(i>j) | Im L_(i>j)] (strict lower in np.tril_indices order) of
 * rho[B][d][d][2] = L L^dagger ; and back (LLh = L L^dagger, not normalised) */
int qt_chol_param(qt_handle_t* h, const double* rho, int B, double* x, int32_t* status, int flags);
int qt_chol_unparam(qt_handle_t* h, const double* x, int B, double* LLh, int flags);

/* ---- a9: state.py:217-229 _nll (value) and its exact gradient ------------------------------ */
/* f[b] = -sum_m freq_m log(d (A' b(x_b))_m + 1e-10), freq = counts / sum(counts);
 * grad (nullable) [B][D] = df/dx (analytic; the reference differentiates numerically) */
int qt_nll_batch(qt_handle_t* h, const double* x, const int64_t* counts, int B, double* f, double* grad, int flags);

/* ---- a10: state.py:204-215 _point_estimate_mle_chol + scipy BFGS --------------------------- */
/* tol is scipy's gtol (inf-norm of the gradient); max_iter its maxiter.  Outputs (each nullable
 * except rho): nit[B] BFGS iterations, nfev[B] value+gradient evaluations (the reference's own
 * nfev, which also counts its finite-difference probes, equals nfev * (D + 1)), fun[B] final
 * objective, status[B]. */
int qt_mle_batch(qt_handle_t* h, const int64_t* counts, int B, int init, int max_iter, double tol, double* rho,
                 int32_t* nit, int32_t* nfev, double* fun, int32_t* status, int flags);

/* ---- a8-a10 + a16 in one pass (interval.py:600-609 with method='mle'): qt_mle_batch plus the Hilbert-Schmidt
 * distance of every estimate to centre[d][d][2]; rho NULLABLE (8 bytes written per resample instead of 16 d^2). */
int qt_mle_dist_batch(qt_handle_t* h, const int64_t* counts, int B, int init, int max_iter, double tol,
                      const double* centre, double* rho, double* dist, int32_t* nit, int32_t* nfev, double* fun,
                      int32_t* status, int flags);

/* ---- a16: quantpy/geometry.py:5-20 hs_dst --------------------------------------------------- */
/* dist[b] = sqrt(|Tr((rho_b - centre)^2)|) / sqrt(2), set to 0 below 1e-15 */
int qt_hs_dist_batch(qt_handle_t* h, const double* rho, const double* centre, int B, double* dist, int flags);
/* the same for dim x dim matrices whatever the handle's n_qubits (Choi matrices of an n-qubit channel: dim = 4^n) */
int qt_hs_dist_dim(qt_handle_t* h, int dim, const double* rho, const double* centre, int B, double* dist, int flags);

/* ---- a16: quantpy/tomography/interval.py:610-612 (and :683-685) -------------------------------- */
/* `dist.sort()`: ascending in-place sort of n float64 values (radix sort on the device; NaN last). */
int qt_sort_f64(qt_handle_t* h, double* x, long long n, int flags);
/* `interp1d(np.linspace(0, 1, n), sorted)(conf_levels)`: scipy's linear interp1d on real 1-D data is numpy.interp, and
 * these are its semantics on the grid x_i = i / (n - 1) -- cell x_j <= q < x_(j+1), a level ON a grid point returns
 * sorted[j] itself, otherwise slope * (q - x_j) + y_j operation by operation; out[n_levels].  A level outside [0, 1]
 * gives NaN (interp1d raises there). */
int qt_sorted_quantiles(qt_handle_t* h, const double* sorted, long long n, const double* conf_levels, int n_levels,
                        double* out, int flags);

/* ---- a16 when the distances are spread over N ranks (one process per GPU): interval.py:610-612 without gathering
 * the sample.  Every rank sorts ITS shard (qt_sort_f64); interp1d at a confidence level q needs the order statistics
 * k0 = floor-cell(q) and k0 + 1 of the union only.  Four small steps with two all-gathers of a few KB in between
 * (quantpy_amd/distributed.py does those with torch.distributed; the C ABI stays free of collectives):
 *   qt_select_splitters : splitters[j] = sorted[j * stride], j < P (padded behind everything)        -> all-gather [N][P]
 *   qt_select_bracket   : from all ranks' splitters and shard sizes: per level the tightest splitter pair
 *                         (lo, hi] that provably holds both order statistics (as radix-sort keys; 0 / ~0 = none)
 *   qt_select_window    : window[l] = { #values <= lo, w = #values in (lo, hi], the first min(w, W) of them } -> all-gather
 *   qt_select_finish    : out[l] = interp1d(linspace(0, 1, n_total), sorted union)(conf_levels[l]); *overflow != 0 when a
 *                         window was clipped (w > W, heavy ties) or the union exceeds the kernel's capacity: the caller
 *                         then gathers the sorted shards and merges them (qt_merge_sorted) instead.
 * W >= (2 N + 3) * stride covers every sample without massive ties.  Same results as np.sort + interp1d, bit for bit. */
int qt_select_splitters(qt_handle_t* h, const double* sorted, long long n, long long stride, int P, double* splitters,
                        int flags);
int qt_select_bracket(qt_handle_t* h, const double* splitters, int N, int P, const int64_t* sizes, long long stride,
                      long long n_total, const double* conf_levels, int L, uint64_t* lo_key, uint64_t* hi_key, int flags);
int qt_select_window(qt_handle_t* h, const double* sorted, long long n, const uint64_t* lo_key, const uint64_t* hi_key, int L,
                     int W, double* window, int flags);
int qt_select_finish(qt_handle_t* h, const double* windows, int N, int L, int W, long long n_total, const double* conf_levels,
                     double* out, int32_t* overflow, int flags);
/* R sorted runs stored back to back in runs (run_lengths[R]: always a HOST array) -> out: their merge in np.sort's
 * order (NaN last), by ceil(log2 R) merge-path passes.  out must not alias runs. */
int qt_merge_sorted(qt_handle_t* h, const double* runs, const int64_t* run_lengths, int R, double* out, int flags);

/* ---- f2: quantpy/stats.py:21-47 over a batch of trials (MomentInterval, interval.py:59-110) ----------------------
 * mean[b], var[b] of ||P (f_b - p)||^2 under multinomial noise, f_b[s][k] = counts[b][s][k] / ns[s] (interval.py:73,
 * :82), n_trials = the shots per setting (interval.py:89: n_measurements[0]), weights W = P^T P with
 * P = inv_matrix[rows][S*K] (the left inverse of the design matrix / dim, interval.py:75-87; W is formed on the matrix
 * cores).  For process tomography S = (input states) x (settings).  S * K <= 8192. */
int qt_moment_batch(qt_handle_t* h, const int64_t* counts, int B, int S, int K, const double* ns, const double* inv_matrix,
                    int rows, double n_trials, double* mean, double* var, int flags);

/* Metropolis-Hastings chains on the Cholesky parameters (mhmc.py:80-119 with `normalized_update`, used by
 * MHMCStateInterval, interval.py:735-750): C independent chains (the reference runs one), each on its
 * own counts[c][S][K]; x_init[C][D]; proposal increments deltas[C][T][D] and uniforms[C][T] drawn by the
 * caller (host RNG, reference order); chain[C][T][D] = state after every step, accepted[C][T].  n <= 3. */
int qt_mhmc_state(qt_handle_t* h, const int64_t* counts, int C, const double* x_init, const double* deltas,
                  const double* uniforms, int T, double step, double* chain, int32_t* accepted, int flags);

/* ---- a11-a15: quantpy/tomography/process.py -------------------------------------------------- */
/* Process tomography of an n-qubit channel (handle created with n_qubits = n; n <= 3: 'lifp', the projections and
 * what builds on them ('states', the bootstrap), 'pgdb' and the process chain all run for n <= 3; at n = 3 through the
 * Kronecker factors of the design matrix).
 * qt_process_setup: input states in_states[D][d][d][2] (process.py:79), the weighted POVM of
 * qt_set_povm (call it first) -> design matrix rows vec(rho_in (x) E_m^T) (process.py:203-208),
 * its left inverse (process.py:210), and the partial-trace operator (routines.py:47-50). */
int qt_process_setup(qt_handle_t* h, const double* in_states, int flags);
/* lifp_oper[D*M][D^2][2] (nullable) and its left inverse [D^2][D*M][2] (nullable) */
int qt_process_get_operators(qt_handle_t* h, double* lifp_oper, double* lifp_oper_inv, int flags);
/* n = 3 (and n = 2 beside the dense form): the design matrix Kronecker-factored, L = (V_S (x) V_P) Pi^T with V_S = [vec rho_s] (D x D) and
 * V_P = [vec E_m] (M x D, column e d + b <-> E_m[e][b]), so that L^+ = Pi (V_S^+ (x) V_P^+) (plain transposes,
 * routines.py:69-71): vs_pinv[D][D][2], vp_pinv[D][M][2] (each nullable).  Entry [(c d + e) D + (a d + b)][s M + m] of
 * the reference's `_lifp_oper_inv` is vs_pinv[a d + c][s] * vp_pinv[e d + b][m]. */
int qt_process_get_factors(qt_handle_t* h, double* vs_pinv, double* vp_pinv, int flags);
/* n = 2 keeps BOTH forms: the dense operator (qt_process_get_operators, 'pgdb', the process chain) and, when M % 4 == 0,
 * the factors, which qt_lifp_batch multiplies by (34 matrix instructions per process instead of 288; same Choi matrix
 * to rounding).  on != 0 makes qt_lifp_batch use the dense left inverse instead (A/B measurements, tests). */
int qt_process_prefer_dense(qt_handle_t* h, int on);
/* counts[B][D][S][K] -> choi[B][D][D][2] (process.py:284-289); cptp != 0 applies the Dykstra
 * projection of process.py:231-257 (n_iter <= 1000, stop 1e-12); iters[B] (nullable) */
int qt_lifp_batch(qt_handle_t* h, const int64_t* counts, int B, int cptp, double* choi, int32_t* iters,
                  int32_t* status, int flags);
/* 'pgdb' (process.py:291-308): projected gradient descent with backtracking from the fully mixed Choi
 * matrix, raw counts as weights, every arithmetic quirk of the reference kept (see qt_process.h).
 * stop_rule 0 = the reference's loop exit (leaves at the first step that lowers the NLL by more than
 * tol and returns the point BEFORE it -- in practice the starting point); 1 = accept steps until the
 * decrease falls below tol.  iters[B], status[B] nullable. */
int qt_pgdb_batch(qt_handle_t* h, const int64_t* counts, int B, int n_iter, double tol, int stop_rule, double* choi,
                  int32_t* iters, int32_t* status, int flags);
/* The pieces of ONE 'pgdb' iteration at the points choi_in[B][D][D][2] (process.py:296-298), for checking the
 * factored n = 3 operators against the reference's dense ones: probas[B][D*M] = Re(L c), grad[B][D][D][2] =
 * -L^H (n / p) laid out like the Choi matrix (entry [i][j] = component j D + i of the reference's column-stacked
 * vector), projected[B][D][D][2] = P_CPTP(c - grad / mu).  Each output nullable.  n = 3 only (n <= 2 keeps the whole
 * loop in one kernel, tested through its results). */
int qt_pgdb_pieces(qt_handle_t* h, const int64_t* counts, int B, const double* choi_in, double* probas, double* grad,
                   double* projected, int flags);
/* Metropolis-Hastings chains on the Choi vector (MHMCProcessInterval, interval.py:808-836): proposals
 * P_CPTP(x + step * delta), target exp(-nll) with the raw counts.  counts[C][D][S][K], choi_init[C][D][D][2],
 * deltas[C][T][D*D] (real, indexed like the column-stacked Choi vector), uniforms[C][T];
 * chain[C][T][D][D][2] = Choi matrix after every step, accepted[C][T]. */
int qt_mhmc_process(qt_handle_t* h, const int64_t* counts, int C, const double* choi_init, const double* deltas,
                    const double* uniforms, int T, double step, double* chain, int32_t* accepted, int flags);
/* projections alone (process.py:231-278): mode 0 = CPTP (Dykstra), 1 = TP, 2 = CP */
int qt_cptp_project_batch(qt_handle_t* h, const double* choi_in, int B, int mode, int n_iter, double tol,
                          double* choi_out, int32_t* iters, int flags);

#ifdef __cplusplus
}
#endif
#endif /* QTOMO_H */
