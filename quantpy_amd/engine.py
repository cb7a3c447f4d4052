"""NumPy-facing wrapper of the C ABI (include/qtomo.h): one `Engine` per (n_qubits, device).

Every method is a thin marshalling layer over one `qt_*` entry point -- the arithmetic is in the
HIP kernels.  Inputs are NumPy arrays (host-pointer calls, synchronous) or torch CUDA tensors
(device-pointer calls, asynchronous on the engine's stream; outputs must then be passed in).
"""
import ctypes
import os
import sys

import numpy as np

from . import _capi

_STATUS_TEXT = {
    _capi.TRIAL_NOT_PD: "matrix is not positive definite (Cholesky)",
    _capi.TRIAL_LINESEARCH: "line search failed (precision loss)",
    _capi.TRIAL_MAXITER: "maximum number of iterations reached",
    _capi.TRIAL_NAN: "NaN encountered",
    _capi.TRIAL_SHOTS: "per-setting totals of the counts differ from the shots registered with set_povm",
}


def _check_shots(status):
    """The reference derives the weights N_s / sum N from each trial's own results (state.py:138-141, 194-197); the
    engine registers them once per POVM, so counts measured with other shot numbers must not pass silently."""
    bad = np.flatnonzero(np.asarray(status) == _capi.TRIAL_SHOTS)
    if bad.size:
        raise ValueError(f"counts of trial(s) {bad[:8].tolist()}{'...' if bad.size > 8 else ''}: per-setting totals are not "
                         "proportional to the n_measurements registered with set_povm (reconstruct them with their own "
                         "tomograph / set_povm call)")


def _torch_stream_ptr(device):
    """hipStream_t of torch's current stream on `device`, as an integer (0 = the legacy default stream)."""
    import torch

    raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)
    if raw is not None:
        return int(raw(device))
    return int(torch.cuda.current_stream(device).cuda_stream)


class EngineError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"libqtomo error {code}: {text}")
        self.code = code


def _is_dev(x):
    return hasattr(x, "data_ptr") and hasattr(x, "is_cuda")


def _ptr(x):
    if x is None:
        return None
    if _is_dev(x):
        assert x.is_cuda and x.is_contiguous()
        return ctypes.c_void_p(x.data_ptr())
    return x.ctypes.data_as(ctypes.c_void_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _c128(a):
    return np.ascontiguousarray(a, dtype=np.complex128)


def _i64(a):
    a = np.asarray(a)
    if a.dtype != np.int64:
        if not np.issubdtype(a.dtype, np.integer):
            if not np.all(np.equal(np.mod(a, 1), 0)):
                raise TypeError("counts must be integers")
        a = a.astype(np.int64)
    return np.ascontiguousarray(a)


_DEFAULT_DEVICE = None  # (LOCAL_RANK string, device index): resolved once per process (see default_device)


def default_device():
    """The device index an engine is created on when the caller names none: the process's own GPU.
    `QTOMO_DEVICE` if set; else torch's current device when the caller has moved it off device 0
    (`torch.cuda.set_device(local_rank)`: one rank per GPU); else `LOCAL_RANK` (torchrun) modulo the number of
    VISIBLE devices (a launcher that masks each rank down to one GPU with HIP_VISIBLE_DEVICES leaves LOCAL_RANK = 3
    pointing at device 0), resolved once per process -- so that torch initialising the GPU half-way through a run
    (current device 0 by default) does not move a rank's engines onto GPU 0; else 0."""
    global _DEFAULT_DEVICE
    env = os.environ.get("QTOMO_DEVICE")
    if env is not None:
        return int(env)
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_available() and torch.cuda.is_initialized():
        cur = int(torch.cuda.current_device())
        if cur != 0 or os.environ.get("LOCAL_RANK") is None:
            return cur
    lr = os.environ.get("LOCAL_RANK")
    if _DEFAULT_DEVICE is None or _DEFAULT_DEVICE[0] != lr:
        if lr is None:
            _DEFAULT_DEVICE = (lr, 0)
        else:
            n = _capi.load().qt_device_count()
            _DEFAULT_DEVICE = (lr, int(lr) % n if n > 0 else int(lr))
    return _DEFAULT_DEVICE[1]


class Engine:
    """Owns a qt_handle (device buffers, stream, cached POVM operators) for `n_qubits`.

    Host-pointer (NumPy) calls are synchronous.  Device-pointer (`*_dev`, torch tensors) calls are
    asynchronous on the handle's stream; each of them first makes sure the handle runs on torch's
    CURRENT stream of its device (re-binding when the caller has switched streams since the last call),
    so that they are ordered with the torch work that produced their inputs and consumes their
    outputs.  `stream="own"` keeps a private non-blocking stream instead (the caller then orders
    producers / consumers itself: `sync()`); an explicit `set_stream()` pins the handle likewise."""

    def __init__(self, n_qubits, device=None, stream="torch"):
        self.lib = _capi.load()
        self.n = int(n_qubits)
        self.d = 2**self.n
        self.D = 4**self.n
        self.device = default_device() if device is None else int(device)
        self._h = self.lib.qt_create(self.device, self.n)
        if not self._h:
            raise _capi.EngineUnavailable(_capi.last_error())
        self._povm_key = None
        self._proc_key = None
        self._stream_policy = stream  # "torch": follow torch's current stream; "own" / "pinned": leave the handle's stream alone
        self._bound_ptr = None        # the torch stream pointer the handle was last bound to
        self.S = self.K = self.M = 0

    def close(self):
        if getattr(self, "_h", None):
            self.lib.qt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, code):
        if code < 0:
            raise EngineError(code, _capi.last_error())
        return code

    # ---- plumbing ---------------------------------------------------------------------------
    def sync(self):
        self._chk(self.lib.qt_sync(self._h))

    def set_stream(self, stream_ptr):
        """Pin the handle to a hipStream_t given as an integer; 0 / None = a private stream, 1 = the legacy default
        stream.  Device-pointer calls stop following torch's current stream (`follow_torch_stream()` resumes that)."""
        self._chk(self.lib.qt_set_stream(self._h, ctypes.c_void_p(stream_ptr or None)))
        self._stream_policy = "pinned"
        self._bound_ptr = None

    def follow_torch_stream(self):
        self._stream_policy = "torch"
        self._bound_ptr = None

    def bind_torch_stream(self):
        """Run this engine's work on torch's current stream of its device (stream 0 = the legacy default
        stream).  Every device-pointer call does this by itself while the policy is "torch"."""
        ptr = _torch_stream_ptr(self.device)
        if ptr != self._bound_ptr:
            self._chk(self.lib.qt_set_stream(self._h, ctypes.c_void_p(ptr if ptr else _capi.QT_STREAM_LEGACY)))
            self._bound_ptr = ptr

    def _dev_call(self):
        # binding once was a silent race for a caller that switches torch streams between calls (the handle
        # kept enqueueing on the old stream, unordered with the producers of its inputs).  One raw-pointer read per call.
        if self._stream_policy == "torch":
            self.bind_torch_stream()

    def set_option(self, option, value):
        """qt_set_option: _capi.QT_OPT_SHOTS_CHECK (0 / 1), _capi.QT_OPT_MLE_FUSED_MAX_WAVES (0 = always the split,
        two-loop BFGS path)."""
        self._chk(self.lib.qt_set_option(self._h, int(option), float(value)))

    def timer_begin(self):
        self._chk(self.lib.qt_timer_begin(self._h))

    def timer_end(self):
        ms = ctypes.c_double()
        self._chk(self.lib.qt_timer_end(self._h, ctypes.byref(ms)))
        return ms.value

    def timer_stop(self):
        """Record the end event without waiting for it (HIP events on the engine's stream)."""
        self._chk(self.lib.qt_timer_stop(self._h))

    def timer_elapsed(self):
        """Milliseconds between timer_begin() and timer_stop(); waits for the end event."""
        ms = ctypes.c_double(0.0)
        self._chk(self.lib.qt_timer_elapsed(self._h, ctypes.byref(ms)))
        return ms.value

    # ---- a1 / a2 ----------------------------------------------------------------------------
    def pauli_basis(self):
        out = np.empty((self.D, self.d, self.d), dtype=np.complex128)
        self._chk(self.lib.qt_pauli_basis(self._h, _ptr(out), _capi.QT_HOST_PTR))
        return out

    def povm_kron(self, table):
        """table (S1, K1, 4) -> (S1^n, K1^n, 4^n)"""
        table = _f64(table)
        if table.ndim == 2:
            table = table[None]
        s1, k1, four = table.shape
        assert four == 4
        out = np.empty((s1**self.n, k1**self.n, self.D))
        self._chk(self.lib.qt_povm_kron(self._h, _ptr(table), s1, k1, _ptr(out), _capi.QT_HOST_PTR))
        return out

    def povm_kron_dev(self, table, out):
        """device-pointer form: table (S1, K1, 4) float64, out (S1^n, K1^n, 4^n) float64 torch CUDA tensors"""
        self._dev_call()
        s1, k1 = (1, table.shape[0]) if table.dim() == 2 else table.shape[:2]
        assert out.numel() == (s1 * k1 * 4) ** self.n
        self._chk(self.lib.qt_povm_kron(self._h, _ptr(table), s1, k1, _ptr(out), _capi.QT_DEVICE_PTR))

    # ---- a5 ---------------------------------------------------------------------------------
    def set_povm(self, povm_matrix, n_meas):
        """Cache A (S, K, D), the shot weights and the left inverse.  Re-uploading the same
        (povm, shots) pair is skipped.  A tensor that carries its one-qubit factor
        (measurements.PovmTensor) is registered in factorised form (qt_set_povm_product)."""
        factor = povm_matrix.valid_factor() if hasattr(povm_matrix, "valid_factor") else None
        a = _f64(np.asarray(povm_matrix))
        if a.ndim == 2:
            a = a[None]
        s, k, dd = a.shape
        if dd != self.D:
            raise ValueError("Incorrect POVM matrix")
        ns = _f64(np.broadcast_to(np.asarray(n_meas, dtype=np.float64), (s,)))
        # a tensor that still is the tensor power of its factor (checksum just verified) is identified by that
        # factor; only plain arrays are compared byte by byte (64 MB at n = 5: 15 ms per call otherwise)
        key = ((a.shape, povm_matrix.digest(), ns.tobytes(), factor.tobytes()) if factor is not None
               else (a.shape, a.tobytes(), ns.tobytes(), None))
        if key == self._povm_key:
            return
        self._povm_key = None
        self._proc_key = None  # the C side drops its process set-up with the POVM (alloc_povm: proc_set = false)
        if factor is not None and (factor.shape[0] ** self.n, factor.shape[1] ** self.n) == (s, k):
            self._chk(self.lib.qt_set_povm_product(self._h, _ptr(factor), factor.shape[0], factor.shape[1], _ptr(ns),
                                                   _capi.QT_HOST_PTR))
            self.product = True
        else:
            self._chk(self.lib.qt_set_povm(self._h, _ptr(a), s, k, _ptr(ns), _capi.QT_HOST_PTR))
            self.product = False
        self._povm_key = key
        self.S, self.K, self.M = s, k, s * k

    def left_inverse(self):
        out = np.empty((self.D, self.M))
        self._chk(self.lib.qt_get_left_inverse(self._h, _ptr(out), _capi.QT_HOST_PTR))
        return out

    def left_inverse_of(self, a):
        """inv(A^T A) A^T (plain transpose) of an arbitrary real or complex matrix."""
        a = np.asarray(a)
        cplx = np.iscomplexobj(a)
        a = _c128(a) if cplx else _f64(a)
        rows, cols = a.shape
        out = np.empty((cols, rows), dtype=a.dtype)
        self._chk(self.lib.qt_left_inverse(self._h, _ptr(a), rows, cols, int(cplx), _ptr(out), _capi.QT_HOST_PTR))
        return out

    # ---- a11 - a15: process tomography ---------------------------------------------------------
    def process_setup(self, in_states):
        """in_states (D, d, d) complex: design matrix + left inverse for the POVM of `set_povm`."""
        s = _c128(in_states)
        assert s.shape == (self.D, self.d, self.d)
        key = (self._povm_key, s.tobytes())
        if key == self._proc_key:
            return
        self._proc_key = None
        self._chk(self.lib.qt_process_setup(self._h, _ptr(s), _capi.QT_HOST_PTR))
        self._proc_key = key

    def process_operators(self):
        rows, cols = self.D * self.M, self.D * self.D
        oper = np.empty((rows, cols), dtype=np.complex128)
        inv = np.empty((cols, rows), dtype=np.complex128)
        self._chk(self.lib.qt_process_get_operators(self._h, _ptr(oper), _ptr(inv), _capi.QT_HOST_PTR))
        return oper, inv

    def process_prefer_dense(self, on=True):
        """n = 2 keeps the dense left inverse AND its Kronecker factors; `lifp` multiplies by the factors unless this is
        switched on (A/B measurements, tests of the dense-operator kernels)."""
        self._chk(self.lib.qt_process_prefer_dense(self._h, int(bool(on))))

    def process_factors(self):
        """(V_S^+ (D, D), V_P^+ (D, M)): the Kronecker factors of the left inverse (n = 3; n = 2 when M % 4 == 0)."""
        vs = np.empty((self.D, self.D), dtype=np.complex128)
        vp = np.empty((self.D, self.M), dtype=np.complex128)
        self._chk(self.lib.qt_process_get_factors(self._h, _ptr(vs), _ptr(vp), _capi.QT_HOST_PTR))
        return vs, vp

    def lifp(self, counts, cptp=True, return_iters=False):
        """counts (B, D, S, K) or (D, S, K) -> Choi (B, D, D) / (D, D)."""
        c = _i64(counts)
        single = c.ndim == 3
        c = c.reshape(-1, self.D, self.S, self.K)
        b = c.shape[0]
        choi = np.empty((b, self.D, self.D), dtype=np.complex128)
        iters = np.zeros(b, dtype=np.int32)
        status = np.zeros(b, dtype=np.int32)
        self._chk(self.lib.qt_lifp_batch(self._h, _ptr(c), b, int(bool(cptp)), _ptr(choi), _ptr(iters), _ptr(status),
                                         _capi.QT_HOST_PTR))
        if single:
            choi, iters = choi[0], iters[0]
        return (choi, iters) if return_iters else choi

    def pgdb(self, counts, n_iter=1000, tol=1e-10, stop="reference", return_iters=False):
        """'pgdb' of process.py:291-308 on counts (B, D, S, K) or (D, S, K) -> Choi.  stop='reference'
        keeps the reference's loop exit (it returns the point before the first improving step),
        stop='converged' accepts steps until the NLL decrease falls below `tol`."""
        if stop not in ("reference", "converged"):
            raise ValueError("stop must be 'reference' or 'converged'")
        c = _i64(counts)
        single = c.ndim == 3
        c = c.reshape(-1, self.D, self.S, self.K)
        b = c.shape[0]
        choi = np.empty((b, self.D, self.D), dtype=np.complex128)
        iters = np.zeros(b, dtype=np.int32)
        status = np.zeros(b, dtype=np.int32)
        self._chk(self.lib.qt_pgdb_batch(self._h, _ptr(c), b, int(n_iter), float(tol), 0 if stop == "reference" else 1,
                                         _ptr(choi), _ptr(iters), _ptr(status), _capi.QT_HOST_PTR))
        if single:
            choi, iters = choi[0], iters[0]
        return (choi, iters) if return_iters else choi

    def pgdb_pieces(self, counts, choi):
        """One 'pgdb' iteration's pieces at `choi` (n = 3, qt_pgdb_pieces): (probas (D*M,), grad (D, D) laid out like
        the Choi matrix, P_CPTP(choi - grad / mu) (D, D))."""
        c = _i64(counts).reshape(1, self.D, self.S, self.K)
        x = _c128(choi).reshape(1, self.D, self.D)
        probas = np.empty((1, c[0].size))
        grad = np.empty((1, self.D, self.D), dtype=np.complex128)
        proj = np.empty((1, self.D, self.D), dtype=np.complex128)
        self._chk(self.lib.qt_pgdb_pieces(self._h, _ptr(c), 1, _ptr(x), _ptr(probas), _ptr(grad), _ptr(proj), _capi.QT_HOST_PTR))
        return probas[0], grad[0], proj[0]

    def mhmc_process(self, counts, choi_init, deltas, uniforms, step):
        """Metropolis-Hastings chain of MHMCProcessInterval: counts (D, S, K), choi_init (D, D), deltas
        (T, D*D) real (column-stacked index), uniforms (T,) -> (chain (T, D, D) complex, accepted (T,))."""
        c = _i64(counts).reshape(1, self.D, self.S, self.K)
        x0 = _c128(choi_init).reshape(1, self.D, self.D)
        dl = _f64(deltas).reshape(1, -1, self.D * self.D)
        t = dl.shape[1]
        un = _f64(uniforms).reshape(1, t)
        chain = np.empty((1, t, self.D, self.D), dtype=np.complex128)
        acc = np.zeros((1, t), dtype=np.int32)
        self._chk(self.lib.qt_mhmc_process(self._h, _ptr(c), 1, _ptr(x0), _ptr(dl), _ptr(un), t, float(step), _ptr(chain),
                                           _ptr(acc), _capi.QT_HOST_PTR))
        return chain[0], acc[0]

    def lifp_dev(self, counts, choi, cptp=True, iters=None, status=None):
        self._dev_call()
        self._chk(self.lib.qt_lifp_batch(self._h, _ptr(counts), counts.shape[0], int(bool(cptp)), _ptr(choi), _ptr(iters),
                                         _ptr(status), _capi.QT_DEVICE_PTR))

    def cptp_project(self, choi, mode="cptp", n_iter=1000, tol=1e-12, return_iters=False):
        """mode 'cptp' (Dykstra) | 'tp' | 'cp' on Choi matrices (B, D, D) / (D, D)."""
        m = {"cptp": 0, "tp": 1, "cp": 2}[mode]
        c = _c128(choi)
        single = c.ndim == 2
        c = c.reshape(-1, self.D, self.D)
        out = np.empty_like(c)
        iters = np.zeros(c.shape[0], dtype=np.int32)
        self._chk(self.lib.qt_cptp_project_batch(self._h, _ptr(c), c.shape[0], m, int(n_iter), float(tol), _ptr(out),
                                                 _ptr(iters), _capi.QT_HOST_PTR))
        if single:
            out, iters = out[0], iters[0]
        return (out, iters) if return_iters else out

    # ---- a4 / a3 ----------------------------------------------------------------------------
    def born_probs(self, bloch, out=None):
        if _is_dev(bloch):
            self._dev_call()
            b = bloch.shape[0]
            self._chk(self.lib.qt_born_probs(self._h, _ptr(bloch), b, _ptr(out), _capi.QT_DEVICE_PTR))
            return out
        bloch = _f64(bloch)
        single = bloch.ndim == 1
        bl = bloch.reshape(-1, self.D)
        p = np.empty((bl.shape[0], self.S, self.K))
        self._chk(self.lib.qt_born_probs(self._h, _ptr(bl), bl.shape[0], _ptr(p), _capi.QT_HOST_PTR))
        return p[0] if single else p

    def bloch_from_matrix(self, mat):
        mat = _c128(mat)
        single = mat.ndim == 2
        m = mat.reshape(-1, self.d, self.d)
        out = np.empty((m.shape[0], self.D))
        self._chk(self.lib.qt_bloch_from_mat(self._h, _ptr(m), m.shape[0], _ptr(out), _capi.QT_HOST_PTR))
        return out[0] if single else out

    def matrix_from_bloch(self, bloch):
        bloch = _f64(bloch)
        single = bloch.ndim == 1
        b = bloch.reshape(-1, self.D)
        out = np.empty((b.shape[0], self.d, self.d), dtype=np.complex128)
        self._chk(self.lib.qt_mat_from_bloch(self._h, _ptr(b), b.shape[0], _ptr(out), _capi.QT_HOST_PTR))
        return out[0] if single else out

    # ---- a6 / a7 ----------------------------------------------------------------------------
    def _counts(self, counts):
        c = _i64(counts)
        single = c.ndim == 2
        c = c.reshape(-1, self.S, self.K) if c.size else c.reshape(0, self.S, self.K)
        return c, single

    def lin(self, counts, physical=True, return_bloch=False):
        c, single = self._counts(counts)
        b = c.shape[0]
        rho = np.empty((b, self.d, self.d), dtype=np.complex128)
        bloch = np.empty((b, self.D)) if return_bloch else None
        status = np.zeros(b, dtype=np.int32)
        self._chk(self.lib.qt_lin_batch(self._h, _ptr(c), b, int(bool(physical)), _ptr(rho), _ptr(bloch), _ptr(status),
                                        _capi.QT_HOST_PTR))
        _check_shots(status)
        if single:
            return (rho[0], bloch[0]) if return_bloch else rho[0]
        return (rho, bloch) if return_bloch else rho

    def lin_dev(self, counts, rho, physical=True, bloch=None, status=None):
        """device-pointer form: counts int64 (B, S, K), rho complex128 (B, d, d) torch CUDA tensors"""
        self._dev_call()
        self._chk(self.lib.qt_lin_batch(self._h, _ptr(counts), counts.shape[0], int(bool(physical)), _ptr(rho),
                                        _ptr(bloch), _ptr(status), _capi.QT_DEVICE_PTR))

    def lin_dist(self, counts, centre, physical=True):
        """hs_dst(point_estimate('lin')(counts_b), centre) for every trial, in ONE pass (qt_lin_dist_batch: the
        estimates themselves are not written).  counts (B, S, K) -> (B,) float64."""
        c, _ = self._counts(counts)
        b = c.shape[0]
        cen = _c128(centre)
        assert cen.shape == (self.d, self.d)
        dist = np.empty(b)
        status = np.zeros(b, dtype=np.int32)
        self._chk(self.lib.qt_lin_dist_batch(self._h, _ptr(c), b, int(bool(physical)), _ptr(cen), None, _ptr(dist),
                                             _ptr(status), _capi.QT_HOST_PTR))
        _check_shots(status)
        return dist

    def lin_dist_dev(self, counts, centre, dist, physical=True, rho=None, status=None):
        """device-pointer form: centre complex128 (d, d), dist float64 (B,) torch CUDA tensors; rho optional"""
        self._dev_call()
        self._chk(self.lib.qt_lin_dist_batch(self._h, _ptr(counts), counts.shape[0], int(bool(physical)), _ptr(centre),
                                             _ptr(rho), _ptr(dist), _ptr(status), _capi.QT_DEVICE_PTR))

    # ---- a8 / a9 ----------------------------------------------------------------------------
    def chol_param(self, rho):
        rho = _c128(rho)
        single = rho.ndim == 2
        r = rho.reshape(-1, self.d, self.d)
        x = np.empty((r.shape[0], self.D))
        status = np.zeros(r.shape[0], dtype=np.int32)
        self._chk(self.lib.qt_chol_param(self._h, _ptr(r), r.shape[0], _ptr(x), _ptr(status), _capi.QT_HOST_PTR))
        return (x[0], status[0]) if single else (x, status)

    def chol_unparam(self, x):
        x = _f64(x)
        single = x.ndim == 1
        xx = x.reshape(-1, self.D)
        out = np.empty((xx.shape[0], self.d, self.d), dtype=np.complex128)
        self._chk(self.lib.qt_chol_unparam(self._h, _ptr(xx), xx.shape[0], _ptr(out), _capi.QT_HOST_PTR))
        return out[0] if single else out

    def nll(self, x, counts, grad=True):
        x = _f64(x)
        c, single = self._counts(counts)
        xx = x.reshape(-1, self.D)
        b = xx.shape[0]
        assert c.shape[0] == b
        f = np.empty(b)
        g = np.empty((b, self.D)) if grad else None
        self._chk(self.lib.qt_nll_batch(self._h, _ptr(xx), _ptr(c), b, _ptr(f), _ptr(g), _capi.QT_HOST_PTR))
        if single:
            return (f[0], g[0]) if grad else f[0]
        return (f, g) if grad else f

    # ---- a10 --------------------------------------------------------------------------------
    def mle(self, counts, init="lin", max_iter=100, tol=1e-3, return_info=False):
        if init not in ("lin", "mixed"):
            raise ValueError("Invalid value for argument `init`")
        c, single = self._counts(counts)
        b = c.shape[0]
        rho = np.empty((b, self.d, self.d), dtype=np.complex128)
        nit = np.zeros(b, dtype=np.int32)
        nfev = np.zeros(b, dtype=np.int32)
        fun = np.zeros(b)
        status = np.zeros(b, dtype=np.int32)
        self._chk(self.lib.qt_mle_batch(self._h, _ptr(c), b, _capi.QT_INIT_LIN if init == "lin" else _capi.QT_INIT_MIXED,
                                        int(max_iter), float(tol), _ptr(rho), _ptr(nit), _ptr(nfev), _ptr(fun),
                                        _ptr(status), _capi.QT_HOST_PTR))
        _check_shots(status)
        if single:
            rho, nit, nfev, fun, status = rho[0], nit[0], nfev[0], fun[0], status[0]
        if return_info:
            return rho, dict(nit=nit, nfev=nfev, fun=fun, status=status)
        return rho

    def mhmc_state(self, counts, x_init, deltas, uniforms, step):
        """Metropolis-Hastings chain(s) of mhmc.py on the Cholesky parameters: counts (S, K) or (C, S, K),
        x_init (D,) / (C, D), deltas (T, D) / (C, T, D), uniforms (T,) / (C, T) -> (chain (.., T, D), accepted (.., T))."""
        c, single = self._counts(counts)
        nchain = c.shape[0]
        x0 = _f64(x_init).reshape(nchain, self.D)
        dl = _f64(deltas).reshape(nchain, -1, self.D)
        t = dl.shape[1]
        un = _f64(uniforms).reshape(nchain, t)
        chain = np.empty((nchain, t, self.D))
        acc = np.zeros((nchain, t), dtype=np.int32)
        self._chk(self.lib.qt_mhmc_state(self._h, _ptr(c), nchain, _ptr(x0), _ptr(dl), _ptr(un), t, float(step),
                                         _ptr(chain), _ptr(acc), _capi.QT_HOST_PTR))
        return (chain[0], acc[0]) if single else (chain, acc)

    def mle_dev(self, counts, rho, init="lin", max_iter=100, tol=1e-3, nit=None, nfev=None, fun=None, status=None):
        self._dev_call()
        self._chk(self.lib.qt_mle_batch(self._h, _ptr(counts), counts.shape[0],
                                        _capi.QT_INIT_LIN if init == "lin" else _capi.QT_INIT_MIXED, int(max_iter),
                                        float(tol), _ptr(rho), _ptr(nit), _ptr(nfev), _ptr(fun), _ptr(status),
                                        _capi.QT_DEVICE_PTR))

    def mle_dist(self, counts, centre, init="lin", max_iter=100, tol=1e-3, return_info=False):
        """hs_dst(point_estimate('mle')(counts_b), centre) for every trial in ONE pass (qt_mle_dist_batch)."""
        if init not in ("lin", "mixed"):
            raise ValueError("Invalid value for argument `init`")
        c, _ = self._counts(counts)
        b = c.shape[0]
        cen = _c128(centre)
        assert cen.shape == (self.d, self.d)
        dist = np.empty(b)
        nit = np.zeros(b, dtype=np.int32)
        status = np.zeros(b, dtype=np.int32)
        self._chk(self.lib.qt_mle_dist_batch(self._h, _ptr(c), b, _capi.QT_INIT_LIN if init == "lin" else _capi.QT_INIT_MIXED,
                                             int(max_iter), float(tol), _ptr(cen), None, _ptr(dist), _ptr(nit), None, None,
                                             _ptr(status), _capi.QT_HOST_PTR))
        _check_shots(status)
        return (dist, dict(nit=nit, status=status)) if return_info else dist

    def mle_dist_dev(self, counts, centre, dist, init="lin", max_iter=100, tol=1e-3, rho=None, nit=None, nfev=None,
                     fun=None, status=None):
        self._dev_call()
        self._chk(self.lib.qt_mle_dist_batch(self._h, _ptr(counts), counts.shape[0],
                                             _capi.QT_INIT_LIN if init == "lin" else _capi.QT_INIT_MIXED, int(max_iter),
                                             float(tol), _ptr(centre), _ptr(rho), _ptr(dist), _ptr(nit), _ptr(nfev),
                                             _ptr(fun), _ptr(status), _capi.QT_DEVICE_PTR))

    # ---- a16 --------------------------------------------------------------------------------
    def hs_dist(self, rho, centre):
        rho = _c128(rho)
        centre = _c128(centre)
        single = rho.ndim == 2
        dim = rho.shape[-1]  # any square size (Choi matrices are 4^n x 4^n), not only this engine's 2^n
        r = rho.reshape(-1, dim, dim)
        assert centre.shape == (dim, dim)
        out = np.empty(r.shape[0])
        self._chk(self.lib.qt_hs_dist_dim(self._h, dim, _ptr(r), _ptr(centre), r.shape[0], _ptr(out), _capi.QT_HOST_PTR))
        return out[0] if single else out

    def device_multinomial(self, n, pvals, rows, seed, first_row=0, out=None):
        """`rows` multinomial draws made on the GPU (qt_device_multinomial): row r ~ multinomial(n[s], pvals[s]),
        s = (first_row + r) % S, from the Philox stream (seed, first_row + r) -- the distribution of
        `np.random.multinomial`, NOT its stream.  n (S,), pvals (S, K) on the host: counts (rows, K) int64 as a NumPy
        array; `out` = an int64 torch CUDA tensor (rows, K): n / pvals may be CUDA tensors too, filled in stream order."""
        if out is not None:
            import torch

            self._dev_call()
            dev = out.device
            n_d = n if _is_dev(n) else torch.as_tensor(np.ascontiguousarray(n, dtype=np.int64)).to(dev)
            p_d = pvals if _is_dev(pvals) else torch.as_tensor(_f64(pvals)).to(dev)
            assert out.dtype == torch.int64 and out.is_contiguous() and out.numel() == rows * p_d.shape[-1]
            self._chk(self.lib.qt_device_multinomial(self._h, seed, first_row, rows, p_d.shape[0], _ptr(n_d), _ptr(p_d),
                                                     p_d.shape[-1], _ptr(out), _capi.QT_DEVICE_PTR))
            self._keep = (n_d, p_d)  # alive until the launch has read them
            return out
        n = np.ascontiguousarray(n, dtype=np.int64)
        pvals = _f64(pvals)
        counts = np.empty((rows, pvals.shape[-1]), dtype=np.int64)
        self._chk(self.lib.qt_device_multinomial(self._h, seed, first_row, rows, pvals.shape[0], _ptr(n), _ptr(pvals),
                                                 pvals.shape[-1], _ptr(counts), _capi.QT_HOST_PTR))
        return counts

    def hs_dist_dev(self, rho, centre, out):
        self._dev_call()
        self._chk(self.lib.qt_hs_dist_batch(self._h, _ptr(rho), _ptr(centre), rho.shape[0], _ptr(out),
                                            _capi.QT_DEVICE_PTR))


    def sort_quantiles(self, dist, conf_levels):
        """interval.py:610-612 on the device: sort `dist` (NumPy array: a sorted copy is returned; torch CUDA
        tensor: sorted in place) and evaluate interp1d(linspace(0, 1, n), dist) at `conf_levels` -> NumPy array."""
        cl = _f64(np.atleast_1d(conf_levels))
        if cl.size and (cl.min() < 0 or cl.max() > 1):
            raise ValueError("A value in x_new is outside the interpolation range.")
        if _is_dev(dist):
            import torch

            self._dev_call()
            n = dist.numel()
            q = torch.from_numpy(cl).to(dist.device)
            out = torch.empty(cl.size, dtype=torch.float64, device=dist.device)
            self._chk(self.lib.qt_sort_f64(self._h, _ptr(dist), n, _capi.QT_DEVICE_PTR))
            self._chk(self.lib.qt_sorted_quantiles(self._h, _ptr(dist), n, _ptr(q), cl.size, _ptr(out), _capi.QT_DEVICE_PTR))
            self.sync()
            return out.cpu().numpy()
        srt = _f64(dist).copy()
        out = np.empty(cl.size)
        self._chk(self.lib.qt_sort_f64(self._h, _ptr(srt), srt.size, _capi.QT_HOST_PTR))
        self._chk(self.lib.qt_sorted_quantiles(self._h, _ptr(srt), srt.size, _ptr(cl), cl.size, _ptr(out), _capi.QT_HOST_PTR))
        return srt, out


    # ---- f2: stats.py:21-47 ------------------------------------------------------------------------
    def moments(self, counts, n_meas, inv_matrix):
        """Mean and variance of the squared weighted l2 error of the frequencies under multinomial noise
        (qt_moment_batch; reference stats.py l2_mean / l2_variance with the weights of interval.py:88).
        counts (B, S, K) or (S, K); n_meas (S,) shots per setting; inv_matrix (rows, S*K) = left inverse of the design
        matrix / dim.  -> (mean, var), each (B,) or scalars."""
        c = _i64(counts)
        single = c.ndim == 2
        if single:
            c = c[None]
        b, s, k = c.shape
        ns = _f64(np.broadcast_to(np.asarray(n_meas, dtype=np.float64), (s,)))
        p = _f64(np.asarray(inv_matrix).reshape(np.asarray(inv_matrix).shape[0], -1))
        assert p.shape[1] == s * k
        mean, var = np.empty(b), np.empty(b)
        self._chk(self.lib.qt_moment_batch(self._h, _ptr(c), b, s, k, _ptr(ns), _ptr(p), p.shape[0], float(ns[0]), _ptr(mean),
                                           _ptr(var), _capi.QT_HOST_PTR))
        return (mean[0], var[0]) if single else (mean, var)

    def sort_dev(self, x):
        """`x.sort()` in place for a float64 torch CUDA tensor (qt_sort_f64; NaN last like np.sort); asynchronous."""
        self._dev_call()
        self._chk(self.lib.qt_sort_f64(self._h, _ptr(x), x.numel(), _capi.QT_DEVICE_PTR))
        return x

    def quantiles_of_sorted(self, srt, conf_levels):
        """interp1d(linspace(0, 1, n), srt)(conf_levels) for an already sorted CUDA tensor -> NumPy array (synchronises)."""
        import torch

        cl = _f64(np.atleast_1d(conf_levels))
        self._dev_call()
        q = torch.from_numpy(cl).to(srt.device)
        out = torch.empty(cl.size, dtype=torch.float64, device=srt.device)
        self._chk(self.lib.qt_sorted_quantiles(self._h, _ptr(srt), srt.numel(), _ptr(q), cl.size, _ptr(out), _capi.QT_DEVICE_PTR))
        self.sync()
        return out.cpu().numpy()

    # ---- a16 over ranks: the four local steps of the distributed selection (quantpy_amd.distributed) -------------
    def select_splitters(self, srt, stride, n_split, out):
        self._dev_call()
        self._chk(self.lib.qt_select_splitters(self._h, _ptr(srt) if srt.numel() else None, srt.numel(), int(stride),
                                               int(n_split), _ptr(out), _capi.QT_DEVICE_PTR))

    def select_bracket(self, splitters, sizes, stride, n_total, levels, lo_key, hi_key):
        """splitters (N, P) float64, sizes (N,) int64, levels (L,) float64, lo_key / hi_key (L,) int64 (bit patterns)"""
        self._dev_call()
        n_ranks, n_split = splitters.shape
        self._chk(self.lib.qt_select_bracket(self._h, _ptr(splitters), n_ranks, n_split, _ptr(sizes), int(stride),
                                             int(n_total), _ptr(levels), levels.numel(), _ptr(lo_key), _ptr(hi_key),
                                             _capi.QT_DEVICE_PTR))

    def select_window(self, srt, lo_key, hi_key, width, window):
        self._dev_call()
        self._chk(self.lib.qt_select_window(self._h, _ptr(srt) if srt.numel() else None, srt.numel(), _ptr(lo_key),
                                            _ptr(hi_key), lo_key.numel(), int(width), _ptr(window), _capi.QT_DEVICE_PTR))

    def select_finish(self, windows, n_total, levels, out, overflow):
        """windows (N, L, 2 + W) float64 -> out (L,); overflow (1,) int32"""
        self._dev_call()
        n_ranks, n_lev, w2 = windows.shape
        self._chk(self.lib.qt_select_finish(self._h, _ptr(windows), n_ranks, n_lev, w2 - 2, int(n_total), _ptr(levels),
                                            _ptr(out), _ptr(overflow), _capi.QT_DEVICE_PTR))

    def merge_sorted(self, runs, lengths, out=None):
        """Merge sorted runs stored back to back (np.sort's order, NaN last).  runs: float64 torch CUDA tensor (then `out`
        likewise, asynchronous) or NumPy array (a new array is returned); lengths: host integers."""
        ln = np.ascontiguousarray(lengths, dtype=np.int64)
        if _is_dev(runs):
            import torch

            self._dev_call()
            if out is None:
                out = torch.empty_like(runs)
            assert int(ln.sum()) == runs.numel() == out.numel() and out.data_ptr() != runs.data_ptr()
            self._chk(self.lib.qt_merge_sorted(self._h, _ptr(runs), _ptr(ln), ln.size, _ptr(out), _capi.QT_DEVICE_PTR))
            return out
        r = _f64(runs)
        assert int(ln.sum()) == r.size
        res = np.empty_like(r)
        self._chk(self.lib.qt_merge_sorted(self._h, _ptr(r), _ptr(ln), ln.size, _ptr(res), _capi.QT_HOST_PTR))
        return res


_ENGINES = {}


def engine_key(n_qubits, device=None):
    return (int(n_qubits), default_device() if device is None else int(device))


def get_engine(n_qubits, device=None):
    """Process-wide engine cache: one handle per (n_qubits, device); `device=None` = this process's GPU
    (`default_device`), so that the drop-in classes of rank r run on GPU r."""
    key = engine_key(n_qubits, device)
    eng = _ENGINES.get(key)
    if eng is None:
        eng = _ENGINES[key] = Engine(key[0], key[1])
    return eng


def any_engine(device=None):
    """An engine of this process's GPU for work that does not depend on the qubit number (the device sampler, sorting):
    one that exists already, else the one-qubit engine."""
    dev = default_device() if device is None else int(device)
    for (_, d), eng in _ENGINES.items():
        if d == dev:
            return eng
    return get_engine(1, dev)


def status_text(code):
    return _STATUS_TEXT.get(int(code), "ok")
