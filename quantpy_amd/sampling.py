"""Counts of simulated experiments on NumPy's legacy global stream (reference state.py:109-114).

The reference draws `np.random.multinomial(n_s, p_s)` once per POVM setting, one Python call each; a bootstrap
repeats that for every resample (interval.py:598-609).  `legacy_multinomial` makes the same draws -- same stream,
same order, same bits, and `np.random`'s state afterwards is what the reference's loop would have left -- in one
call of `qt_legacy_multinomial` (csrc/qt_sampler.h), which restates NumPy's legacy sampler in C and advances the
MT19937 state of the global `RandomState` in place (through the address NumPy's own `BitGenerator.ctypes` interface
publishes, under the generator's lock).

`device_multinomial` is the opt-in alternative for callers that need the distribution but not the reference's stream
(`sampler="device"` on `experiment` / `experiment_batch` / the bootstrap intervals): the draws are made on the GPU
(`qt_device_multinomial`, one Philox stream per row), which takes the sampler out of a bootstrap's critical path.
"""
import ctypes

import numpy as np

from . import _capi

_STATE_WORDS = 624  # mt19937_state: uint32 key[624]; int pos  (numpy/random/src/mt19937/mt19937.h)
_layout_checked = {}  # id(bit generator) -> (bit generator, address or None)


def _global_bit_generator():
    try:
        return np.random.get_bit_generator()
    except AttributeError:  # NumPy < 1.24
        return np.random.mtrand._rand._bit_generator


def _mt19937_address(bitgen):
    """Address of the generator's mt19937_state, or None when the memory there is not (key[624], pos) as
    `bitgen.state` reports it (checked once per generator object: a different NumPy build layout is noticed, not
    trusted)."""
    hit = _layout_checked.get(id(bitgen))
    if hit is not None and hit[0] is bitgen:
        return hit[1]
    address = None
    try:
        candidate = int(bitgen.ctypes.state_address)
        reported = bitgen.state["state"]
        words = np.frombuffer((ctypes.c_uint32 * _STATE_WORDS).from_address(candidate), dtype=np.uint32)
        pos = ctypes.c_int.from_address(candidate + 4 * _STATE_WORDS).value
        if np.array_equal(words, reported["key"]) and pos == int(reported["pos"]):
            address = candidate
    except Exception:
        address = None
    _layout_checked[id(bitgen)] = (bitgen, address)
    return address


def _numpy_loop(n, pvals, repeats):
    """The reference's own loop; used when the global generator is not MT19937 (np.random.set_bit_generator)."""
    return np.asarray([[np.random.multinomial(n_s, p_s) for p_s, n_s in zip(pvals, n)] for _ in range(repeats)],
                      dtype=np.int64).reshape(repeats, len(n), pvals.shape[1])


def legacy_multinomial(n, pvals, repeats=1):
    """`repeats` x [np.random.multinomial(n[s], pvals[s]) for s in range(S)], drawn repeat after repeat, setting after
    setting, from `np.random`'s global state.  n: (S,) (converted to integers the way RandomState.multinomial's
    `long n` argument is), pvals: (S, K).  Returns int64 (repeats, S, K)."""
    pvals = np.ascontiguousarray(pvals, dtype=np.float64)
    if pvals.ndim != 2:
        raise ValueError("pvals must be (settings, outcomes)")
    n_set, n_out = pvals.shape
    n = np.ascontiguousarray(np.asarray(n).astype(np.int64))
    if n.shape != (n_set,):
        raise ValueError("one `n` per row of pvals")
    bitgen = _global_bit_generator()
    if type(bitgen).__name__ != "MT19937":
        return _numpy_loop(n, pvals, repeats)
    lib = _capi.load()
    out = np.empty((repeats, n_set, n_out), dtype=np.int64)
    address = _mt19937_address(bitgen)
    if address is not None:
        with bitgen.lock:
            rc = lib.qt_legacy_multinomial(address, ctypes.cast(address + 4 * _STATE_WORDS, ctypes.POINTER(ctypes.c_int)),
                                           repeats * n_set, n_set, n.ctypes.data, pvals.ctypes.data, n_out, out.ctypes.data)
    else:  # copy the state out and back
        state = np.random.get_state()
        key = np.ascontiguousarray(state[1], dtype=np.uint32).copy()
        pos = ctypes.c_int(int(state[2]))
        rc = lib.qt_legacy_multinomial(key.ctypes.data, ctypes.byref(pos), repeats * n_set, n_set, n.ctypes.data,
                                       pvals.ctypes.data, n_out, out.ctypes.data)
        if rc >= 0:
            np.random.set_state((state[0], key, pos.value) + tuple(state[3:]))
    if rc < 0:
        raise ValueError(_capi.last_error())  # NumPy raises ValueError for the same conditions
    return out


SAMPLERS = ("numpy", "device")


def resolve_seed(seed):
    """64-bit Philox key: the caller's, or two words of np.random's global stream (np.random.seed keeps a run reproducible)."""
    if seed is None:
        lo, hi = (int(w) for w in np.random.randint(0, 2**32, size=2, dtype=np.uint64))
        seed = lo | (hi << 32)
    return int(seed) & (2**64 - 1)


def device_multinomial(n, pvals, repeats=1, seed=None, engine=None):
    """The table `legacy_multinomial` returns, in distribution: int64 (repeats, S, K) with row (r, s) ~
    multinomial(n[s], pvals[s]), drawn on the GPU from the Philox stream (seed, r * S + s).  seed=None takes 64 bits
    from `np.random`'s global stream (so `np.random.seed` still makes a run reproducible); the counts are NOT those
    of the reference for that seed."""
    pvals = np.ascontiguousarray(pvals, dtype=np.float64)
    if pvals.ndim != 2:
        raise ValueError("pvals must be (settings, outcomes)")
    n_set, n_out = pvals.shape
    n = np.ascontiguousarray(np.asarray(n).astype(np.int64))
    if n.shape != (n_set,):
        raise ValueError("one `n` per row of pvals")
    seed = resolve_seed(seed)
    if engine is None:
        from .engine import any_engine

        engine = any_engine()
    from .engine import EngineError

    try:
        counts = engine.device_multinomial(n, pvals, repeats * n_set, int(seed) & (2**64 - 1))
    except EngineError as err:
        if err.code == _capi.QT_ERR_ARG:
            raise ValueError(_capi.last_error()) from None
        raise
    return counts.reshape(repeats, n_set, n_out)


def draw_counts(n, pvals, repeats=1, sampler="numpy", seed=None):
    """One entry for both samplers: 'numpy' = the reference's stream bit for bit (seed must be None: the stream is
    `np.random`'s), 'device' = the GPU sampler."""
    if sampler == "numpy":
        if seed is not None:
            raise ValueError("sampler='numpy' draws from np.random's global stream: seed it with np.random.seed")
        return legacy_multinomial(n, pvals, repeats)
    if sampler == "device":
        return device_multinomial(n, pvals, repeats, seed)
    raise ValueError(f"sampler must be one of {SAMPLERS}, not {sampler!r}")
