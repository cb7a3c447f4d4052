"""ctypes binding of libqtomo.so (include/qtomo.h).  Loads the in-tree shared library built by
`__graft_entry__.build()` / `quantpy_amd.build.build_library()`; raises if it is missing --
there is deliberately no CPU fallback behind this module."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QTOMO_LIB") or os.path.join(_HERE, "lib", "libqtomo.so")  # QTOMO_LIB: profile build

QT_HOST_PTR = 0
QT_DEVICE_PTR = 1
QT_INIT_LIN = 0
QT_INIT_MIXED = 1
QT_STREAM_LEGACY = 1  # qt_set_stream: the legacy default ("null") stream
QT_OPT_SHOTS_CHECK, QT_OPT_MLE_FUSED_MAX_WAVES = 1, 2  # qt_set_option

# status codes (include/qtomo.h)
QT_ERR_ARG, QT_ERR_STATE, QT_ERR_HIP, QT_ERR_SINGULAR, QT_ERR_UNSUPPORTED = -1, -2, -3, -4, -5
TRIAL_OK, TRIAL_NOT_PD, TRIAL_LINESEARCH, TRIAL_MAXITER, TRIAL_NAN, TRIAL_SHOTS = 0, 1, 2, 3, 4, 5

_c_int = ctypes.c_int
_c_dbl = ctypes.c_double
_vp = ctypes.c_void_p

# name -> (restype, argtypes); every symbol declared in include/qtomo.h appears here
SIGNATURES = {
    "qt_version": (_c_int, []),
    "qt_last_error": (ctypes.c_char_p, []),
    "qt_device_count": (_c_int, []),
    "qt_create": (_vp, [_c_int, _c_int]),
    "qt_destroy": (None, [_vp]),
    "qt_sync": (_c_int, [_vp]),
    "qt_set_stream": (_c_int, [_vp, _vp]),
    "qt_set_option": (_c_int, [_vp, _c_int, _c_dbl]),
    "qt_timer_begin": (_c_int, [_vp]),
    "qt_timer_end": (_c_int, [_vp, ctypes.POINTER(_c_dbl)]),
    "qt_timer_stop": (_c_int, [_vp]),
    "qt_timer_elapsed": (_c_int, [_vp, ctypes.POINTER(_c_dbl)]),
    "qt_pauli_basis": (_c_int, [_vp, _vp, _c_int]),
    "qt_povm_kron": (_c_int, [_vp, _vp, _c_int, _c_int, _vp, _c_int]),
    "qt_set_povm": (_c_int, [_vp, _vp, _c_int, _c_int, _vp, _c_int]),
    "qt_set_povm_product": (_c_int, [_vp, _vp, _c_int, _c_int, _vp, _c_int]),
    "qt_get_left_inverse": (_c_int, [_vp, _vp, _c_int]),
    "qt_left_inverse": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _vp, _c_int]),
    "qt_born_probs": (_c_int, [_vp, _vp, _c_int, _vp, _c_int]),
    "qt_bloch_from_mat": (_c_int, [_vp, _vp, _c_int, _vp, _c_int]),
    "qt_mat_from_bloch": (_c_int, [_vp, _vp, _c_int, _vp, _c_int]),
    "qt_lin_batch": (_c_int, [_vp, _vp, _c_int, _c_int, _vp, _vp, _vp, _c_int]),
    "qt_lin_dist_batch": (_c_int, [_vp, _vp, _c_int, _c_int, _vp, _vp, _vp, _vp, _c_int]),
    "qt_chol_param": (_c_int, [_vp, _vp, _c_int, _vp, _vp, _c_int]),
    "qt_chol_unparam": (_c_int, [_vp, _vp, _c_int, _vp, _c_int]),
    "qt_nll_batch": (_c_int, [_vp, _vp, _vp, _c_int, _vp, _vp, _c_int]),
    "qt_mle_batch": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_dbl, _vp, _vp, _vp, _vp, _vp, _c_int]),
    "qt_mle_dist_batch": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_dbl, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _c_int]),
    "qt_mhmc_state": (_c_int, [_vp, _vp, _c_int, _vp, _vp, _vp, _c_int, _c_dbl, _vp, _vp, _c_int]),
    "qt_hs_dist_batch": (_c_int, [_vp, _vp, _vp, _c_int, _vp, _c_int]),
    "qt_hs_dist_dim": (_c_int, [_vp, _c_int, _vp, _vp, _c_int, _vp, _c_int]),
    "qt_sort_f64": (_c_int, [_vp, _vp, ctypes.c_longlong, _c_int]),
    "qt_sorted_quantiles": (_c_int, [_vp, _vp, ctypes.c_longlong, _vp, _c_int, _vp, _c_int]),
    "qt_select_splitters": (_c_int, [_vp, _vp, ctypes.c_longlong, ctypes.c_longlong, _c_int, _vp, _c_int]),
    "qt_select_bracket": (_c_int, [_vp, _vp, _c_int, _c_int, _vp, ctypes.c_longlong, ctypes.c_longlong, _vp, _c_int, _vp, _vp,
                                   _c_int]),
    "qt_select_window": (_c_int, [_vp, _vp, ctypes.c_longlong, _vp, _vp, _c_int, _c_int, _vp, _c_int]),
    "qt_select_finish": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, ctypes.c_longlong, _vp, _vp, _vp, _c_int]),
    "qt_merge_sorted": (_c_int, [_vp, _vp, _vp, _c_int, _vp, _c_int]),
    "qt_moment_batch": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _vp, _vp, _c_int, _c_dbl, _vp, _vp, _c_int]),
    "qt_legacy_multinomial": (_c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.c_longlong, _c_int, _vp, _vp, _c_int, _vp]),
    "qt_device_multinomial": (_c_int, [_vp, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_longlong, _c_int, _vp, _vp, _c_int, _vp,
                                       _c_int]),
    "qt_philox4x32_10": (None, [_vp, _vp, _vp]),
    "qt_pgdb_pieces": (_c_int, [_vp, _vp, _c_int, _vp, _vp, _vp, _vp, _c_int]),
    "qt_process_setup": (_c_int, [_vp, _vp, _c_int]),
    "qt_process_get_operators": (_c_int, [_vp, _vp, _vp, _c_int]),
    "qt_process_get_factors": (_c_int, [_vp, _vp, _vp, _c_int]),
    "qt_process_prefer_dense": (_c_int, [_vp, _c_int]),
    "qt_lifp_batch": (_c_int, [_vp, _vp, _c_int, _c_int, _vp, _vp, _vp, _c_int]),
    "qt_pgdb_batch": (_c_int, [_vp, _vp, _c_int, _c_int, _c_dbl, _c_int, _vp, _vp, _vp, _c_int]),
    "qt_mhmc_process": (_c_int, [_vp, _vp, _c_int, _vp, _vp, _vp, _c_int, _c_dbl, _vp, _vp, _c_int]),
    "qt_cptp_project_batch": (_c_int, [_vp, _vp, _c_int, _c_int, _c_int, _c_dbl, _vp, _vp, _c_int]),
}

_lib = None


class EngineUnavailable(RuntimeError):
    """The HIP library (or a HIP device) is missing.  Nothing falls back to the CPU."""


def load():
    """Load libqtomo.so once and attach the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own HSA/HIP runtime.  If libqtomo.so (linked against the system
    # ROCm) initialises the GPU first and torch is imported afterwards, the process ends up with two
    # HSA runtimes and torch reports "No HIP GPUs are available"; the other order works (the loader
    # shares the HSA runtime by soname).  So when torch is installed, let it load first.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise EngineUnavailable(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  quantpy_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header / library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    return load().qt_last_error().decode("utf-8", "replace")
