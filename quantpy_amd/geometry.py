"""Distances between quantum objects (reference quantpy/geometry.py)."""
import numpy as np
import scipy.linalg as la


def _as_matrix(obj):
    return obj if isinstance(obj, np.ndarray) else obj.matrix


def hs_dst(A, B):
    """Hilbert-Schmidt distance sqrt(|Tr((A-B)^2)|)/sqrt(2), zero below 1e-15 (geometry.py:16-20).
    Evaluated by the HIP engine (qt_hs_dist_batch) -- the same kernel the bootstrap uses."""
    from .engine import get_engine

    a, b = _as_matrix(A), _as_matrix(B)
    n_qubits = int(np.log2(a.shape[0]))
    # (the kernel takes any square size; a 64 x 64 Choi matrix of a 3-qubit channel runs on the 3-qubit engine)
    dist = float(get_engine(n_qubits if n_qubits <= 5 else n_qubits // 2).hs_dist(a, b))
    return 0 if dist < 1e-15 else dist


def trace_dst(A, B):
    """Trace distance |Tr sqrt((A-B)^2)| / 2 (host: a reporting metric, not on the hot path)."""
    diff = _as_matrix(A) - _as_matrix(B)
    dist = abs(np.trace(la.sqrtm(diff @ diff))) / 2
    return 0 if dist < 1e-15 else dist


def if_dst(A, B):
    """Infidelity 1 - |Tr sqrt(sqrt(A) B sqrt(A))|^2 (host: a reporting metric)."""
    a, b = _as_matrix(A), _as_matrix(B)
    root = la.sqrtm(a)
    dist = 1 - np.abs(np.trace(la.sqrtm(root @ b @ root)) ** 2)
    return 0 if dist < 1e-15 else dist


def product(A, B):
    """Hermitian inner product Tr(A B^dagger)."""
    return np.trace(_as_matrix(A) @ np.conj(_as_matrix(B).T), dtype=np.complex128)
