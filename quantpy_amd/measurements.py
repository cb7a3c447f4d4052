"""POVM tensors in the Bloch representation (reference quantpy/measurements.py:4-94)."""
import numpy as np

from .engine import get_engine

_S3 = 1 / np.sqrt(3)
# one-qubit tables: rows are Bloch vectors (I, X, Y, Z coefficients) of the POVM elements
_ONE_QUBIT = {
    # six projectors |+x>,|-x>,|+y>,|-y>,|0>,|1> as ONE povm (weights 1/6)
    "proj": lambda: np.array(
        [[1, 1, 0, 0], [1, -1, 0, 0], [1, 0, 1, 0], [1, 0, -1, 0], [1, 0, 0, 1], [1, 0, 0, -1]]) / 6,
    # the same projectors as THREE two-outcome povms (X, Y, Z bases)
    "proj-set": lambda: np.array(
        [[[1, 1, 0, 0], [1, -1, 0, 0]], [[1, 0, 1, 0], [1, 0, -1, 0]], [[1, 0, 0, 1], [1, 0, 0, -1]]]) / 2,
    # |+x>, |+y>, |0>, |1>
    "proj4": lambda: np.array([[1, 1, 0, 0], [1, 0, 1, 0], [1, 0, 0, 1], [1, 0, 0, -1]]) / 4,
    # tetrahedral SIC
    "sic": lambda: np.array(
        [[1, _S3, _S3, _S3], [1, _S3, -_S3, -_S3], [1, -_S3, _S3, -_S3], [1, -_S3, -_S3, _S3]]) / 4,
}


_PROBES = {}


def _probe_vectors(shape):
    """Two fixed pseudo-random rank-one probes a (x) b (x) c per tensor shape (weights in [1, 2))."""
    pv = _PROBES.get(shape)
    if pv is None:
        rng = np.random.default_rng(0x51ED0 + 31 * shape[0] + 7 * shape[1] + shape[2])
        ab = np.stack([np.kron(rng.uniform(1, 2, shape[0]), rng.uniform(1, 2, shape[1])) for _ in range(2)], axis=1)
        pv = _PROBES[shape] = (ab, rng.uniform(1, 2, (shape[2], 2)))
    return pv


class PovmTensor(np.ndarray):
    """The (S, K, 4^n) POVM tensor as a plain ndarray that also remembers the one-qubit table it is the
    n-fold tensor power of (`factor`, shape (S1, K1, 4)).  The engine uses the factor to contract
    qubit by qubit (qt_set_povm_product); anything derived from the array (slices, arithmetic,
    np.vstack, np.asarray) is ordinary data without a factor, and a position-weighted checksum guards
    against in-place edits of the tensor -- value edits and permutations of outcomes or settings alike."""

    def __new__(cls, tensor, factor):
        obj = np.ascontiguousarray(tensor, dtype=np.float64).view(cls)
        obj.factor = np.ascontiguousarray(factor, dtype=np.float64)
        obj._digest = cls._checksum(obj)
        return obj

    def __array_finalize__(self, obj):
        self.factor = None
        self._digest = None

    @staticmethod
    def _checksum(arr):
        """(shape, two position-weighted sums, their scale).  sum_{s,k,j} t[s,k,j] a[s] b[k] c[j] with fixed
        pseudo-random a, b, c: every entry carries its own weight, so moving a value changes the sum -- the
        plain sum / sum of squares used before is blind to permutations (103 of 108 outcome swaps of
        'proj-set' went unnoticed).  One GEMM pass over the tensor (64 MB at n = 5: ~5 ms)."""
        a = np.asarray(arr)
        ab, c = _probe_vectors(a.shape)
        v = a.reshape(-1, a.shape[-1]) @ c  # (S*K, 2)
        return (a.shape, float(v[:, 0] @ ab[:, 0]), float(v[:, 1] @ ab[:, 1]), float(np.abs(v).sum()) + 1e-300)

    def valid_factor(self):
        """The one-qubit table, or None when the tensor no longer is its tensor power."""
        if self.factor is None or self._digest is None:
            return None
        now = self._checksum(self)
        # BLAS may split the sums differently from call to call (thread count): rounding moves them by
        # ~1e-16 of the scale, an edit or a permutation by many orders more
        tol = 1e-12 * self._digest[3]
        if now[0] != self._digest[0] or abs(now[1] - self._digest[1]) > tol or abs(now[2] - self._digest[2]) > tol:
            return None
        return self.factor

    def digest(self):
        """The checksum taken at construction (what `valid_factor` compares against)."""
        return self._digest


def generate_measurement_matrix(povm="proj", n_qubits=1):
    """POVM tensor of shape (settings, outcomes, 4^n_qubits).

    povm : 'proj' | 'proj-set' | 'proj4' | 'sic', or an array.  Arrays whose last axis is 4 are
        one-qubit tables (2-D: one POVM, 3-D: a set) that are tensored up to `n_qubits`; arrays
        whose last axis is 4^n_qubits are returned as they are (2-D gains a leading axis).
    The n-fold Kronecker product of all three axes is assembled on the GPU (qt_povm_kron) in the
    same left-to-right multiplication order as repeated np.kron, so entries are bit-identical.
    """
    if isinstance(povm, str):
        if povm not in _ONE_QUBIT:
            raise ValueError("Incorrect string shortcut for argument `povm`")
        table = _ONE_QUBIT[povm]()
    elif isinstance(povm, np.ndarray):
        if povm.shape[-1] == 4:
            table = povm
        elif povm.shape[-1] == 4**n_qubits:
            return povm[None, :, :] if povm.ndim == 2 else povm
        else:
            raise ValueError("Incorrect POVM matrix")
    else:
        raise ValueError("Incorrect value for argument `povm`")
    if table.ndim == 2:
        table = table[None, :, :]
    if n_qubits == 1:
        return PovmTensor(table, table)
    return PovmTensor(get_engine(n_qubits).povm_kron(table), table)
