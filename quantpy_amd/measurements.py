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
        return table
    return get_engine(n_qubits).povm_kron(table)
