"""Qobj: a quantum state / operator held as a matrix and/or as a Pauli (Bloch) vector, with lazy
conversion between the two (API of reference quantpy/qobj.py).

This is host-side container code.  The two conversions are kept here, in NumPy, in exactly the
reference's floating-point evaluation order, because `StateTomograph.experiment` feeds the
Bloch vector into NumPy's legacy multinomial sampler, and for a structured state one ulp of p
decides which branch that sampler takes (tests: test_process_sampling_order):
  * bloch_k  = Re Tr(P_k M^dagger) / d     -- np.trace's pairwise summation order (qobj.py:132),
  * matrix   = sum_k bloch_k P_k            -- accumulated in increasing k (qobj.py:116-117).
The batched conversions inside the estimators run on the GPU (qt_bloch_from_mat /
qt_mat_from_bloch and the fused forms in the lin / MLE kernels).
"""
import math
import sys
from copy import deepcopy

import numpy as np
import scipy.linalg as la

from .base_quantum import BaseQuantum
from .routines import _density, pauli_tables


def _bloch_of(matrix, n_qubits):
    xmask, cols, phase = pauli_tables(n_qubits)
    d = 2**n_qubits
    rows = np.arange(d)[None, :]
    diag_terms = phase * np.conj(matrix[rows, cols])  # (P_k M^dagger)_rr for every k, r
    return np.real(np.add.reduce(diag_terms, axis=-1)) / d


def _matrix_of(bloch, n_qubits):
    xmask, cols, phase = pauli_tables(n_qubits)
    d = 2**n_qubits
    out = np.zeros((d, d), dtype=np.complex128)
    rows = np.arange(d)
    for k in range(d * d):  # increasing k: each entry receives its d terms in the reference's order
        out[rows, cols[k]] += phase[k] * bloch[k]
    return out


class Qobj(BaseQuantum):
    """Quantum object.

    data : 2-D array-like -> matrix; 1-D -> Bloch vector (length 4^n, or 4^n - 1 in which case the
        identity component 1/2^n is prepended); 1-D with `is_ket=True` -> ket |psi>, stored as
        |psi><psi|; another Qobj -> deep copy.
    """

    def __init__(self, data, is_ket=False):
        if isinstance(data, self.__class__):
            self.__dict__ = deepcopy(data.__dict__)
            return
        self._types = set()
        if is_ket:
            data = _density(data)
        data = np.array(data)
        if data.ndim == 1:
            half_log = math.log2(data.shape[0]) / 2
            self.n_qubits = math.ceil(half_log)
            dim = 2**self.n_qubits
            if half_log.is_integer():
                self._bloch = data
            else:
                self._bloch = np.ones(dim**2) / dim
                self._bloch[1:] = data
            self._matrix = None
            self._types.add("bloch")
        elif data.ndim == 2:
            self._matrix = data
            self._bloch = None
            self._types.add("matrix")
            self.n_qubits = int(np.log2(data.shape[0]))
        else:
            raise ValueError("Invalid data format")

    # ---- dual representation ----------------------------------------------------------------
    @property
    def matrix(self):
        if "matrix" not in self._types:
            self._matrix = _matrix_of(self._bloch, self.n_qubits)
            self._types.add("matrix")
        return self._matrix

    @matrix.setter
    def matrix(self, data):
        self._matrix = np.array(data)
        self._types.add("matrix")
        self._types.discard("bloch")

    @property
    def bloch(self):
        if "bloch" not in self._types:
            self._bloch = _bloch_of(self._matrix, self.n_qubits)
            self._types.add("bloch")
        return self._bloch

    @bloch.setter
    def bloch(self, data):
        self._bloch = np.array(data)
        self._types.add("bloch")
        self._types.discard("matrix")

    # ---- state / operator utilities -----------------------------------------------------------
    def ptrace(self, keep=(0,)):
        """Partial trace keeping the subsystems listed in `keep`."""
        keep = np.array(keep)
        n = self.n_qubits
        row_axes = list(range(n))
        col_axes = [n + q if q in keep else q for q in range(n)]  # repeated label = traced out
        tensor = self.matrix.reshape([2] * (2 * n))
        reduced = np.einsum(tensor, row_axes + col_axes)
        return Qobj(reduced.reshape(2 ** len(keep), 2 ** len(keep)))

    def schmidt(self):
        """SVD of the ket reshaped as a square bipartite amplitude matrix -> (U, s, Vh)."""
        side = 2 ** int(self.n_qubits / 2)
        return la.svd(np.reshape(self.ket(), (side, side)))

    def eig(self):
        """(eigenvalues, eigenvectors as columns) of the matrix (general, non-Hermitian solver)."""
        return la.eig(self.matrix)

    def is_density_matrix(self, verbose=True):
        m = self.matrix
        hermitian = np.allclose(m, m.T.conj())
        positive = np.allclose(np.minimum(np.real(self.eig()[0]), 0), 0)
        unit_trace = np.allclose(np.trace(m), 1)
        if verbose:
            for ok, text in ((hermitian, "Non-hermitian"), (positive, "Non-positive"), (unit_trace, "Trace is not 1")):
                if not ok:
                    print(text, file=sys.stderr)
        return hermitian and positive and unit_trace

    def trace(self):
        return np.trace(self.matrix)

    def impurity(self):
        return 1 - (self @ self).trace()

    def is_pure(self):
        return np.allclose(self.impurity(), 0) and self.is_density_matrix()

    def ket(self):
        if not self.is_pure():
            raise ValueError("Quantum object is not pure")
        return self.eig()[1][:, 0]

    def __repr__(self):
        return "Quantum object\n" + repr(self.matrix)

    def _repr_latex_(self):
        """LaTeX matrix for notebooks; large matrices are elided to a 10 x 10 corner."""
        m = self.matrix
        rows, cols = m.shape
        limit = 10

        def fmt(z):
            z = complex(z)
            re, im = round(z.real, 3), round(z.imag, 3)
            if im == 0:
                return f"{re:g}"
            if re == 0:
                return f"{im:g}j"
            return f"({re:g}{im:+g}j)"

        show_r, show_c = min(rows, limit), min(cols, limit)
        lines = []
        for r in range(show_r):
            cells = [fmt(m[r, c]) for c in range(show_c)]
            if cols > limit:
                cells.append(r"\cdots")
            lines.append(" & ".join(cells))
        if rows > limit:
            lines.append(" & ".join([r"\vdots"] * (show_c + (cols > limit))))
        body = r"\\".join(lines)
        return (r"Quantum object: \begin{equation*}\left(\begin{array}{*{" + str(show_c + (cols > limit)) + r"}c}" +
                body + r"\end{array}\right)\end{equation*}")


def fully_mixed(n_qubits=1):
    """I / 2^n."""
    dim = 2**n_qubits
    return Qobj(np.eye(dim, dtype=np.complex128) / dim)


# noinspection PyPep8Naming
def GHZ(n_qubits=3):
    """(|0...0> + |1...1>) / sqrt(2)."""
    ket = np.zeros(2**n_qubits)
    ket[0] = ket[-1] = 1
    return Qobj(ket / np.sqrt(2), is_ket=True)


def zero(n_qubits=1):
    """|0...0>."""
    ket = np.zeros(2**n_qubits)
    ket[0] = 1
    return Qobj(ket, is_ket=True)
