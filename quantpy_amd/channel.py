"""Quantum channels in three interchangeable representations -- a map on states, a Choi matrix,
or a Kraus list (API of reference quantpy/channel.py).  Host-side glue: channels enter the hot
path only as the *input* of process tomography (the true map whose outputs are measured)."""
import sys
from copy import deepcopy

import numpy as np

from .base_quantum import BaseQuantum
from .operator import Operator, Z, _choi_to_kraus
from .qobj import Qobj, fully_mixed
from .routines import generate_single_entries, kron

_SCALARS = (int, float, complex)


class Channel(BaseQuantum):
    """data : callable (needs `n_qubits`) | ndarray / Qobj (Choi matrix) | list (Kraus operators)."""

    def __init__(self, data, n_qubits=None):
        self._types = set()
        if isinstance(data, self.__class__):
            self.__dict__ = deepcopy(data.__dict__)
            return
        self._choi = self._func = self._kraus = None
        if callable(data):
            if n_qubits is None:
                raise ValueError("`n_qubits` argument is compulsory when using init with function")
            self._func = data
            self.n_qubits = n_qubits
            self._types.add("func")
        elif isinstance(data, (np.ndarray, Qobj)):
            self._choi = Qobj(data)
            self.n_qubits = int(self._choi.n_qubits / 2)
            self._types.add("choi")
        elif isinstance(data, list):
            self._kraus = data
            self.n_qubits = data[0].n_qubits
            self._types.add("kraus")
        else:
            raise ValueError("Invalid data format")

    def set_func(self, data, n_qubits):
        """Redefine the channel by a map on states."""
        self._types = {"func"}
        self._func = data
        self.n_qubits = n_qubits

    # ---- representations ----------------------------------------------------------------------
    @property
    def choi(self):
        """sum_ij E_ij (x) channel(E_ij), built on first use from whichever form is available."""
        if "choi" not in self._types:
            dim = 2**self.n_qubits
            acc = Qobj(np.zeros((dim * dim, dim * dim), dtype=np.complex128))
            for unit in generate_single_entries(dim):
                acc += kron(Qobj(unit), self.transform(unit))
            self._choi = acc
            self._types.add("choi")
        return self._choi

    @choi.setter
    def choi(self, data):
        if not isinstance(data, Qobj):
            data = Qobj(data)
        elif not isinstance(data, np.ndarray):
            raise ValueError("Invalid data format")
        self._types = {"choi"}
        self._choi = data
        self.n_qubits = int(np.log2(data.shape[0]) / 2)

    @property
    def kraus(self):
        if "kraus" not in self._types:
            self._kraus = _choi_to_kraus(self.choi)
            self._types.add("kraus")
        return self._kraus

    @kraus.setter
    def kraus(self, data):
        if not isinstance(data, list):
            raise ValueError("Invalid data format")
        self._types = {"kraus"}
        self._kraus = data
        self.n_qubits = data[0].n_qubits

    def transform(self, state):
        """Apply the channel to a state (Qobj or array)."""
        if not isinstance(state, Qobj):
            state = Qobj(state)
        if "kraus" in self._types:
            return np.sum([op.transform(state) for op in self.kraus])
        if "func" in self._types:
            return self._func(state)
        # Choi form: Tr_in[(rho^T (x) I) C]
        lifted = kron(state.T, Qobj(np.eye(2**self.n_qubits)))
        return (lifted @ self.choi).ptrace(list(range(self.n_qubits, 2 * self.n_qubits)))

    def is_cptp(self, atol=1e-5, verbose=True):
        """Trace preservation (Tr_out C = I) and complete positivity (C >= 0) within `atol`."""
        reduced = self.choi.ptrace(list(range(self.n_qubits)))
        tp = np.allclose(reduced.matrix, np.eye(2**reduced.n_qubits), atol=atol)
        cp = np.allclose(np.minimum(np.real(self.choi.eig()[0]), 0), 0, atol=atol)
        if verbose and not tp:
            print("Not trace-preserving", file=sys.stderr)
        if verbose and not cp:
            print("Not completely positive", file=sys.stderr)
        return tp and cp

    # ---- arithmetic acts on the Choi matrix -----------------------------------------------------
    @property
    def T(self):
        return self.__class__(self.choi.T)

    @property
    def H(self):
        return self.__class__(self.choi.H)

    def conj(self):
        return self.__class__(self.choi.conj())

    def __repr__(self):
        return "Quantum channel with Choi matrix\n" + repr(self.choi.matrix)

    def _repr_latex_(self):
        return r"Choi matrix: " + Qobj(self.choi.matrix)._repr_latex_()

    def __eq__(self, other):
        return np.array_equal(self.choi.matrix, other.choi.matrix)

    def __ne__(self, other):
        return not np.array_equal(self.choi.matrix, other.choi.matrix)

    def __neg__(self):
        return self.__class__(-self.choi)

    def __add__(self, other):
        return self.__class__(self.choi + other.choi)

    def __sub__(self, other):
        return self.__class__(self.choi - other.choi)

    def __mul__(self, other):
        if not isinstance(other, _SCALARS):
            raise ValueError("Only multiplication by a scalar is allowed")
        return self.__class__(self.choi * other)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if not isinstance(other, _SCALARS):
            raise ValueError("Only division by a scalar is allowed")
        return self.__class__(self.choi / other)

    def __iadd__(self, other):
        self.choi = self.choi + other.choi
        return self

    def __isub__(self, other):
        self.choi = self.choi - other.choi
        return self

    def __imul__(self, other):
        if type(other) not in _SCALARS:
            raise ValueError("Only multiplication by a scalar is supported")
        self.choi = self.choi * other
        return self

    def __idiv__(self, other):
        if type(other) not in _SCALARS:
            raise ValueError("Only division by a scalar is supported")
        self.choi = self.choi / other
        return self


# ---- channel library --------------------------------------------------------------------------------
def depolarizing(p=1, n_qubits=1):
    """rho -> p Tr(rho) I/d + (1 - p) rho"""
    return Channel(lambda rho: p * rho.trace() * fully_mixed(n_qubits) + (1 - p) * rho, n_qubits)


def dephasing(p=1, n_qubits=1):
    """rho -> (1 - p) rho + p Z rho Z"""
    return Channel(lambda rho: p * Z.transform(rho) + (1 - p) * rho, n_qubits)


def amplitude_damping(gamma):
    """Decay |1> -> |0> with probability gamma (Kraus form)."""
    decay = np.sqrt(gamma) * Operator([[0, 1], [0, 0]])
    keep = Operator([[1, 0], [0, 0]]) + np.sqrt(1 - gamma) * Operator([[0, 0], [0, 1]])
    return Channel([decay, keep])


def walsh_hadamard(n_qubits):
    from .operator import H

    op = H
    for _ in range(n_qubits - 1):
        op = op.kron(H)
    return op.as_channel()


def depolarize(channel, p):
    """Mix `channel` with the fully depolarizing map: (1-p) channel + p Tr(.) I/d."""
    n = channel.n_qubits
    return Channel(lambda rho: (1 - p) * channel.transform(rho) + p * rho.trace() * fully_mixed(n), n)
