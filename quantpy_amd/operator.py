"""Unitary / general operators and the standard gate set (API of reference quantpy/operator.py).
Host-side: these are 2x2 ... 8x8 constant matrices."""
from copy import deepcopy

import numpy as np

from .base_quantum import BaseQuantum
from .qobj import Qobj
from .routines import _SIGMA_I, _SIGMA_X, _SIGMA_Y, _SIGMA_Z, _vec2mat


class Operator(BaseQuantum):
    """Matrix of a quantum operator; `transform(state)` applies U rho U^dagger."""

    def __init__(self, data):
        if isinstance(data, self.__class__):
            self.__dict__ = deepcopy(data.__dict__)
            return
        self.matrix = data

    @property
    def matrix(self):
        return self._matrix

    @matrix.setter
    def matrix(self, data):
        self._matrix = np.array(data, dtype=np.complex128)
        self.n_qubits = int(np.log2(self._matrix.shape[0]))

    def transform(self, state):
        return Qobj((self @ state @ self.H).matrix)

    def as_channel(self):
        from .channel import Channel

        return Channel(self.transform, self.n_qubits)

    def trace(self):
        return np.trace(self.matrix)

    def __repr__(self):
        return "Quantum Operator\n" + repr(self.matrix)


# ---- parametrised one-qubit gates ---------------------------------------------------------------
# noinspection PyPep8Naming
def PHASE(theta):
    return Operator(np.diag([1, np.exp(1j * theta)]))


# noinspection PyPep8Naming
def RX(theta):
    c, s = np.cos(theta / 2), np.sin(theta / 2)
    return Operator([[c, -1j * s], [-1j * s, c]])


# noinspection PyPep8Naming
def RY(theta):
    c, s = np.cos(theta / 2), np.sin(theta / 2)
    return Operator([[c, -s], [s, c]])


# noinspection PyPep8Naming
def RZ(theta):
    return Operator(np.diag([np.exp(-0.5j * theta), np.exp(0.5j * theta)]))


# ---- constant gates -----------------------------------------------------------------------------
Id = Operator(_SIGMA_I)
X = Operator(_SIGMA_X)
Y = Operator(_SIGMA_Y)
Z = Operator(_SIGMA_Z)
H = Operator(np.array([[1, 1], [1, -1]]) / np.sqrt(2))
T = PHASE(np.pi / 4)
S = PHASE(np.pi / 2)


def _controlled(u):
    """|0><0| (x) I + |1><1| (x) u"""
    dim = u.shape[0]
    out = np.eye(2 * dim, dtype=np.complex128)
    out[dim:, dim:] = u
    return out


def _permutation(perm):
    dim = len(perm)
    out = np.zeros((dim, dim))
    out[np.arange(dim), perm] = 1
    return out


CNOT = Operator(_controlled(_SIGMA_X))
CY = Operator(_controlled(_SIGMA_Y))
CZ = Operator(_controlled(_SIGMA_Z))
SWAP = Operator(_permutation([0, 2, 1, 3]))
ISWAP = Operator([[1, 0, 0, 0], [0, 0, 1j, 0], [0, 1j, 0, 0], [0, 0, 0, 1]])
MS = Operator(np.array([[1, 0, 0, 1j], [0, 1, -1j, 0], [0, -1j, 1, 0], [1j, 0, 0, 1]]) / np.sqrt(2))
Toffoli = Operator(_permutation([0, 1, 2, 3, 4, 5, 7, 6]))
Fredkin = Operator(_permutation([0, 1, 2, 3, 4, 6, 5, 7]))


def _choi_to_kraus(choi):
    """Kraus operators sqrt(v) * vec2mat(u) from the eigen-pairs of a Choi matrix."""
    values, vectors = choi.eig()
    return [Operator(_vec2mat(vec) * np.sqrt(val)) for val, vec in zip(values, vectors.T) if abs(val) > 1e-15]
