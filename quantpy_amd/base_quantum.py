"""Arithmetic shared by Qobj / Operator / Channel (reference quantpy/base_quantum.py): every
operator acts on `.matrix` and wraps the result in the caller's class."""
from abc import ABC, abstractmethod
from copy import deepcopy

import numpy as np

_SCALARS = (int, float, complex)


class BaseQuantum(ABC):
    @abstractmethod
    def __repr__(self):
        ...

    def _wrap(self, matrix):
        return self.__class__(matrix)

    @property
    def T(self):
        return self._wrap(self.matrix.T)

    @property
    def H(self):
        return self._wrap(self.matrix.T.conj())

    def conj(self):
        return self._wrap(self.matrix.conj())

    def copy(self):
        return deepcopy(self)

    def kron(self, other):
        return self._wrap(np.kron(self.matrix, other.matrix))

    def __eq__(self, other):
        return np.array_equal(self.matrix, other.matrix)

    def __ne__(self, other):
        return not np.array_equal(self.matrix, other.matrix)

    def __neg__(self):
        return self._wrap(-self.matrix)

    def __matmul__(self, other):
        return self._wrap(self.matrix @ other.matrix)

    def __add__(self, other):
        return self._wrap(self.matrix + other.matrix)

    def __sub__(self, other):
        return self._wrap(self.matrix - other.matrix)

    def __mul__(self, other):
        if not isinstance(other, _SCALARS):
            raise ValueError("Only multiplication by a scalar is allowed")
        return self._wrap(self.matrix * other)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if not isinstance(other, _SCALARS):
            raise ValueError("Only division by a scalar is allowed")
        return self._wrap(self.matrix / other)

    def __iadd__(self, other):
        self.matrix = self.matrix + other.matrix
        return self

    def __isub__(self, other):
        self.matrix = self.matrix - other.matrix
        return self

    def __imul__(self, other):
        if type(other) not in _SCALARS:
            raise ValueError("Only multiplication by a scalar is supported")
        self.matrix = self.matrix * other
        return self

    def __idiv__(self, other):
        if type(other) not in _SCALARS:
            raise ValueError("Only division by a scalar is supported")
        self.matrix = self.matrix / other
        return self
