"""Arithmetic shared by Qobj / Operator / Channel (API of reference quantpy/base_quantum.py).

Every operator acts on the object's *payload* -- the `.matrix` ndarray for Qobj / Operator, the
`.choi` Qobj for Channel -- and wraps the result in the caller's class.  The dunder methods are
generated from three small templates instead of being spelled out one by one.
"""
import operator as _op
from abc import ABC, abstractmethod
from copy import deepcopy

import numpy as np

_SCALARS = (int, float, complex)  # numpy's float64 / complex128 scalars subclass these


def _payload(obj):
    return getattr(obj, obj._payload_name)


def _as_array(obj):
    data = _payload(obj)
    return data if isinstance(data, np.ndarray) else data.matrix


def _binary(fn):
    def method(self, other):
        return self.__class__(fn(_payload(self), _payload(other)))

    return method


def _scalar(fn, verb, tense):
    def method(self, other, _inplace=False):
        allowed = type(other) in _SCALARS if _inplace else isinstance(other, _SCALARS)
        if not allowed:
            raise ValueError(f"Only {verb} by a scalar is {tense}")
        return fn(_payload(self), other)

    return method


def _inplace(fn):
    def method(self, other):
        setattr(self, self._payload_name, fn(_payload(self), _payload(other)))
        return self

    return method


class BaseQuantum(ABC):
    _payload_name = "matrix"

    @abstractmethod
    def __repr__(self):
        ...

    # ---- structural maps --------------------------------------------------------------------
    @property
    def T(self):
        return self.__class__(_payload(self).T)

    @property
    def H(self):
        data = _payload(self)
        return self.__class__(data.T.conj() if isinstance(data, np.ndarray) else data.H)

    def conj(self):
        return self.__class__(_payload(self).conj())

    def copy(self):
        return deepcopy(self)

    def kron(self, other):
        return self.__class__(np.kron(self.matrix, other.matrix))

    # ---- comparisons / sign --------------------------------------------------------------------
    def __eq__(self, other):
        return np.array_equal(_as_array(self), _as_array(other))

    def __ne__(self, other):
        return not np.array_equal(_as_array(self), _as_array(other))

    def __neg__(self):
        return self.__class__(-_payload(self))

    # ---- object (+, -, @) and scalar (*, /) arithmetic ------------------------------------------
    __add__ = _binary(_op.add)
    __sub__ = _binary(_op.sub)
    __matmul__ = _binary(_op.matmul)
    __iadd__ = _inplace(_op.add)
    __isub__ = _inplace(_op.sub)

    _times = _scalar(_op.mul, "multiplication", "allowed")
    _over = _scalar(_op.truediv, "division", "allowed")
    _times_strict = _scalar(_op.mul, "multiplication", "supported")
    _over_strict = _scalar(_op.truediv, "division", "supported")

    def __mul__(self, other):
        return self.__class__(self._times(other))

    __rmul__ = __mul__

    def __truediv__(self, other):
        return self.__class__(self._over(other))

    def __imul__(self, other):
        setattr(self, self._payload_name, self._times_strict(other, _inplace=True))
        return self

    def __idiv__(self, other):
        setattr(self, self._payload_name, self._over_strict(other, _inplace=True))
        return self
