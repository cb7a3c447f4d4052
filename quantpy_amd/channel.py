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

class Channel(BaseQuantum):
    """A channel held in one or more of three forms: 'func' (map on states), 'choi', 'kraus'.

    data : callable (needs `n_qubits`) | ndarray / Qobj (Choi matrix) | list of Kraus operators.
    Missing forms are derived on first use and cached; assigning `.choi` / `.kraus` or calling
    `set_func` makes that form the only valid one.
    """

    def __init__(self, data, n_qubits=None):
        if isinstance(data, self.__class__):
            self.__dict__ = deepcopy(data.__dict__)
            return
        self._forms = {}
        if callable(data):
            if n_qubits is None:
                raise ValueError("`n_qubits` argument is compulsory when using init with function")
            self._define("func", data, n_qubits)
        elif isinstance(data, (np.ndarray, Qobj)):
            choi = Qobj(data)
            self._define("choi", choi, int(choi.n_qubits / 2))
        elif isinstance(data, list):
            self._define("kraus", data, data[0].n_qubits)
        else:
            raise ValueError("Invalid data format")

    def _define(self, form, value, n_qubits):
        self._forms = {form: value}
        self.n_qubits = n_qubits

    @property
    def _types(self):  # the set of currently valid forms (name kept from the reference)
        return set(self._forms)

    def set_func(self, data, n_qubits):
        """Redefine the channel by a map on states."""
        self._define("func", data, n_qubits)

    # ---- representations ----------------------------------------------------------------------
    @property
    def choi(self):
        """sum_ij E_ij (x) channel(E_ij), built on first use from whichever form is available."""
        if "choi" not in self._forms:
            dim = 2**self.n_qubits
            acc = Qobj(np.zeros((dim * dim, dim * dim), dtype=np.complex128))
            for unit in generate_single_entries(dim):
                acc += kron(Qobj(unit), self.transform(unit))
            self._forms["choi"] = acc
        return self._forms["choi"]

    @choi.setter
    def choi(self, data):
        if not isinstance(data, Qobj):
            data = Qobj(data)
        self._define("choi", data, int(data.n_qubits / 2))

    @property
    def kraus(self):
        if "kraus" not in self._forms:
            self._forms["kraus"] = _choi_to_kraus(self.choi)
        return self._forms["kraus"]

    @kraus.setter
    def kraus(self, data):
        if not isinstance(data, list):
            raise ValueError("Invalid data format")
        self._define("kraus", data, data[0].n_qubits)

    def transform(self, state):
        """Apply the channel to a state (Qobj or array); Kraus form first, then the map, then Choi."""
        if not isinstance(state, Qobj):
            state = Qobj(state)
        if "kraus" in self._forms:
            return np.sum([op.transform(state) for op in self._forms["kraus"]])
        if "func" in self._forms:
            return self._forms["func"](state)
        # Choi form: Tr_in[(rho^T (x) I) C]
        lifted = kron(state.T, Qobj(np.eye(2**self.n_qubits)))
        return (lifted @ self.choi).ptrace(list(range(self.n_qubits, 2 * self.n_qubits)))

    def is_cptp(self, atol=1e-5, verbose=True):
        """Trace preservation (Tr_out C = I) and complete positivity (C >= 0) within `atol`."""
        reduced = self.choi.ptrace(list(range(self.n_qubits)))
        checks = (
            (np.allclose(reduced.matrix, np.eye(2**reduced.n_qubits), atol=atol), "Not trace-preserving"),
            (np.allclose(np.minimum(np.real(self.choi.eig()[0]), 0), 0, atol=atol), "Not completely positive"),
        )
        for passed, text in checks:
            if verbose and not passed:
                print(text, file=sys.stderr)
        return all(passed for passed, _ in checks)

    # ---- arithmetic: inherited from BaseQuantum, acting on the Choi matrix -------------------------
    _payload_name = "choi"

    def kron(self, other):
        raise NotImplementedError("Kronecker products of channels are not defined in quantpy")

    def __matmul__(self, other):
        raise NotImplementedError("composition of channels is not defined in quantpy")

    def __repr__(self):
        return "Quantum channel with Choi matrix\n" + repr(self.choi.matrix)

    def _repr_latex_(self):
        return r"Choi matrix: " + Qobj(self.choi.matrix)._repr_latex_()


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
