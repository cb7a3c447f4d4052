"""A (generally non-orthogonal) basis of a matrix space with its Gram matrix (API of reference
quantpy/basis.py).  Host-side set-up code for process tomography (4^n elements of size 2^n x 2^n)."""
import numpy as np
import scipy.linalg as la

from .geometry import product


def _matrix(obj):
    return obj if isinstance(obj, np.ndarray) else obj.matrix


class Basis:
    """elements : sequence of Qobj / arrays.
    inner_product : 'trace' for (A, B) = Tr(A B^dagger), or any callable (A, B) -> scalar."""

    def __init__(self, elements, inner_product="trace"):
        self.elements = elements
        self.dim = len(elements)
        self.inner_product = product if inner_product == "trace" else inner_product
        if inner_product == "trace" and self.dim:
            stack = np.stack([np.asarray(_matrix(e), dtype=np.complex128) for e in elements])
            self.gram = np.einsum("aij,bij->ab", stack, stack.conj())  # Tr(A_a A_b^dagger) for all pairs
        else:
            self.gram = np.zeros((self.dim, self.dim), dtype=np.complex128)
            for a, ea in enumerate(elements):
                for b, eb in enumerate(elements):
                    self.gram[a, b] = self.inner_product(ea, eb)

    def decompose(self, obj):
        """Coefficients c with obj = sum_i c_i elements[i] (solves the Gram system)."""
        overlaps = np.array([self.inner_product(e, obj) for e in self.elements], dtype=np.complex128)
        return np.conj(la.solve(self.gram, overlaps))

    def compose(self, vector):
        """sum_i vector[i] elements[i]."""
        return np.sum([self.elements[i] * vector[i] for i in range(self.dim)])

    def __repr__(self):
        return "Basis object\n" + repr(self.elements)
