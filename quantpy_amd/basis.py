"""A (generally non-orthogonal) basis of a matrix space with its Gram matrix (API of reference
quantpy/basis.py).  Host-side set-up code for process tomography (D elements of size d x d)."""
import numpy as np
import scipy.linalg as la

from .geometry import product


class Basis:
    """elements : sequence of Qobj / arrays; inner_product : 'trace' (Tr A B^dagger) or a callable."""

    def __init__(self, elements, inner_product="trace"):
        self.elements = elements
        self.dim = len(elements)
        self.inner_product = product if inner_product == "trace" else inner_product
        self.gram = np.array(
            [[self.inner_product(a, b) for b in elements] for a in elements], dtype=np.complex128
        ).reshape(self.dim, self.dim)

    def decompose(self, obj):
        """Coefficients c with obj = sum_i c_i elements[i]."""
        rhs = np.array([self.inner_product(e, obj) for e in self.elements], dtype=np.complex128)
        return np.conj(la.solve(self.gram, rhs))

    def compose(self, vector):
        """sum_i vector[i] elements[i]."""
        return np.sum([self.elements[i] * vector[i] for i in range(self.dim)])

    def __repr__(self):
        return "Basis object\n" + repr(self.elements)
