"""Numerical helpers with quantpy's names (reference quantpy/routines.py).  Anything on the hot
path (Pauli basis assembly, left inverse, Cholesky parametrisation) is a call into the HIP
engine; index bookkeeping (vec/mat reshapes, the partial-trace operator) stays on the host."""
import functools

import numpy as np

from .engine import get_engine

_SIGMA_I = np.array([[1, 0], [0, 1]], dtype=np.complex128)
_SIGMA_X = np.array([[0, 1], [1, 0]], dtype=np.complex128)
_SIGMA_Y = np.array([[0, -1j], [1j, 0]], dtype=np.complex128)
_SIGMA_Z = np.array([[1, 0], [0, -1]], dtype=np.complex128)
_PAULI_1 = [_SIGMA_I, _SIGMA_X, _SIGMA_Y, _SIGMA_Z]


def generate_pauli(n_qubits):
    """n-qubit Pauli basis, index k = sum_j k_j 4^(n-1-j)  (reference routines.py:14-19; like the
    reference, one qubit gives the plain list of the four 2x2 matrices).  Assembled on the GPU
    (qt_pauli_basis)."""
    if n_qubits == 1:
        return _PAULI_1
    return get_engine(n_qubits).pauli_basis()


@functools.lru_cache(maxsize=None)
def pauli_tables(n_qubits):
    """Sparse description of the Pauli strings used by the host-side Qobj conversions:
    P_k[r, r ^ xmask[k]] = phase[k, r], every other entry zero."""
    d, dd = 2**n_qubits, 4**n_qubits
    k = np.arange(dd)
    xmask = np.zeros(dd, dtype=np.int64)
    zmask = np.zeros(dd, dtype=np.int64)
    n_y = np.zeros(dd, dtype=np.int64)
    for b in range(n_qubits):
        digit = (k >> (2 * b)) & 3
        xmask |= np.where((digit == 1) | (digit == 2), 1 << b, 0)
        zmask |= np.where((digit == 2) | (digit == 3), 1 << b, 0)
        n_y += digit == 2
    r = np.arange(d)
    parity = np.zeros((dd, d), dtype=np.int64)
    for b in range(n_qubits):
        parity ^= ((r[None, :] & zmask[:, None]) >> b) & 1
    phase = ((-1j) ** (n_y % 4))[:, None] * (1 - 2 * parity)
    cols = r[None, :] ^ xmask[:, None]
    return xmask, cols, phase


def generate_single_entries(dim):
    """All dim x dim matrices with a single unit entry, row-major order (routines.py:22-31)."""
    out = []
    for flat in range(dim * dim):
        e = np.zeros((dim, dim))
        e.flat[flat] = 1
        out.append(e)
    return out


def kron(A, B):
    """`A.kron(B)` for Qobj / Operator / Channel instances."""
    return A.kron(B)


def join_gates(gates):
    """Compose gates applied left to right: gates[-1] @ ... @ gates[0]."""
    joined = gates[0]
    for gate in gates[1:]:
        joined = gate @ joined
    return joined


def _out_ptrace_oper(n_qubits):
    """(D, D^2) operator tracing out the output half of a column-stacked bipartite matrix
    (routines.py:47-50).  0/1 entries: built by index instead of a sum of Kronecker products."""
    d = 2**n_qubits
    oper = np.zeros((d * d, d**4))
    i, o, j = np.meshgrid(np.arange(d), np.arange(d), np.arange(d), indexing="ij")
    # vec index of C[(j,o),(i,o)] under column stacking: col * d^2 + row
    rows = (i * d + j).ravel()
    cols = ((i * d + o) * d * d + (j * d + o)).ravel()
    oper[rows, cols] = 1
    return oper


def _vec2mat(vector):
    """Column-stacked vector -> square matrix."""
    side = int(np.sqrt(len(vector)))
    return vector.reshape(side, side).T


def _mat2vec(matrix):
    """Square matrix -> column-stacked vector."""
    return matrix.T.reshape(np.prod(matrix.shape))


def _density(psi):
    """|psi><psi| for a ket given as a sequence."""
    psi = np.asarray(psi, dtype=np.complex128)
    return np.outer(psi.T, np.conj(psi))


def _left_inv(A):
    """inv(A^T A) A^T with a plain (non-conjugating) transpose, as routines.py:69-71 defines it;
    computed on the GPU (Gram GEMM, pivoted Gauss-Jordan, GEMM)."""
    A = np.asarray(A)
    return get_engine(1).left_inverse_of(A)


def _real_to_complex(z):
    half = len(z) // 2
    return z[:half] + 1j * z[half:]


def _complex_to_real(z):
    return np.concatenate((np.real(z), np.imag(z)))


def _matrix_to_real_tril_vec(matrix):
    """Cholesky parametrisation [diag L | Re L_(i>j) | Im L_(i>j)]  (routines.py:84-90), on the GPU.
    Raises numpy.linalg.LinAlgError where scipy.linalg.cholesky would."""
    matrix = np.asarray(matrix, dtype=np.complex128)
    n_qubits = int(np.log2(matrix.shape[0]))
    x, status = get_engine(n_qubits).chol_param(matrix)
    if status != 0:
        raise np.linalg.LinAlgError("matrix is not positive definite")
    return x


def _real_tril_vec_to_matrix(vector):
    """L L^dagger from the Cholesky parametrisation (routines.py:93-101), on the GPU."""
    vector = np.asarray(vector, dtype=np.float64)
    n_qubits = int(round(np.log2(len(vector)) / 2))
    return get_engine(n_qubits).chol_unparam(vector)
