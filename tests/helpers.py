"""Shared test helpers: seeded random chains, dense contractions, bit utilities.
Mirrors the reference's test/preamble_test.jl helpers (to_dense_mps :65-72,
to_dense_mpo :74-81, apply_dense :83-101, dense_compose_mpos :103-125)."""
import numpy as np

import oracle as O
from oracle.analytic import int_to_bits


def saturated_profile(L, cap, base=2):
    """chi_i = min(base^i, base^(L-i), cap) for the L-1 internal bonds."""
    return [int(min(base ** (i + 1), base ** (L - 1 - i), cap)) for i in range(L - 1)]


def random_mps_data(bonds, rng, dtype=np.float64, normalize=True):
    dims = [1] + list(bonds) + [1]
    data = []
    for i in range(len(dims) - 1):
        shp = (dims[i], 2, dims[i + 1])
        A = rng.standard_normal(shp)
        if np.issubdtype(dtype, np.complexfloating):
            A = A + 1j * rng.standard_normal(shp)
        data.append(A.astype(dtype) / np.sqrt(dims[i] * 2.0))
    if normalize:
        nrm = O.norm(O.SignalMPS(data))
        data[0] = data[0] / nrm
    return data


def random_mpo_data(bonds, rng, dtype=np.complex128):
    dims = [1] + list(bonds) + [1]
    data = []
    for i in range(len(dims) - 1):
        shp = (dims[i], 2, 2, dims[i + 1])
        W = rng.standard_normal(shp)
        if np.issubdtype(dtype, np.complexfloating):
            W = W + 1j * rng.standard_normal(shp)
        data.append(W.astype(dtype) / np.sqrt(dims[i] * 2.0))
    return data


def dense_mps(data):
    T = data[0][0]
    for A in data[1:]:
        T = np.tensordot(T, A, axes=([-1], [0]))
    return T[..., 0]                                           # (s1..sn)


def dense_mpo(data):
    """M[in (site 1 = MSB), out (site 1 = MSB)]."""
    n = len(data)
    T = data[0][0]
    for A in data[1:]:
        T = np.tensordot(T, A, axes=([-1], [0]))
    T = T[..., 0]
    perm = list(range(0, 2 * n, 2)) + list(range(1, 2 * n, 2))
    return T.transpose(perm).reshape(2 ** n, 2 ** n)


def apply_dense(Wdata, Adata):
    """out[out bits] = sum_in M[in, out] psi[in]."""
    return dense_mpo(Wdata).T @ dense_mps(Adata).reshape(-1)


def basis_mps(j, n):
    data = []
    for b in int_to_bits(j, n):
        A = np.zeros((1, 2, 1))
        A[0, b, 0] = 1
        data.append(A)
    return O.SignalMPS(data)


def basis_ztmps(j, n):
    data = []
    for b in int_to_bits(j, n):
        for _ in range(2):
            A = np.zeros((1, 2, 1))
            A[0, b, 0] = 1
            data.append(A)
    return O.ZTMPS(data)


def interleave(main_bits, copy_bits):
    out = []
    for a, b in zip(main_bits, copy_bits):
        out += [int(a), int(b)]
    return out


def all_bits(n):
    N = 2 ** n
    return np.array([int_to_bits(j, n) for j in range(N)], dtype=np.uint8)
