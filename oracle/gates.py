"""Elementary gates and bond-dim-2 controlled MPO blocks (test infrastructure).

Follows src/circuits/qft_gates.jl:12-97, dt_gates.jl:11-229, zt_gates.jl:12-114.
A gate is the 2x2 matrix M[s_in, s_out] (the reference builds ``ITensor(M, s', s)``
and ``apply`` contracts the primed leg with the state, apply.jl:98-101); blocks
are lists of W[a, s_in, s_out, b] with explicit dim-1 edge bonds.
"""
from __future__ import annotations

import numpy as np

from .containers import SingleSiteMPO, PairedSiteMPO


# ---------------------------------------------------------------- elementary gates
def gate_I(dtype=np.float64):
    """qft_gates.jl:12."""
    return np.eye(2, dtype=dtype)


def gate_H(dtype=np.float64):
    """qft_gates.jl:15-21."""
    return (np.array([[1, 1], [1, -1]], dtype=dtype) / np.sqrt(2.0)).astype(dtype)


def gate_P(theta):
    """qft_gates.jl:24-30: diag(1, exp(-i theta))."""
    return np.array([[1, 0], [0, np.exp(-1j * theta)]], dtype=np.complex128)


def gate_Pi(i, dtype=np.float64):
    """qft_gates.jl:32-38: projector on |i>."""
    if i not in (0, 1):
        raise ValueError(f"Pi: index dimension is less than {i}")
    M = np.zeros((2, 2), dtype=dtype)
    M[i, i] = 1
    return M


def gate_dampedH(wr, dtype=np.float64):
    """dt_gates.jl:11-17."""
    return (np.array([[1, 1], [1, np.exp(-wr / 2.0)]], dtype=dtype) / np.sqrt(2.0)).astype(dtype)


def gate_R(wr, dtype=np.float64):
    """dt_gates.jl:19-25: diag(1, exp(-wr))."""
    return np.array([[1, 0], [0, np.exp(-wr)]], dtype=dtype)


def _site(Dl, Dr, dtype):
    return np.zeros((Dl, 2, 2, Dr), dtype=dtype)


# ---------------------------------------------------------------- QFT block
def control_Hphase_mpo(k, sites=None):
    """qft_gates.jl:43-97.  First site: H, then project the OUTPUT onto |c> and
    emit bond value c (:76-79); site l = 2..k: I (bond 0) or P(2pi/2^l) (bond 1)."""
    if k < 1:
        raise ValueError(f"control_Hphase_mpo: k must be >= 1, got {k}")
    if sites is not None and len(sites) != k:
        raise ValueError("control_Hphase_mpo: number of sites must equal k")
    ct = np.complex128
    if k == 1:
        return SingleSiteMPO([gate_H(ct).reshape(1, 2, 2, 1)], sites)
    H = gate_H(ct)
    data = []
    W = _site(1, 2, ct)
    for c in (0, 1):
        # H[s_in, tmp] * Pi_c[tmp, s_out] * onehot(bond = c)
        W[0, :, :, c] = H @ gate_Pi(c, ct)
    data.append(W)
    for l in range(2, k):
        W = _site(2, 2, ct)
        W[0, :, :, 0] = gate_I(ct)
        W[1, :, :, 1] = gate_P(2 * np.pi / 2.0 ** l)
        data.append(W)
    W = _site(2, 1, ct)
    W[0, :, :, 0] = gate_I(ct)
    W[1, :, :, 0] = gate_P(2 * np.pi / 2.0 ** k)
    data.append(W)
    return SingleSiteMPO(data, sites)


# ---------------------------------------------------------------- DT blocks
def control_damping_mpo(n, k, wr, sites=None):
    """dt_gates.jl:30-130.  Control = INPUT bit of main_k (project, then dampedH,
    :102-113); if 1, R(wr * 2^(l-k-1)) on main_l for l < k; copy sites carry the bond."""
    if k < 1:
        raise ValueError(f"control_damping_mpo: k must be >= 1, got {k}")
    if sites is not None and len(sites) != 2 * k:
        raise ValueError("control_damping_mpo: number of sites must equal 2k")
    dt = np.float64
    sm = None if sites is None else list(sites[0::2])
    sc = None if sites is None else list(sites[1::2])
    if k == 1:
        return PairedSiteMPO([gate_dampedH(wr).reshape(1, 2, 2, 1),
                              gate_I(dt).reshape(1, 2, 2, 1)], sm, sc)
    data = []
    for l in range(1, k):
        Rf = gate_R(wr * 2.0 ** (l - k - 1))
        W = _site(1 if l == 1 else 2, 2, dt)
        if l == 1:
            W[0, :, :, 0] = gate_I(dt)
            W[0, :, :, 1] = Rf
        else:
            W[0, :, :, 0] = gate_I(dt)
            W[1, :, :, 1] = Rf
        data.append(W)
        C = _site(2, 2, dt)
        C[0, :, :, 0] = gate_I(dt)
        C[1, :, :, 1] = gate_I(dt)
        data.append(C)
    Hd = gate_dampedH(wr)
    W = _site(2, 2, dt)
    for c in (0, 1):
        # Pi_c[s_in, tmp] * Hd[tmp, s_out] on bond values (c, c)
        W[c, :, :, c] = gate_Pi(c, dt) @ Hd
    data.append(W)
    C = _site(2, 1, dt)
    C[0, :, :, 0] = gate_I(dt)
    C[1, :, :, 0] = gate_I(dt)
    data.append(C)
    return PairedSiteMPO(data, sm, sc)


def control_damping_copy_mpo(n, k, wr, sites=None):
    """dt_gates.jl:133-229.  Acts on pairs k..n (L = n-k+1).  Control = projector on
    copy_k (:184-189); if 1, R(wr * 2^(j-2)) on relative main_j, j = 2..L."""
    if k < 1:
        raise ValueError(f"control_damping_copy_mpo: k must be >= 1, got {k}")
    L = n - k + 1
    if sites is not None and len(sites) != 2 * L:
        raise ValueError("control_damping_copy_mpo: number of sites must equal 2(n-k+1)")
    dt = np.float64
    sm = None if sites is None else list(sites[0::2])
    sc = None if sites is None else list(sites[1::2])
    if L == 1:
        eye = gate_I(dt).reshape(1, 2, 2, 1)
        return PairedSiteMPO([eye.copy(), eye.copy()], sm, sc)
    data = []
    W = _site(1, 2, dt)
    W[0, :, :, 0] = gate_I(dt)                       # only bond value 0 populated (:181)
    data.append(W)
    C = _site(2, 2, dt)
    C[0, :, :, 0] = gate_Pi(0, dt)
    C[0, :, :, 1] = gate_Pi(1, dt)
    data.append(C)
    for j in range(2, L + 1):
        Rf = gate_R(wr * 2.0 ** (j - 2))
        W = _site(2, 2, dt)
        W[0, :, :, 0] = gate_I(dt)
        W[1, :, :, 1] = Rf
        data.append(W)
        last = j == L
        C = _site(2, 1 if last else 2, dt)
        C[0, :, :, 0] = gate_I(dt)
        C[1, :, :, 0 if last else 1] = gate_I(dt)
        data.append(C)
    return PairedSiteMPO(data, sm, sc)


# ---------------------------------------------------------------- zT (paired QFT) block
def control_Hphase_ztmps_mpo(k, sites=None):
    """zt_gates.jl:12-114.  Acts on COPY sites; control = INPUT bit of copy_k
    (project, then H: index roles at :104-107, gate test test_zt_gates.jl:42-68);
    if 1, P(2pi/2^(k-j+1)) on copy_j, j < k; main sites pass the bond through."""
    if k < 1:
        raise ValueError(f"control_Hphase_ztmps_mpo: k must be >= 1, got {k}")
    if sites is not None and len(sites) != 2 * k:
        raise ValueError("control_Hphase_ztmps_mpo: number of sites must equal 2k")
    ct = np.complex128
    sm = None if sites is None else list(sites[0::2])
    sc = None if sites is None else list(sites[1::2])
    if k == 1:
        return PairedSiteMPO([gate_I(ct).reshape(1, 2, 2, 1),
                              gate_H(ct).reshape(1, 2, 2, 1)], sm, sc)
    data = []
    W = _site(1, 2, ct)
    W[0, :, :, 0] = gate_I(ct)
    W[0, :, :, 1] = gate_I(ct)
    data.append(W)
    C = _site(2, 2, ct)
    C[0, :, :, 0] = gate_I(ct)
    C[1, :, :, 1] = gate_P(2 * np.pi / 2.0 ** k)
    data.append(C)
    for j in range(2, k):
        W = _site(2, 2, ct)
        W[0, :, :, 0] = gate_I(ct)
        W[1, :, :, 1] = gate_I(ct)
        data.append(W)
        C = _site(2, 2, ct)
        C[0, :, :, 0] = gate_I(ct)
        C[1, :, :, 1] = gate_P(2 * np.pi / 2.0 ** (k - j + 1))
        data.append(C)
    W = _site(2, 2, ct)
    W[0, :, :, 0] = gate_I(ct)
    W[1, :, :, 1] = gate_I(ct)
    data.append(W)
    H = gate_H(ct)
    C = _site(2, 1, ct)
    for c in (0, 1):
        C[c, :, :, 0] = gate_Pi(c, ct) @ H           # Pi_c[s_in, tmp] H[tmp, s_out]
    data.append(C)
    return PairedSiteMPO(data, sm, sc)
