"""MPS algebra on the hot path (test infrastructure): coefficient, mps_to_vector,
norm, canonicalize!, compress!.  Restates src/mps.jl:609-999.
"""
from __future__ import annotations

import re

import numpy as np

from .containers import SignalMPS, ZTMPS
from .linalg import svd_trunc


def _chain(psi):
    return psi.as_signal_2n() if isinstance(psi, ZTMPS) else psi


# ---------------------------------------------------------------- coefficient (C1)
def parse_config(spec, n):
    """All front-ends of ``coefficient`` (src/mps.jl:616-645, 680-693) -> list of bits.

    * list/tuple of ints      -> as is
    * "1010"                  -> bit string
    * "[1,0,1]" / "1 0 1"     -> separator string
    * non-negative int        -> n-bit big-endian (mps.jl:633-645)
    """
    if isinstance(spec, str):
        s = spec.strip().strip("[](){}").strip()
        if not s:
            raise ValueError("coefficient: configuration string is empty")
        if re.search(r"[,\s]", s):
            toks = [t for t in re.split(r"[,\s]+", s) if t]
            if not toks:
                raise ValueError("coefficient: configuration string did not contain any entries")
            return [int(t) for t in toks]
        if any(c not in "01" for c in s):
            raise ValueError("coefficient: bit strings may contain only '0' or '1'")
        return [1 if c == "1" else 0 for c in s]
    if isinstance(spec, (int, np.integer)) and not isinstance(spec, bool):
        v = int(spec)
        if v < 0:
            raise ValueError("coefficient: integer configuration must be non-negative")
        bits = [(v >> (n - 1 - i)) & 1 for i in range(n)]
        if v >> n:
            raise ValueError(f"coefficient: integer {v} requires more than {n} bits")
        return bits
    return [int(b) for b in spec]


def coefficient(psi, config):
    """amplitude * prod_i A_i[:, cfg_i, :] (src/mps.jl:669-678); cfg[0] <-> site 1."""
    chain = _chain(psi)
    N = len(chain.data)
    bits = parse_config(config, N)
    if len(bits) != N:
        raise ValueError(f"coefficient: expected {N} entries, got {len(bits)}")
    for b in bits:
        if not 0 <= b < 2:
            raise ValueError(f"coefficient: bit value {b} outside [0,1]")
    v = chain.data[0][:, bits[0], :]
    for i in range(1, N):
        # the reference contracts the whole site tensor and projects afterwards
        # (mps.jl:675); selecting the slice first is the same arithmetic.
        v = v @ chain.data[i][:, bits[i], :]
    return chain.amplitude * v[0, 0]


def coefficient_batch(psi, bits):
    """Vectorised ``coefficient`` over a (nb, N) array of bits -> complex/real (nb,)."""
    chain = _chain(psi)
    bits = np.asarray(bits)
    nb, N = bits.shape
    if N != len(chain.data):
        raise ValueError(f"coefficient: expected {len(chain.data)} entries, got {N}")
    if bits.min(initial=0) < 0 or bits.max(initial=0) > 1:
        raise ValueError("coefficient: bit value outside [0,1]")
    dt = np.result_type(*[t.dtype for t in chain.data])
    v = np.ones((nb, 1), dtype=dt)
    for i in range(N):
        A = chain.data[i]
        nv = np.empty((nb, A.shape[2]), dtype=dt)
        for b in (0, 1):
            sel = bits[:, i] == b
            if sel.any():
                nv[sel] = v[sel] @ A[:, b, :]
        v = nv
    return chain.amplitude * v[:, 0]


def lazy_coefficient_batch(W, psi, bits):
    """<bits| W psi> WITHOUT materialising W*psi: carries a (D x chi) matrix per
    query.  Same numbers as coefficient_batch(apply(W, psi), bits); used as the
    checker when W*psi is too large for the host."""
    Wd = W.as_single_site_mpo().data if hasattr(W, "as_single_site_mpo") else W.data
    chain = _chain(psi)
    bits = np.asarray(bits)
    nb, N = bits.shape
    dt = np.result_type(Wd[0].dtype, chain.data[0].dtype)
    M = np.ones((nb, 1, 1), dtype=dt)                          # M[q, a, alpha]
    for i in range(N):
        Wi, Ai = Wd[i], chain.data[i]
        D, D2, c, c2 = Wi.shape[0], Wi.shape[3], Ai.shape[0], Ai.shape[2]
        nM = np.empty((nb, D2, c2), dtype=dt)
        A2 = np.ascontiguousarray(Ai).reshape(c, 2 * c2).astype(dt, copy=False)          # [alpha, (sp, beta)]
        for b in (0, 1):
            sel = bits[:, i] == b
            ns = int(sel.sum())
            if ns == 0:
                continue
            # sum_{a,sp,alpha} W[a,sp,b,a'] M[q,a,alpha] A[alpha,sp,beta] as two LARGE GEMMs over all selected queries
            # (a loop of per-query products is the same arithmetic one query at a time)
            X = (M[sel].reshape(ns * D, c) @ A2).reshape(ns, D * 2, c2)                   # [q, (a, sp), beta]
            Wb = np.ascontiguousarray(Wi[:, :, b, :]).reshape(D * 2, D2)                  # [(a, sp), a']
            Y = Wb.T @ np.ascontiguousarray(X.transpose(1, 0, 2)).reshape(D * 2, ns * c2)   # [a', (q, beta)]
            nM[sel] = Y.reshape(D2, ns, c2).transpose(1, 0, 2)
        M = nM
    return chain.amplitude * M[:, 0, 0]


# ---------------------------------------------------------------- mps_to_vector (C2)
def mps_to_vector(psi, reverse=False):
    """src/mps.jl:716-743.  reverse=False: index j = sum_i b_i 2^(n-i) (site 1 = MSB);
    reverse=True: site 1 = LSB (FFT order for a QFT output)."""
    chain = _chain(psi)
    T = chain.data[0][0]                                       # (s1, beta)
    for A in chain.data[1:]:
        T = np.tensordot(T, A, axes=([-1], [0]))
    T = T[..., 0]                                              # (s1, ..., sn)
    n = len(chain.data)
    if reverse:
        T = T.transpose(tuple(range(n - 1, -1, -1)))
    return np.ascontiguousarray(T).reshape(-1) * chain.amplitude


# ---------------------------------------------------------------- norm (K3)
def norm(psi):
    """sqrt(|<psi|psi>|) by transfer-matrix contraction, WITHOUT amplitude (mps.jl:754-771)."""
    chain = _chain(psi)
    E = np.ones((1, 1), dtype=np.result_type(chain.data[0].dtype, np.float64))
    for A in chain.data:
        t = np.tensordot(E, A, axes=([0], [0]))                # (alpha', s, beta)
        E = np.tensordot(A.conj(), t, axes=([0, 1], [0, 1]))   # (beta', beta)  -> conj side first
        E = E.T
    return float(np.sqrt(abs(E[0, 0])))


# ---------------------------------------------------------------- canonicalize! (K2)
def canonicalize(psi, direction, center=None, cutoff=1e-12, maxdim=None):
    """In-place gauge sweep (src/mps.jl:787-847; ZTMPS :866-901).  Because a ``cutoff``
    is always passed, ITensors' ``factorize`` takes its SVD branch [upstream-recall],
    so the sweep truncates at relative weight ``cutoff``."""
    if direction not in ("right", "left"):
        raise ValueError("Direction must be :right or :left")
    chain = _chain(psi)
    d = chain.data
    N = len(d)
    if direction == "right":
        c = N if center is None else center
        if not 1 <= c <= N:
            raise ArithmeticError(f"Center out of range [1,{N}]")   # DomainError
        for n in range(c - 1):
            cl, _, cr = d[n].shape
            U, S, Vh = svd_trunc(d[n].reshape(cl * 2, cr), cutoff=cutoff, maxdim=maxdim)
            r = len(S)
            d[n] = U.reshape(cl, 2, r)                               # ortho="left": L = U
            d[n + 1] = np.tensordot(S[:, None] * Vh, d[n + 1], axes=([1], [0]))
    else:
        c = 1 if center is None else center
        if not 1 <= c <= N:
            raise ArithmeticError(f"Center out of range [1,{N}]")
        for n in range(N - 1, c - 1, -1):
            cl, _, cr = d[n].shape
            U, S, Vh = svd_trunc(d[n].reshape(cl, 2 * cr), cutoff=cutoff, maxdim=maxdim)
            r = len(S)
            d[n] = Vh.reshape(r, 2, cr)                              # ortho="right": R = V
            d[n - 1] = np.tensordot(d[n - 1], U * S[None, :], axes=([2], [0]))
    if isinstance(psi, ZTMPS):
        psi.data = d
    return psi


# ---------------------------------------------------------------- compress! (K1)
def compress(psi, maxdim=None, tol=1e-12, sweeps=1):
    """In-place two-site SVD compression (src/mps.jl:913-973; ZTMPS :975-999)."""
    chain = _chain(psi)
    d = chain.data
    N = len(d)
    if N < 2:
        raise ArithmeticError("SignalMPS must have at least 2 sites.")   # DomainError mps.jl:918
    cutoff = tol ** 2 / ((N - 1) * sweeps)                               # mps.jl:920
    canonicalize(chain, "left")                                          # mps.jl:923
    for _ in range(sweeps):
        for j in range(N - 1):                                           # L -> R, mps.jl:927-942
            cl, cr = d[j].shape[0], d[j + 1].shape[2]
            theta = np.tensordot(d[j], d[j + 1], axes=([2], [0])).reshape(cl * 2, 2 * cr)
            U, S, Vh = svd_trunc(theta, cutoff=cutoff, maxdim=maxdim)
            r = len(S)
            d[j] = U.reshape(cl, 2, r)
            d[j + 1] = (S[:, None] * Vh).reshape(r, 2, cr)
        for j in range(N - 2, -1, -1):                                   # R -> L, mps.jl:944-959
            cl, cr = d[j].shape[0], d[j + 1].shape[2]
            theta = np.tensordot(d[j], d[j + 1], axes=([2], [0])).reshape(cl * 2, 2 * cr)
            U, S, Vh = svd_trunc(theta, cutoff=cutoff, maxdim=maxdim)
            r = len(S)
            d[j] = (U * S[None, :]).reshape(cl, 2, r)
            d[j + 1] = Vh.reshape(r, 2, cr)
    canonicalize(chain, "left")                                          # mps.jl:963
    nrm = norm(chain)                                                    # mps.jl:967-971
    if nrm != 0:
        chain.amplitude *= nrm
        d[0] = d[0] * (1.0 / nrm)
    if isinstance(psi, ZTMPS):
        psi.data = d
        psi.amplitude = chain.amplitude
    return psi
