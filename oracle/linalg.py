"""Truncated SVD / QR / randomized SVD for the oracle (test infrastructure).

The decompositions live in ITensors.jl / LAPACK, which are NOT in the reference
tree (Project.toml:9,18).  What is restated here is the published truncation
rule of ITensors' ``svd(...; cutoff, maxdim, mindim)`` [upstream-recall,
NDTensors ``truncate!``], pinned by the bond dimensions the reference's executed
tutorials print (docs/src/tutorials/signal.md:66-74, dft.md:71-75, 117-121,
dt.md:104-113, zt.md:185-190; see tests/test_oracle_pins.py):

  1. spectrum P = sigma^2 sorted descending;
  2. drop from the tail while the kept count exceeds ``maxdim``;
  3. keep dropping from the tail while (discarded weight + P[n]) <= cutoff * sum(P)
     and the kept count exceeds ``mindim``.

``rsvd`` follows src/linalg/rsvd.jl:38-121 step by step.  Julia's global RNG
stream (rsvd.jl:74-76) cannot be reproduced, so the Gaussian sketch is drawn
from numpy's PCG64 with the same seed semantics (one reseed per call).
"""
from __future__ import annotations

import numpy as np


def truncation_rank(s, cutoff=None, maxdim=None, mindim=1):
    """Number of singular values ITensors keeps for spectrum ``s`` (descending)."""
    s = np.asarray(s, dtype=np.float64)
    P = s * s
    n = len(P)
    if n == 0:
        return 0
    if P[0] <= 0.0:
        return 1
    if n == 1:
        return 1
    if maxdim is None:
        maxdim = n
    mindim = max(int(mindim), 1)
    truncerr = 0.0
    while n > maxdim:
        truncerr += P[n - 1]
        n -= 1
    if cutoff is not None:
        scale = P.sum()
        if scale == 0.0:
            scale = 1.0
        while n > mindim and truncerr + P[n - 1] <= cutoff * scale:
            truncerr += P[n - 1]
            n -= 1
    return max(n, 1)


def svd_trunc(M, cutoff=None, maxdim=None, mindim=1):
    """M = U diag(S) Vh truncated by the ITensors rule.  Returns (U, S, Vh)."""
    M = np.asarray(M)
    try:
        U, S, Vh = np.linalg.svd(M, full_matrices=False)
    except np.linalg.LinAlgError:  # pragma: no cover - gesdd fallback
        import scipy.linalg
        U, S, Vh = scipy.linalg.svd(M, full_matrices=False, lapack_driver="gesvd")
    r = truncation_rank(S, cutoff, maxdim, mindim)
    return U[:, :r], S[:r], Vh[:r, :]


def qr_positive(M):
    """Thin QR with a non-negative real diagonal of R (``qr(...; positive=true)``,
    src/linalg/rsvd.jl:83,90,94)."""
    Q, R = np.linalg.qr(M, mode="reduced")
    d = np.diagonal(R).copy()
    ph = np.where(d == 0, 1.0, d / np.where(d == 0, 1.0, np.abs(d)))
    Q = Q * ph[None, :]
    R = R * np.conj(ph)[:, None]
    return Q, R


def rsvd(A, k=20, p=10, q=0, random_seed=1234, cutoff=1e-15, maxdim=None, mindim=1):
    """Halko randomized SVD of the matrix ``A`` (m x n), src/linalg/rsvd.jl:38-121.

    Returns (U, S, Vh) with A ~= U diag(S) Vh.  ``maxdim`` defaults to ``k``
    (rsvd.jl:47).  Raises ValueError if either side is empty (rsvd.jl:56-60).
    """
    A = np.asarray(A)
    if A.ndim != 2 or A.shape[0] == 0 or A.shape[1] == 0:
        raise ValueError("rsvd: left or right index set is empty")
    if maxdim is None:
        maxdim = k
    m, n = A.shape
    l = min(k + p, m, n)                                     # rsvd.jl:72
    rng = np.random.default_rng(random_seed)                 # rsvd.jl:74 (own stream)
    if np.iscomplexobj(A):
        Om = (rng.standard_normal((n, l)) + 1j * rng.standard_normal((n, l))) / np.sqrt(2.0)
    else:
        Om = rng.standard_normal((n, l))                     # rsvd.jl:76
    Y = A @ Om                                               # rsvd.jl:79
    Q, _ = qr_positive(Y)                                    # rsvd.jl:83
    for _ in range(q):                                       # rsvd.jl:86-95
        Z = A.conj().T @ Q
        QZ, _ = qr_positive(Z)
        Y = A @ QZ
        Q, _ = qr_positive(Y)
    B = Q.conj().T @ A                                       # rsvd.jl:98
    Us, S, Vh = svd_trunc(B, cutoff=cutoff, maxdim=maxdim, mindim=mindim)  # :103-111
    U = Q @ Us                                               # rsvd.jl:114
    return U, S, Vh
