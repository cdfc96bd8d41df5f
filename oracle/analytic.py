"""Closed-form oracles the reference's own tests compare against (test infrastructure).

  dft_unitary     test/test_qft_transformer.jl:6-20
  qn_matrix       test/test_qft_transformer.jl:23-34   (bit-reversed DFT)
  analytical_dt   test/test_dt_transformer.jl:71-92
  analytical_zt   test/test_zt_transformer.jl:20-39
"""
from __future__ import annotations

import numpy as np


def int_to_bits(v, n, order="msb"):
    bits = [(v >> (n - 1 - i)) & 1 for i in range(n)]
    return bits if order == "msb" else bits[::-1]


def bitrev(v, n):
    r = 0
    for i in range(n):
        r |= ((v >> i) & 1) << (n - 1 - i)
    return r


def dft_unitary(v):
    """|j> -> 1/sqrt(N) sum_k exp(-2 pi i j k / N) |k>."""
    v = np.asarray(v)
    N = len(v)
    jk = np.outer(np.arange(N), np.arange(N))
    return (np.exp(-2j * np.pi * jk / N) @ v) / np.sqrt(N)


def qn_matrix(n):
    """Q_n[j, k] = exp(-2 pi i bitrev(j) k / N) / sqrt(N)."""
    N = 2 ** n
    jr = np.array([bitrev(j, n) for j in range(N)])
    return np.exp(-2j * np.pi * np.outer(jr, np.arange(N)) / N) / np.sqrt(N)


def analytical_dt(vec, wr):
    """out[k] = (1/sqrt N) sum_j vec[j] exp(-wr k j / N)."""
    vec = np.asarray(vec)
    N = len(vec)
    kj = np.outer(np.arange(N), np.arange(N))
    return (np.exp(-wr * kj / N) @ vec.astype(np.complex128)) / np.sqrt(N)


def analytical_zt(x, wr=2 * np.pi, wi=2 * np.pi, dt=1.0, normalize=True):
    """chi[k, l] = (1/N) dt sum_j x[j] exp(-((wr k + i wi l)/N) j dt)."""
    x = np.asarray(x, dtype=np.complex128)
    N = len(x)
    k = np.arange(N)
    s = (wr * k[:, None] + 1j * wi * k[None, :]) / N           # s[k, l]
    E = np.exp(-s[:, :, None] * (np.arange(N) * dt)[None, None, :])
    return ((1.0 / N) if normalize else 1.0) * dt * (E @ x)
