"""Signal -> MPS encoders (test infrastructure).  Restates
src/signals/SignalConverters.jl:

  _array_to_tensor      :16-46
  _tensor_to_mps_svd    :49-104   (signal_mps method=:svd,  E1)
  _tensor_to_mps_rsvd   :107-196  (signal_mps method=:rsvd, E2; recursion compress_tt! :145-184)
  signal_mps            :228-233
  signal_ztmps          :247-283  (E4)
"""
from __future__ import annotations

import warnings

import numpy as np

from .containers import SignalMPS, ZTMPS
from .linalg import svd_trunc, rsvd


def array_to_tensor(x):
    """Zero-pad to 2^n (n = round(log2 N)), normalise; returns (x_hat, amplitude, n).
    Site 1 is the MOST significant bit of the sample index (:39-41), i.e. x_hat viewed
    as a C-ordered (2,)*n array has axis 0 = site 1."""
    x = np.asarray(x)
    N = len(x)
    n = int(round(np.log2(N)))
    if N < 2 ** n:
        warnings.warn(f"_array_to_tensor: input length {N} is not a power of 2; zero-filling to {2**n}")
        xf = np.zeros(2 ** n, dtype=x.dtype)
        xf[:N] = x
        x = xf
    if len(x) != 2 ** n:
        raise ValueError("_array_to_tensor: length of signal vector must be a power of 2")
    amp = float(np.linalg.norm(x))
    x = x / amp
    if not np.iscomplexobj(x):
        x = x.astype(np.float64)
    return x, amp, n


def _tensor_to_mps_svd(xh, n, cutoff=1e-15, maxdim=None):
    """Sequential SVD sweep (:77-98): step i splits (bond_{i-1}, site_i | rest)."""
    if n == 1:
        return [xh.reshape(1, 2, 1)]
    data = []
    cur = xh.reshape(1, -1)                                    # (bond_{i-1} | s_i ... s_n)
    for i in range(n - 1):
        r = cur.shape[0]
        M = cur.reshape(r * 2, -1)
        U, S, Vh = svd_trunc(M, cutoff=cutoff, maxdim=maxdim)
        k = len(S)
        data.append(U.reshape(r, 2, k))
        cur = S[:, None] * Vh
    data.append(cur.reshape(cur.shape[0], 2, 1))
    return data


def _tensor_to_mps_rsvd(xh, n, cutoff=1e-15, maxdim=None, **kwargs):
    """Divide-and-conquer RSVD (:107-196).  ``cutoff``/``maxdim`` override same-named
    kwargs (:133); maxdim=None (typemax(Int) in the reference) leaves the rank capped
    by the sketch width l = k + p only."""
    if n == 1:
        return [xh.reshape(1, 2, 1)]
    kw = dict(kwargs)
    kw["cutoff"] = cutoff
    kw["maxdim"] = maxdim if maxdim is not None else np.iinfo(np.int64).max
    data = [None] * n

    def compress_tt(T, first, last):
        # T has shape (lb, 2^(last-first+1), rb); sites first..last (0-based, inclusive)
        if first == last:
            data[first] = T.reshape(T.shape[0], 2, T.shape[2])
            return
        # reference (1-based): mid = (first + last - 1) div 2   (:161)
        mid = (first + last + 1) // 2 - 1
        nl = mid - first + 1
        lb, rb = T.shape[0], T.shape[2]
        M = T.reshape(lb * 2 ** nl, -1)
        U, S, Vh = rsvd(M, **kw)
        k = len(S)
        compress_tt(U.reshape(lb, 2 ** nl, k), first, mid)
        compress_tt((S[:, None] * Vh).reshape(k, -1, rb), mid + 1, last)

    compress_tt(xh.reshape(1, -1, 1), 0, n - 1)
    return data


def signal_mps(x, method="svd", **kwargs):
    """signal_mps(x; method=:svd, kwargs...) (:228-233)."""
    if method not in ("svd", "rsvd"):
        raise ValueError(f"tensor_to_mps: unknown method {method}. Use :svd or :rsvd.")
    xh, amp, n = array_to_tensor(x)
    if method == "svd":
        data = _tensor_to_mps_svd(xh, n, **kwargs)
    else:
        data = _tensor_to_mps_rsvd(xh, n, **kwargs)
    return SignalMPS(data, amplitude=amp)


def signal_ztmps(x, cutoff=1e-10, maxdim=None, **kwargs):
    """signal_ztmps(x; cutoff=1e-10, maxdim, kwargs...) (:247-283): per site fuse
    delta(s, s_main, s_copy) and split (bond_{i-1}, s_main | s_copy, bond_i) by SVD."""
    psi = signal_mps(x, cutoff=cutoff, maxdim=maxdim, **kwargs)
    data2n = []
    for A in psi.data:
        cl, _, cr = A.shape
        T = np.zeros((cl, 2, 2, cr), dtype=A.dtype)            # (b_{i-1}, s_main, s_copy, b_i)
        T[:, 0, 0, :] = A[:, 0, :]
        T[:, 1, 1, :] = A[:, 1, :]
        U, S, Vh = svd_trunc(T.reshape(cl * 2, 2 * cr), cutoff=cutoff, maxdim=maxdim)
        c = len(S)
        data2n.append(U.reshape(cl, 2, c))
        data2n.append((S[:, None] * Vh).reshape(c, 2, cr))
    return ZTMPS(data2n, amplitude=psi.amplitude)
