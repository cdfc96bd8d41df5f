"""Transform-MPO builders (test infrastructure) -- the producers of the hot path's
operand ``W``.  Index-free restatement of

  build_qft_mpo  src/transforms/qft_transformer.jl:121-165 (zip_up_mpos :13-66, zip_down_mpos :69-101)
  build_dt_mpo   src/transforms/dt_transformer.jl:312-412  (zip_to_combine_mpos :20-164,
                                                            zip_to_compress_mpo :167-288)
  build_zt_mpo   src/transforms/zt_transformer.jl:41-112

Only gauge-invariant outcomes (dense operator, bond dimensions) are pinned; the
QR/SVD gauges are LAPACK's, not ITensors'.
"""
from __future__ import annotations

import numpy as np

from .containers import SingleSiteMPO, PairedSiteMPO
from .linalg import svd_trunc
from .gates import (control_Hphase_mpo, control_damping_mpo, control_damping_copy_mpo,
                    control_Hphase_ztmps_mpo)
from .apply import apply_mpo_mpo


# ================================================================ QFT
def _zip_up(M, B):
    """zip_up_mpos (qft_transformer.jl:13-66).  ``B`` (shorter) acts AFTER ``M`` and is
    aligned with M's last sites.  No truncation: QR-type factorize, V right-orthogonal."""
    L1, L2 = len(M), len(B)
    if not L1 > L2:
        raise ValueError("zip_up_mpos: mpo1 must be longer than mpo2")
    new = list(M)
    T = np.ones((1, 1, 1), dtype=np.complex128)                # T[b1, b2, r]
    for r in range(L2):
        i, j = L1 - 1 - r, L2 - 1 - r
        # C[a1, a2, in, out, r] = M[a1,in,mid,b1] B[a2,mid,out,b2] T[b1,b2,r]   (:42)
        C = np.einsum("aimb,cmod,bdr->acior", M[i], B[j], T)
        a1, a2 = C.shape[0], C.shape[1]
        rr = C.shape[4]
        Cm = C.reshape(a1 * a2, 4 * rr)
        Q, R = np.linalg.qr(Cm.T, mode="reduced")              # Cm = R^T Q^T, Q^T rows orthonormal
        nb = Q.shape[1]
        new[i] = Q.T.reshape(nb, 2, 2, rr)                     # V  (:56)
        T = R.T.reshape(a1, a2, nb)                            # remainder towards the left
    k = L1 - L2 - 1
    # last remainder absorbed into the first untouched site (:63); a2 has dim 1 there
    new[k] = np.einsum("aiob,bcr->aior", new[k], T)
    return new, L1 - L2                                        # (data, oc 1-based)


def _zip_down(M, oc, cutoff, maxdim):
    """zip_down_mpos (qft_transformer.jl:69-101): truncating SVD sweep oc .. L-1."""
    new = list(M)
    L = len(new)
    for k in range(oc - 1, L - 1):
        a, _, _, b = new[k].shape
        U, S, Vh = svd_trunc(new[k].reshape(a * 4, b), cutoff=cutoff, maxdim=maxdim)
        r = len(S)
        new[k] = U.reshape(a, 2, 2, r)
        new[k + 1] = np.tensordot(S[:, None] * Vh, new[k + 1], axes=([1], [0]))
    return new


def build_qft_mpo(n, sites=None, cutoff=1e-14, maxdim=1000):
    """build_qft_mpo(n, sites; cutoff=1e-14, maxdim=1000) (qft_transformer.jl:121-160)."""
    if n < 1:
        raise ValueError(f"build_qft_mpo: n must be at least 1, got {n}")
    if sites is not None and len(sites) != n:
        raise ValueError("build_qft_mpo: number of sites must equal n")
    if n == 1:
        return control_Hphase_mpo(1, sites)
    M = list(control_Hphase_mpo(n).data)
    for it in range(1, n):
        B = control_Hphase_mpo(n - it).data                    # sites it+1..n, acts after M (:141-153)
        M, oc = _zip_up(M, B)
        M = _zip_down(M, oc, cutoff, maxdim)
    return SingleSiteMPO(M, sites)


# ================================================================ DT
def _combine_down(M, B):
    """dt_transformer.jl:38-95 (aligned at the first site)."""
    new = list(M)
    L1, L2 = len(M), len(B)
    dt = np.result_type(M[0].dtype, B[0].dtype)
    T = np.ones((1, 1, 1), dtype=dt)                           # T[r, a1, a2]
    for k in range(L2):
        # core[r, in, out, b1, b2] = T[r,a1,a2] M[a1,in,mid,b1] B[a2,mid,out,b2]
        core = np.einsum("rac,aimb,cmod->riobd", T, M[k], B[k])
        r, b1, b2 = core.shape[0], core.shape[3], core.shape[4]
        Q, R = np.linalg.qr(core.reshape(r * 4, b1 * b2), mode="reduced")   # (:65-75)
        nb = Q.shape[1]
        new[k] = Q.reshape(r, 2, 2, nb)
        T = R.reshape(nb, b1, b2)
    # B has ended, so b2 has dimension 1: T[new, b1, 1]
    if L1 > L2:                                                # absorb remainder (:90-95)
        new[L2] = np.einsum("nb,bioc->nioc", T[:, :, 0], new[L2])
    else:
        new[L2 - 1] = np.einsum("rion,nb->riob", new[L2 - 1], T[:, :, 0])
    return new


def _combine_up(M, B):
    """dt_transformer.jl:97-153 (aligned at the last site)."""
    new = list(M)
    L1, L2 = len(M), len(B)
    dt = np.result_type(M[0].dtype, B[0].dtype)
    T = np.ones((1, 1, 1), dtype=dt)                           # T[b1, b2, r]
    for k in range(L2):
        i1, i2 = L1 - 1 - k, L2 - 1 - k
        # core[a1, a2, in, out, r]
        core = np.einsum("aimb,cmod,bdr->acior", M[i1], B[i2], T)
        a1, a2, r = core.shape[0], core.shape[1], core.shape[4]
        Q, R = np.linalg.qr(core.reshape(a1 * a2, 4 * r).T, mode="reduced")   # rows = (in,out,r)
        nb = Q.shape[1]
        new[i1] = Q.T.reshape(nb, 2, 2, r)
        T = R.T.reshape(a1, a2, nb)
    # B has started, so a2 has dimension 1: T[a1, 1, new]
    if L1 > L2:                                                # (:148-153)
        j = L1 - L2 - 1
        new[j] = np.einsum("aiob,bn->aion", new[j], T[:, 0, :])
    else:
        new[0] = np.einsum("an,nior->aior", T[:, 0, :], new[0])
    return new


def _compress(M, direction, cutoff, maxdim):
    """zip_to_compress_mpo over the whole chain (dt_transformer.jl:167-288)."""
    new = list(M)
    L = len(new)
    if L < 2:
        return new
    if direction == "down":
        for i in range(L - 1):                                 # QR gauge sweep L->R (:186-203)
            a, _, _, b = new[i].shape
            Q, R = np.linalg.qr(new[i].reshape(a * 4, b), mode="reduced")
            new[i] = Q.reshape(a, 2, 2, Q.shape[1])
            new[i + 1] = np.tensordot(R, new[i + 1], axes=([1], [0]))
        for i in range(L - 1, 0, -1):                          # truncating SVD sweep R->L (:207-229)
            a0 = new[i - 1].shape[0]
            b1 = new[i].shape[3]
            core = np.tensordot(new[i - 1], new[i], axes=([3], [0]))   # (a0,i,o, i',o',b1)
            U, S, Vh = svd_trunc(core.reshape(a0 * 4, 4 * b1), cutoff=cutoff, maxdim=maxdim)
            r = len(S)
            new[i] = Vh.reshape(r, 2, 2, b1)                   # "U" of the reference (right_inds side)
            new[i - 1] = (U * S[None, :]).reshape(a0, 2, 2, r)
    elif direction == "up":
        for i in range(L - 1, 0, -1):                          # QR gauge sweep R->L (:233-251)
            a, _, _, b = new[i].shape
            Q, R = np.linalg.qr(new[i].reshape(a, 4 * b).T, mode="reduced")
            new[i] = Q.T.reshape(Q.shape[1], 2, 2, b)
            new[i - 1] = np.tensordot(new[i - 1], R.T, axes=([3], [0]))
        for i in range(L - 1):                                 # truncating SVD sweep L->R (:255-276)
            a0 = new[i].shape[0]
            b1 = new[i + 1].shape[3]
            core = np.tensordot(new[i], new[i + 1], axes=([3], [0]))
            U, S, Vh = svd_trunc(core.reshape(a0 * 4, 4 * b1), cutoff=cutoff, maxdim=maxdim)
            r = len(S)
            new[i] = U.reshape(a0, 2, 2, r)
            new[i + 1] = (S[:, None] * Vh).reshape(r, 2, 2, b1)
    else:
        raise ValueError(f"zip_to_compress_mpo: unknown direction '{direction}'")
    return new


def _extend_identity(M, dtype):
    """Append identity tensors for one more (main, copy) pair with fresh dim-1 bonds
    (dt_transformer.jl:354-380; zt_transformer.jl:81-95)."""
    eye = np.eye(2, dtype=dtype).reshape(1, 2, 2, 1)
    return list(M) + [eye.copy(), eye.copy()]


def build_dt_mpo(n, wr, sites_main=None, sites_copy=None, cutoff=1e-14, maxdim=1000):
    """build_dt_mpo(n, wr, sites_main, sites_copy; cutoff=1e-14, maxdim=1000)
    (dt_transformer.jl:312-407)."""
    if n < 1:
        raise ValueError(f"build_dt_mpo: n must be >= 1, got {n}")
    for s in (sites_main, sites_copy):
        if s is not None and len(s) != n:
            raise ValueError(f"build_dt_mpo: site lists must have {n} elements")
    if n == 1:
        return PairedSiteMPO(control_damping_mpo(1, 1, wr).data, sites_main, sites_copy)
    M = list(control_damping_mpo(n, 1, wr).data)                       # :348
    for k in range(2, n + 1):                                          # part 1, "down"
        M = _extend_identity(M, np.float64)
        B = control_damping_mpo(n, k, wr).data                         # :383
        M = _combine_down(M, B)
        M = _compress(M, "down", cutoff, maxdim)                       # :389
    for k in range(1, n):                                              # part 2, "up" (:396-405)
        B = control_damping_copy_mpo(n, k, wr).data
        M = _combine_up(M, B)
        M = _compress(M, "up", cutoff, maxdim)
    return PairedSiteMPO(M, sites_main, sites_copy)


def build_zt_mpo(n, wr, sites_main=None, sites_copy=None, cutoff=1e-14, maxdim=1000):
    """build_zt_mpo (zt_transformer.jl:41-106): DT first, then the paired QFT, fused once."""
    if n < 1:
        raise ValueError(f"build_zt_mpo: n must be >= 1, got {n}")
    W_dt = build_dt_mpo(n, wr, cutoff=cutoff, maxdim=maxdim)
    if n == 1:                                                         # :68-72
        out = apply_mpo_mpo(W_dt, control_Hphase_ztmps_mpo(1))
        return PairedSiteMPO(out.data, sites_main, sites_copy)
    Q = list(control_Hphase_ztmps_mpo(1).data)                         # :78
    for k in range(2, n + 1):
        Q = _extend_identity(Q, np.complex128)
        B = control_Hphase_ztmps_mpo(k).data                           # :96
        Q = _combine_down(Q, B)
        Q = _compress(Q, "down", cutoff, maxdim)                       # :97-98
    W = apply_mpo_mpo(W_dt, PairedSiteMPO(Q))                          # :103
    data = _compress(list(W.data), "down", cutoff, maxdim)             # :104
    return PairedSiteMPO(data, sites_main, sites_copy)
