"""MPO x MPS and MPO x MPO apply (test infrastructure).

Restates src/linalg/apply.jl:
  apply(SingleSiteMPO, SignalMPS)  :75-122   (A1)
  apply(PairedSiteMPO, ZTMPS)      :201-218  (A2)
  apply(MPO, MPO)                  :124-199, :220-230 (A3)

The fused-bond ordering inside ``apply`` is internal to the reference (a
``combiner``, apply.jl:108) and never observable through ``coefficient``; the
oracle fixes it as  row = alpha + chi_l * a,  col = beta + chi_r * b  (MPS bond
fastest), which is also the layout the HIP kernel writes, so
site tensors can be compared element-wise between oracle and device.
"""
from __future__ import annotations

import numpy as np

from .containers import SignalMPS, ZTMPS, SingleSiteMPO, PairedSiteMPO


def apply_site(W, A):
    """B[(a,alpha), s, (b,beta)] = sum_{s'} W[a, s', s, b] * A[alpha, s', beta]
    (apply.jl:92-119: contraction :101 + the two combiner passes :114, :118).

    Computed the way the reference/NDTensors does it: one K=2 GEMM of the
    matricised operands, then a permutation into the fused layout."""
    Dl, _, _, Dr = W.shape
    cl, _, cr = A.shape
    Wm = np.ascontiguousarray(W.transpose(0, 2, 3, 1)).reshape(Dl * 2 * Dr, 2)   # (a,s,b | s')
    Am = np.ascontiguousarray(A.transpose(1, 0, 2)).reshape(2, cl * cr)          # (s' | alpha,beta)
    T = (Wm @ Am).reshape(Dl, 2, Dr, cl, cr)                                     # a s b alpha beta
    # fused index = alpha + chi * a  -> (a, alpha) with alpha fastest == C-order (a, alpha)
    return np.ascontiguousarray(T.transpose(0, 3, 1, 2, 4)).reshape(Dl * cl, 2, Dr * cr)


def apply(W, psi, **kwargs):
    """apply(W, psi; kwargs...) -> psi_out.  ``kwargs`` (cutoff, maxdim) are accepted
    and ignored exactly like the reference (apply.jl:75; no truncation in apply)."""
    if isinstance(W, PairedSiteMPO) and isinstance(psi, ZTMPS):
        if len(W.data) != 2 * len(psi.sites_main):                    # apply.jl:202-203
            raise ValueError("apply: MPO and MPS must have compatible sizes.")
        out2n = apply(W.as_single_site_mpo(), psi.as_signal_2n(), **kwargs)
        res = ZTMPS.from_signal_2n(out2n, psi.sites_main, psi.sites_copy)
        res.amplitude = psi.amplitude                                 # apply.jl:216
        return res
    if isinstance(W, (SingleSiteMPO, PairedSiteMPO)) and isinstance(psi, (SingleSiteMPO, PairedSiteMPO)):
        return apply_mpo_mpo(W, psi)
    if not (isinstance(W, SingleSiteMPO) and isinstance(psi, SignalMPS)):
        raise TypeError("apply: unsupported operand types")
    if len(W) != len(psi):                                            # apply.jl:76-80
        raise ValueError(
            f"apply: MPO and MPS must have the same number of sites. "
            f"Found length(W)={len(W)}, length(psi)={len(psi)}")
    if list(W.sites) != list(psi.sites):                              # apply.jl:81-85
        raise ValueError("apply: MPO and MPS must have the same site indices.")
    data = [apply_site(Wi, Ai) for Wi, Ai in zip(W.data, psi.data)]
    return SignalMPS(data, psi.sites, amplitude=psi.amplitude)        # apply.jl:121


def _compose_site(T1, T2):
    """W1 acts first, then W2: W1's output leg joins W2's input leg (apply.jl:163-171).
    out[(a1,a2), s_in, s_out, (b1,b2)] with the W1 bond fastest."""
    D1l, _, _, D1r = T1.shape
    D2l, _, _, D2r = T2.shape
    T = np.einsum("aimb,cmod->caiodb", T1, T2)       # (a2, a1, in, out, b2, b1)
    return T.reshape(D2l * D1l, 2, 2, D2r * D1r)


def apply_mpo_mpo(W1, W2):
    """Operator product "W1 first, then W2" over the overlapping site window
    (apply.jl:124-199); base = the longer MPO, non-overlapping sites copied."""
    paired = isinstance(W1, PairedSiteMPO)
    if paired != isinstance(W2, PairedSiteMPO):
        raise TypeError("apply: cannot mix SingleSiteMPO and PairedSiteMPO")
    if paired:                                                         # apply.jl:220-230
        return PairedSiteMPO.from_single(
            apply_mpo_mpo(W1.as_single_site_mpo(), W2.as_single_site_mpo()))
    s1, s2 = list(W1.sites), list(W2.sites)
    n1, n2 = len(s1), len(s2)
    start1 = next((i for i, s in enumerate(s1) if s in s2), None)      # apply.jl:129-131
    if start1 is None:
        raise ValueError("apply: No matching sites found")
    start2 = s2.index(s1[start1])
    match = 0
    while start1 + match < n1 and start2 + match < n2 and s1[start1 + match] == s2[start2 + match]:
        match += 1
    base, base_start = (W1, start1) if n1 >= n2 else (W2, start2)      # apply.jl:141-147
    new_data = [t.copy() for t in base.data]
    for i in range(match):
        T = _compose_site(W1.data[start1 + i], W2.data[start2 + i])
        new_data[base_start + i] = T
    # Outside the window the base MPO's bonds are untouched; inside, bonds are the
    # products.  At the window edges the non-base operand's bond has dimension 1
    # (it is an edge of that MPO) whenever the reference's output validates.
    return SingleSiteMPO(new_data, base.sites)
