"""On-disk interchange of MPS / MPO objects.

The reference has no serialisation beyond JLD2 benchmark artefacts (scripts/benchmark/common.jl:193-212).
This flat ``.npz`` layout lets a Julia user with real ITensors objects hand data to the library (NPZ.jl
writes the same container) while no Julia runtime is available on the GPU box:

    kind        "SignalMPS" | "ZTMPS" | "SingleSiteMPO" | "PairedSiteMPO"
    amplitude   float64 scalar (MPS only)
    site_ids    int64[n_tensors]
    site_%04d   the site tensors in index order A[alpha, s, beta] / W[a, s_in, s_out, b]
                (s_in = the reference's primed leg), float64 or complex128, explicit dim-1 edge bonds
"""
from __future__ import annotations

import numpy as np

from .containers import SignalMPS, ZTMPS, SingleSiteMPO, PairedSiteMPO

_KINDS = {"SignalMPS": SignalMPS, "ZTMPS": ZTMPS, "SingleSiteMPO": SingleSiteMPO, "PairedSiteMPO": PairedSiteMPO}


def save(path, obj):
    """Write a device MPS/MPO to ``path`` (.npz)."""
    kind = type(obj).__name__
    if kind not in _KINDS:
        raise TypeError(f"cannot save objects of type {kind}")
    sites = obj.to_host()
    payload = {"kind": np.array(kind), "site_ids": np.asarray(obj.site_ids, dtype=np.int64)}
    if isinstance(obj, SignalMPS):
        payload["amplitude"] = np.float64(obj.amplitude)
    for i, t in enumerate(sites):
        payload[f"site_{i:04d}"] = np.asarray(t)
    np.savez(path, **payload)


def load(path, ctx=None):
    """Read an object written by :func:`save` (or by NPZ.jl following the same layout) onto the device."""
    with np.load(path, allow_pickle=False) as z:
        kind = str(z["kind"])
        if kind not in _KINDS:
            raise ValueError(f"unknown kind {kind!r} in {path}")
        n = sum(1 for k in z.files if k.startswith("site_") and k != "site_ids")
        sites = [z[f"site_{i:04d}"] for i in range(n)]
        ids = [int(v) for v in z["site_ids"]] if "site_ids" in z.files else None
        cls = _KINDS[kind]
        if issubclass(cls, SignalMPS):
            return cls(sites, sites=ids, amplitude=float(z["amplitude"]), ctx=ctx)
        return cls(sites, sites=ids, ctx=ctx)
