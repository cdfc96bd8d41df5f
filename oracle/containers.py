"""Index-free containers for the oracle (test infrastructure, see oracle/__init__.py).

Restates the reference's container semantics without ITensors' symbolic Index
objects: site identities are plain hashable labels (ints/strings) so the
reference's "same site indices" checks (apply.jl:76-85) have a counterpart.

  SignalMPS      src/mps.jl:70-79     (ctor/validator :121-146, :188-250)
  ZTMPS          src/mps.jl:98-117    (PairCore :37-55; validator :258-330)
  SingleSiteMPO  src/mpo.jl:26-44     (identity ctor :77-96)
  PairedSiteMPO  src/mpo.jl:57-74     (identity ctor :98-147)
"""
from __future__ import annotations

import numpy as np


def _check_chain(data, phys_rank):
    """Structural validation: neighbouring bond dims agree, edges are dim 1."""
    n = len(data)
    if n == 0:
        raise ValueError("empty tensor chain")
    for i, t in enumerate(data):
        if t.ndim != phys_rank + 2:
            raise ValueError(f"site {i+1}: expected rank {phys_rank + 2}, got {t.ndim}")
        if any(d != 2 for d in t.shape[1:-1]):
            raise ValueError(f"site {i+1}: physical dims must be 2, got {t.shape[1:-1]}")
    if data[0].shape[0] != 1 or data[-1].shape[-1] != 1:
        raise ValueError("edge bonds must have dimension 1")
    for i in range(n - 1):
        if data[i].shape[-1] != data[i + 1].shape[0]:
            raise ValueError(
                f"bond {i+1}: dims disagree ({data[i].shape[-1]} vs {data[i+1].shape[0]})")


class SignalMPS:
    """n-site MPS, tensors A[alpha, s, beta]; ``amplitude`` = ||x||_2 of the signal
    (src/mps.jl:70-79).  Tensor data are kept unit-norm by the encoders."""

    def __init__(self, data, sites=None, amplitude=1.0):
        self.data = [np.asarray(t) for t in data]
        _check_chain(self.data, 1)
        self.sites = list(range(1, len(self.data) + 1)) if sites is None else list(sites)
        if len(self.sites) != len(self.data):
            raise ValueError("sites/data length mismatch")
        self.amplitude = float(amplitude)

    def __len__(self):
        return len(self.data)

    @property
    def bond_dims(self):
        return [t.shape[-1] for t in self.data[:-1]]

    def copy(self):
        return SignalMPS([t.copy() for t in self.data], self.sites, self.amplitude)


class ZTMPS:
    """Paired-register MPS (src/mps.jl:98-117).  Stored directly in the interleaved
    2n-site chain main_1, copy_1, main_2, copy_2, ... which is what
    ``_as_signal_2n`` (src/mps.jl:421-444) produces: bonds[2i-1] = bonds_copy[i]
    (intra), bonds[2i] = bonds_main[i] (inter)."""

    def __init__(self, data2n, sites_main=None, sites_copy=None, amplitude=1.0):
        self.data = [np.asarray(t) for t in data2n]
        if len(self.data) % 2:
            raise ValueError("ZTMPS needs an even number of tensors (main/copy pairs)")
        _check_chain(self.data, 1)
        n = len(self.data) // 2
        self.sites_main = [("main", i) for i in range(1, n + 1)] if sites_main is None else list(sites_main)
        self.sites_copy = [("copy", i) for i in range(1, n + 1)] if sites_copy is None else list(sites_copy)
        self.amplitude = float(amplitude)

    def __len__(self):
        return len(self.data) // 2

    @property
    def sites(self):
        out = []
        for m, c in zip(self.sites_main, self.sites_copy):
            out += [m, c]
        return out

    @property
    def bonds_copy(self):
        return [self.data[2 * i].shape[-1] for i in range(len(self))]

    @property
    def bonds_main(self):
        return [self.data[2 * i + 1].shape[-1] for i in range(len(self) - 1)]

    def as_signal_2n(self):
        """src/mps.jl:421-444 (zero-copy relabel)."""
        return SignalMPS(self.data, self.sites, self.amplitude)

    @staticmethod
    def from_signal_2n(psi2n, sites_main=None, sites_copy=None):
        """src/mps.jl:447-472."""
        if sites_main is None:
            sites_main, sites_copy = psi2n.sites[0::2], psi2n.sites[1::2]
        return ZTMPS(psi2n.data, sites_main, sites_copy, psi2n.amplitude)


class SingleSiteMPO:
    """n-site MPO, tensors W[a, s_in, s_out, b] (src/mpo.jl:26-44)."""

    def __init__(self, data, sites=None):
        self.data = [np.asarray(t) for t in data]
        _check_chain(self.data, 2)
        self.sites = list(range(1, len(self.data) + 1)) if sites is None else list(sites)
        if len(self.sites) != len(self.data):
            raise ValueError("sites/data length mismatch")

    def __len__(self):
        return len(self.data)

    @property
    def bond_dims(self):
        return [t.shape[-1] for t in self.data[:-1]]

    @staticmethod
    def identity(n, sites=None, dtype=np.float64):
        """src/mpo.jl:77-96."""
        eye = np.eye(2, dtype=dtype).reshape(1, 2, 2, 1)
        return SingleSiteMPO([eye.copy() for _ in range(n)], sites)


class PairedSiteMPO:
    """2n-tensor MPO alternating main/copy sites (src/mpo.jl:57-74); stored in the
    interleaved order used by ``_as_single_site_mpo`` (src/linalg/apply.jl:16-32)."""

    def __init__(self, data2n, sites_main=None, sites_copy=None):
        self.data = [np.asarray(t) for t in data2n]
        if len(self.data) % 2:
            raise ValueError("PairedSiteMPO needs an even number of tensors")
        _check_chain(self.data, 2)
        n = len(self.data) // 2
        self.sites_main = [("main", i) for i in range(1, n + 1)] if sites_main is None else list(sites_main)
        self.sites_copy = [("copy", i) for i in range(1, n + 1)] if sites_copy is None else list(sites_copy)

    def __len__(self):
        return len(self.data) // 2

    @property
    def sites(self):
        out = []
        for m, c in zip(self.sites_main, self.sites_copy):
            out += [m, c]
        return out

    @property
    def bond_dims(self):
        return [t.shape[-1] for t in self.data[:-1]]

    def as_single_site_mpo(self):
        """src/linalg/apply.jl:16-32."""
        return SingleSiteMPO(self.data, self.sites)

    @staticmethod
    def from_single(W):
        """src/linalg/apply.jl:34-58."""
        if len(W) % 2:
            raise ValueError("_paired_from_single: length must be even")
        return PairedSiteMPO([t.copy() for t in W.data], W.sites[0::2], W.sites[1::2])

    @staticmethod
    def identity(n, dtype=np.float64):
        """src/mpo.jl:98-147."""
        eye = np.eye(2, dtype=dtype).reshape(1, 2, 2, 1)
        return PairedSiteMPO([eye.copy() for _ in range(2 * n)])
