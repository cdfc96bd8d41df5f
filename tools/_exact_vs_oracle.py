"""Exact compress!(apply) of the bench's bond-1008 product on the GPU against the CPU oracle's compress! of the same downloaded
product (256 sampled coefficients): python tools/_exact_vs_oracle.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qilaplace_jl_amd as qil
import oracle as O
n, N = 24, 2 ** 24
j = np.arange(N, dtype=np.float64)
x = np.sin(2 * np.pi * 5.0 * j / N) * np.exp(-3.0 * j / N) + 0.5 * np.cos(2 * np.pi * 11.0 * j / N)
rng = np.random.default_rng(1001)
x = x + sum(0.1 * rng.random() * np.sin(40.0 * (rng.random() - 0.5) * j / N) for _ in range(6))
psi = qil.signal_ztmps(x, method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
W = qil.build_zt_mpo(psi, 2 * np.pi)
prod = W * psi
bits = np.random.default_rng(3).integers(0, 2, size=(256, 2 * n)).astype(np.uint8)
c_x = qil.coefficient_batch(prod, bits)
ph = O.ZTMPS([t.copy() for t in prod.to_host()], amplitude=prod.amplitude) if hasattr(O, "ZTMPS") else None
t0 = time.perf_counter(); qil.compress(prod, maxdim=64, tol=1e-8); qil.default_context().synchronize(); t1 = time.perf_counter()
O.compress(ph, maxdim=64, tol=1e-8)
c_e, c_c = qil.coefficient_batch(prod, bits), O.coefficient_batch(ph, bits)
s = np.abs(c_x).max()
print("exact route %.1f ms; bonds hip %d cpu %d; |hip - cpu| %.2e, |hip - product| %.2e, |cpu - product| %.2e (of the largest coefficient)" % (
    (t1 - t0) * 1e3, max(prod.bond_dims), max(t.shape[-1] for t in ph.data), np.abs(c_e - c_c).max() / s, np.abs(c_e - c_x).max() / s, np.abs(c_c - c_x).max() / s))
