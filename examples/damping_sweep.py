#!/usr/bin/env python3
"""A damping sweep (BASELINE.json configs[3]; the loop of docs/src/tutorials/dt.jl:150-197 and zt.jl:300-348 of QILaplace.jl)
with the batch entry points: every damping value's DT MPO is built in ONE launch, applied and sampled concurrently, and the
truncated products (apply-and-truncate per value) come from one batch call.

    python examples/damping_sweep.py

The reference runs `W = build_dt_mpo(psi, wr); out = W * psi; coefficient(out, ...)` once per value, one after another."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil  # noqa: E402


def main():
    n = 12
    N = 2 ** n
    j = np.arange(N)
    signal = np.sin(2 * np.pi * 3.0 * j / N) * np.exp(-2.0 * j / N) + 0.3 * np.cos(2 * np.pi * 7.0 * j / N)
    psi = qil.signal_ztmps(signal, method="svd", cutoff=1e-20)
    sigmas = np.linspace(0.25, 8.0, 16)

    # 1. the sweep body of the reference, batched: 16 operators in one launch, 64 samples of each product
    rng = np.random.default_rng(5)
    kk, jj = rng.integers(0, N, 64), rng.integers(0, N, 64)
    lsb = lambda v: [(int(v) >> i) & 1 for i in range(n)]
    msb = lambda v: [(int(v) >> (n - 1 - i)) & 1 for i in range(n)]
    bits = np.array([[b for pair in zip(lsb(k), msb(q)) for b in pair] for k, q in zip(kk, jj)], dtype=np.uint8)
    coeffs = qil.damping_sweep(psi, sigmas, bits)                       # (16, 64)
    closed = np.array([signal[jj] * np.exp(-s * kk * jj / N) / np.sqrt(N) for s in sigmas])
    err = np.abs(coeffs - closed).max() / np.abs(closed).max()
    print(f"damping sweep: {len(sigmas)} values x {len(kk)} samples, max error vs x_j e^(-s k j / N) / sqrt(N): {err:.2e}")

    # 2. apply-and-truncate for every value in one call (the products stay on the device, bond <= 32)
    Ws = qil.build_dt_mpo_batch(psi, sigmas)
    outs = qil.apply_compress_batch(Ws, psi, maxdim=32, tol=1e-9)
    one = qil.apply_compress(Ws[3], psi, maxdim=32, tol=1e-9)
    same = all(np.array_equal(a, b) for a, b in zip(one.to_host(), outs[3].to_host()))
    err_t = max(np.abs(qil.coefficient_batch(o, bits) - c).max() for o, c in zip(outs, closed)) / np.abs(closed).max()
    print(f"apply_compress_batch: bonds {max(max(o.bond_dims) for o in outs)}, item 3 identical to the single call: {same}, "
          f"max error after truncation {err_t:.2e}")

    # 3. independent chains compressed together
    chains = [W * psi for W in Ws]
    qil.compress_batch(chains, maxdim=8, tol=1e-8)
    print(f"compress_batch: {len(chains)} chains, bonds {sorted({max(c.bond_dims) for c in chains})}")
    ok = err < 1e-6 and same and err_t < 1e-6          # the DT builder truncates at 1e-14 per bond (the reference default)
    print("OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
