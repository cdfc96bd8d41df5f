#!/usr/bin/env python3
"""BASELINE.json configs[4]: n=30 paired-register signal, signal_ztmps(:rsvd, k=128) encode + zT apply,
HBM-resident, 1 MI355X.  The materialised W*psi at saturated bonds does not fit one GPU (SURVEY.md 8d:
406 GB padded), so coefficients of W*psi are read through the lazy path (qil_apply_coefficient_batch),
and -- for the structured signal, whose encoded bonds are small -- also through the materialised apply.
Checks: sampled chi(k, l) against the closed form (1/N) sum_j x_j exp(-(wr k + 2 pi i l) j / N)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qilaplace_jl_amd as qil  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    kind = sys.argv[2] if len(sys.argv) > 2 else "structured"
    N = 2 ** n
    ctx = qil.default_context()
    # the samples are produced IN HBM (torch) and handed to the encoder as a device pointer: the 8.6 GB signal
    # never exists on the host ("generated on device")
    import torch
    dev = torch.device("cuda", 0)
    t0 = time.perf_counter()
    jd = torch.arange(N, dtype=torch.float64, device=dev)
    if kind == "structured":
        xd = torch.sin(2 * np.pi * 5.0 * jd / N) * torch.exp(-3.0 * jd / N) + 0.5 * torch.cos(2 * np.pi * 11.0 * jd / N)
    else:
        g = torch.Generator(device=dev)
        g.manual_seed(30)
        xd = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
    torch.cuda.synchronize()
    t_gen = time.perf_counter() - t0
    t0 = time.perf_counter()
    psi = qil.signal_ztmps(xd, method="rsvd", k=128, p=5, q=2, cutoff=1e-12, maxdim=128)
    ctx.synchronize()
    t_enc = time.perf_counter() - t0
    wr = 2 * np.pi
    t0 = time.perf_counter()
    W = qil.build_zt_mpo_batch(psi, [wr], cutoff=1e-14)[0]      # DT half, MPO x MPO product, compression on the GPU
    ctx.synchronize()
    t_build = time.perf_counter() - t0
    rng = np.random.default_rng(5)
    nq = 64
    kk = rng.integers(0, 64, size=nq)            # small k: chi(k, l) decays like exp(-wr k j / N)
    ll = rng.integers(0, 32, size=nq)
    bits = np.zeros((nq, 2 * n), dtype=np.uint8)
    for q in range(nq):
        for i in range(n):
            bits[q, 2 * i] = (kk[q] >> i) & 1      # main_i  <- bit i of k (lsb first)
            bits[q, 2 * i + 1] = (ll[q] >> i) & 1  # copy_i  <- bit i of l (lsb first)
    t0 = time.perf_counter()
    lazy = qil.apply_coefficient_batch(W, psi, bits)
    t_lazy = time.perf_counter() - t0
    out_bytes = sum(16 * a * 2 * b for a, b in zip([1] + [c * d for c, d in zip(psi.bond_dims, W.bond_dims)],
                                                  [c * d for c, d in zip(psi.bond_dims, W.bond_dims)] + [1]))
    res = {"case": "config5", "n": n, "signal": kind, "mps_bonds_max": max(psi.bond_dims),
           "mpo_bonds_max": max(W.bond_dims), "seconds_generate_device": t_gen, "seconds_encode": t_enc,
           "seconds_build_device_assisted": t_build, "seconds_lazy_64_coefficients": t_lazy,
           "materialised_output_bytes": out_bytes}
    if out_bytes < 200e9:
        t0 = time.perf_counter()
        out = W * psi
        ctx.synchronize()
        res["seconds_apply"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        for _ in range(5):
            del out
            out = W * psi
        ctx.synchronize()
        res["seconds_apply_steady"] = (time.perf_counter() - t0) / 5
        res["site_contractions_per_s"] = 2 * n / res["seconds_apply_steady"]
        mat = qil.coefficient_batch(out, bits)
        res["lazy_vs_materialised_rel"] = float(np.abs(mat - lazy).max() / np.abs(mat).max())
    if kind == "structured":
        # closed form on 4 sample points (each a 2^n-term sum)
        err = 0.0
        for q in range(4):
            ph = torch.exp(-(wr * float(kk[q])) * jd / N) * xd
            ang = -2 * np.pi * float(ll[q]) * jd / N
            ref = complex(torch.sum(ph * torch.cos(ang)).item(), torch.sum(ph * torch.sin(ang)).item()) / N
            err = max(err, abs(lazy[q] - ref) / max(abs(ref), 1e-300))
        res["max_rel_err_vs_closed_form_4pts"] = float(err)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
