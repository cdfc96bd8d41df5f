
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
import oracle as O
rng = np.random.default_rng(0)
for (m, n, r, cplx) in [(900, 700, 50, False), (1008, 2016, 60, True), (2016, 1008, 110, False), (1500, 1200, 300, False)]:
    G = lambda *s: rng.standard_normal(s) + (1j * rng.standard_normal(s) if cplx else 0)
    A = (G(m, r) * np.logspace(0, -5, r)) @ G(r, n)
    A += 1e-17 * np.abs(A).max() * G(m, n)
    Sref = np.linalg.svd(A, compute_uv=False)
    qil.svd_trunc(A, cutoff=1e-12)
    t0 = time.perf_counter(); U, S, Vh = qil.svd_trunc(A, cutoff=1e-12); dt = time.perf_counter() - t0
    kk = O.truncation_rank(Sref, cutoff=1e-12)
    print(os.environ.get("QIL_SVD_LOWRANK", "1"), (m, n, r, cplx), "%.1f ms" % (dt * 1e3), "rank", len(S), "ref", kk,
          "S err %.1e" % (np.abs(S - Sref[:len(S)]).max() / Sref[0]), "recon %.1e" % (np.abs((U * S) @ Vh - A).max() / np.abs(A).max()),
          "iso %.1e %.1e" % (np.abs(U.conj().T @ U - np.eye(len(S))).max(), np.abs(Vh @ Vh.conj().T - np.eye(len(S))).max()), flush=True)
