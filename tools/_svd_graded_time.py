"""Jacobi SVD on graded spectra (singular values decaying geometrically to the cutoff -- what every truncation
after an apply sees) across the three regimes; QIL_RT_MIN moves the R^H preconditioning threshold."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
rng = np.random.default_rng(0)
for (m, k) in ((64, 32), (128, 64), (192, 96), (96, 96), (512, 256), (1024, 512)):
    for kind in ("graded", "random"):
        A = rng.standard_normal((m, k))
        if kind == "graded":
            A = (A * np.exp(-np.arange(k) * (32.0 / k))) @ np.linalg.qr(rng.standard_normal((k, k)))[0]
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); U, S, Vh = qil.svd_trunc(A, cutoff=1e-16); ts.append(time.perf_counter() - t0)
        err = float(np.abs((U * S) @ Vh - A).max())
        print(dict(case="svd_" + kind, m=m, n=k, kept=len(S), ms=round(1e3 * min(ts), 3), err=err), flush=True)
