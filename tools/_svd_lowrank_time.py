"""signal_mps(:svd) / compress! on numerically rank-deficient operands (structured signals, product bonds):
the case the negligible-column rule of the Jacobi sweeps is for.  QIL_SVD_NEGLIGIBLE=0 switches the rule off."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
for n in (16, 20, 22):
    t = np.arange(2 ** n) / 2 ** n
    x = np.sin(2 * np.pi * 5 * t) * np.exp(-3 * t) + 0.3 * np.cos(2 * np.pi * 17.3 * t)
    for rep in range(2):
        t0 = time.perf_counter(); psi = qil.signal_mps(x, method="svd", cutoff=1e-15); ctx.synchronize()
        dt = time.perf_counter() - t0
    print(dict(case="signal_mps_svd_structured", n=n, seconds=round(dt, 4), maxbond=max(psi.bond_dims)), flush=True)
rng = np.random.default_rng(0)
for (m, k, r) in ((512, 256, 40), (1024, 300, 60), (2048, 1024, 100)):
    A = rng.standard_normal((m, r)) @ rng.standard_normal((r, k))
    for rep in range(2):
        t0 = time.perf_counter(); U, S, Vh = qil.svd_trunc(A, cutoff=1e-14); dt = time.perf_counter() - t0
    print(dict(case="svd_trunc_lowrank", m=m, n=k, rank=r, kept=len(S), seconds=round(dt, 4),
               err=float(np.abs((U * S) @ Vh - A).max())), flush=True)
