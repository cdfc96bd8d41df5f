import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
n, N = 24, 2 ** 24
j = np.arange(N, dtype=np.float64)
x = np.sin(2 * np.pi * 5.0 * j / N) * np.exp(-3.0 * j / N) + 0.5 * np.cos(2 * np.pi * 11.0 * j / N)
rng = np.random.default_rng(1001)
x = x + sum(0.1 * rng.random() * np.sin(40.0 * (rng.random() - 0.5) * j / N) for _ in range(6))
psi = qil.signal_ztmps(x, method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
W = qil.build_zt_mpo(psi, 2 * np.pi)
f = qil.apply_compress(W, psi, maxdim=64, tol=1e-8)
ctx.synchronize()
os.environ["QIL_FUSED_DEBUG"] = "1"
f = qil.apply_compress(W, psi, maxdim=64, tol=1e-8)
