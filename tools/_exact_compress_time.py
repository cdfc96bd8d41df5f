"""The exact route of the bench's truncate block alone: compress!(W_zt * psi; maxdim=64, tol=1e-8) on the bond-1008 product,
timed (3 repetitions; for rocprofv3 --kernel-trace --stats).  QIL_SVD_DEBUG=1 prints the per-site SVD phases."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
n, N = 24, 2 ** 24
j = np.arange(N, dtype=np.float64)
x = np.sin(2 * np.pi * 5.0 * j / N) * np.exp(-3.0 * j / N) + 0.5 * np.cos(2 * np.pi * 11.0 * j / N)
rng = np.random.default_rng(1001)
x = x + sum(0.1 * rng.random() * np.sin(40.0 * (rng.random() - 0.5) * j / N) for _ in range(6))
psi = qil.signal_ztmps(x, method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
W = qil.build_zt_mpo(psi, 2 * np.pi)
ts = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    prod = W * psi
    ctx.synchronize()
    t0 = time.perf_counter()
    qil.compress(prod, maxdim=64, tol=1e-8)
    ctx.synchronize()
    ts.append(time.perf_counter() - t0)
print("exact compress of the bond-%d product: %s ms, bonds %d" % (max(c * d for c, d in zip(psi.bond_dims, W.bond_dims)),
      " ".join("%.1f" % (t * 1e3) for t in ts), max(prod.bond_dims)), flush=True)
