"""The bench's fused apply_compress alone (for rocprofv3): python tools/_apply_compress_one.py [reps]."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, N = 24, 2 ** 24
j = np.arange(N, dtype=np.float64)
x = np.sin(2 * np.pi * 5.0 * j / N) * np.exp(-3.0 * j / N) + 0.5 * np.cos(2 * np.pi * 11.0 * j / N)
rng = np.random.default_rng(1001)
x = x + sum(0.1 * rng.random() * np.sin(40.0 * (rng.random() - 0.5) * j / N) for _ in range(6))
psi = qil.signal_ztmps(x, method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
W = qil.build_zt_mpo(psi, 2 * np.pi)
qil.apply_compress(W, psi, maxdim=64, tol=1e-8)
ctx.synchronize()
print("MARK", flush=True)
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); f = qil.apply_compress(W, psi, maxdim=64, tol=1e-8); ctx.synchronize(); ts.append(time.perf_counter() - t0)
print("apply_compress ms:", " ".join(f"{t*1e3:.1f}" for t in ts), "bonds", max(f.bond_dims), flush=True)
