"""One-off: qil_apply_compress_batch on nb (operator, state) pairs of the cfg4-shaped pipeline against one pair alone.
gpurun -- python tools/_apply_compress_batch_time.py [nb]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8
ctx = qil.default_context()
n, N = 24, 2 ** 24
j = np.arange(N, dtype=np.float64)
rng = np.random.default_rng(1001)
def signal(seed):
    r = np.random.default_rng(seed)
    x = np.sin(2 * np.pi * 5.0 * j / N) * np.exp(-3.0 * j / N) + 0.5 * np.cos(2 * np.pi * 11.0 * j / N)
    return x + sum(0.1 * r.random() * np.sin(40.0 * (r.random() - 0.5) * j / N) for _ in range(6))
psis = [qil.signal_ztmps(signal(s), method="rsvd", k=15, p=5, q=2, cutoff=1e-12) for s in range(nb)]
W = qil.build_zt_mpo(psis[0], 2 * np.pi)
for rep in range(2):
    ctx.synchronize(); t0 = time.perf_counter(); one = qil.apply_compress(W, psis[0], maxdim=64, tol=1e-8); ctx.synchronize(); t1 = time.perf_counter() - t0
for rep in range(2):
    ctx.synchronize(); t0 = time.perf_counter(); outs = qil.apply_compress_batch(W, psis, maxdim=64, tol=1e-8); ctx.synchronize(); tb = time.perf_counter() - t0
print(f"apply_compress n=24 paired, zT MPO D{max(W.bond_dims)} x chi{max(psis[0].bond_dims)}, maxdim 64: one pair {t1*1e3:.1f} ms, "
      f"{nb} pairs as one batch {tb*1e3:.1f} ms = {tb/t1:.2f} x one pair; bonds {max(outs[-1].bond_dims)}", flush=True)
