import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import qilaplace_jl_amd as qil
import bench
ctx = qil.default_context()
W, psi = bench.truncate_operands(qil, 24)
bits = np.random.default_rng(3).integers(0, 2, size=(256, 48)).astype(np.uint8)
c_x = qil.apply_coefficient_batch(W, psi, bits); scale = np.abs(c_x).max()
for cap in (None, 88, 80, 76, 72, 68):
    f = qil.apply_compress(W, psi, maxdim=64, tol=1e-8, zip_maxdim=cap); ctx.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); f = qil.apply_compress(W, psi, maxdim=64, tol=1e-8, zip_maxdim=cap); ctx.synchronize(); ts.append(time.perf_counter() - t0)
    err = np.abs(qil.coefficient_batch(f, bits) - c_x).max() / scale
    print(f"zip cap {cap}: {min(ts)*1e3:.1f} ms, bonds {max(f.bond_dims)}, err vs exact product {err:.2e}", flush=True)
