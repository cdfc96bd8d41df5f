"""One-off: compress!(apply(W, psi)) on a product with bond 15 x 89 = 1335 (the cfg4-shaped product VERDICT r01 timed at 7-9 s):
random operands with saturated bond profiles, 48 sites.  gpurun -- python tools/_compress_product_1335.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import qilaplace_jl_amd as qil
from helpers import random_mps_data, random_mpo_data, saturated_profile
ctx = qil.default_context()
rng = np.random.default_rng(1335)
L = 48
a = random_mps_data(saturated_profile(L, 15), rng)
w = random_mpo_data(saturated_profile(L, 89, base=4), rng, dtype=np.complex128)
psi, W = qil.ZTMPS(a), qil.PairedSiteMPO(w)
for rep in range(2):
    prod = W * psi
    ctx.synchronize(); t0 = time.perf_counter(); qil.compress(prod, maxdim=64, tol=1e-8); ctx.synchronize(); te = time.perf_counter() - t0
    ctx.synchronize(); t0 = time.perf_counter(); f = qil.apply_compress(W, psi, maxdim=64, tol=1e-8); ctx.synchronize(); tf = time.perf_counter() - t0
print(f"product bond {15 * 89}: compress!(apply) {te*1e3:.0f} ms (bonds {max(prod.bond_dims)}), fused apply_compress {tf*1e3:.0f} ms (bonds {max(f.bond_dims)})")
