"""64 coefficients of the materialised cfg3 product (80 GB): batched-GEMM read-out time."""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
L, chi, D = 48, 64, 128
cb = [min(2 ** (i + 1), 2 ** (L - 1 - i), chi) for i in range(L - 1)]
db = [min(4 ** (i + 1), 4 ** (L - 1 - i), D) for i in range(L - 1)]
psi = qil.ZTMPS.alloc(cb, dtype=np.float64, ctx=ctx); psi.fill_random(1)
W = qil.PairedSiteMPO.alloc(db, dtype=np.complex128, ctx=ctx); W.fill_random(2)
out = W * psi
ctx.synchronize()
bits = np.random.default_rng(0).integers(0, 2, size=(64, L)).astype(np.uint8)
qil.coefficient_batch(out, bits[:4])
for nb in (64, 256):
    b = np.random.default_rng(nb).integers(0, 2, size=(nb, L)).astype(np.uint8)
    t0 = time.perf_counter(); c = qil.coefficient_batch(out, b); t = time.perf_counter() - t0
    print(dict(queries=nb, seconds=round(t, 4), product_GB=80.1, effective_read_TBps=round(80.1e9 / t / 1e12, 2)), flush=True)
