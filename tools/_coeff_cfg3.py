"""64 coefficients of a materialised cfg3-shaped product (80 GB): bit-sorted GEMM read-out time; argv[1] = repetitions of the
64-query read-out (tools/collect_pmc_truncate.py counts its f64 MFMA instructions as the difference of a 3- and a 1-repetition run)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
L, chi, D = 48, 64, 128
cb = [min(2 ** (i + 1), 2 ** (L - 1 - i), chi) for i in range(L - 1)]
db = [min(4 ** (i + 1), 4 ** (L - 1 - i), D) for i in range(L - 1)]
psi = qil.ZTMPS.alloc(cb, dtype=np.float64, ctx=ctx); psi.fill_random(1)
W = qil.PairedSiteMPO.alloc(db, dtype=np.complex128, ctx=ctx); W.fill_random(2)
out = W * psi
ctx.synchronize()
bits = np.random.default_rng(64).integers(0, 2, size=(64, L)).astype(np.uint8)
for r in range(reps):
    t0 = time.perf_counter(); c = qil.coefficient_batch(out, bits); t = time.perf_counter() - t0
    print(dict(queries=64, seconds=round(t, 4), product_GB=80.1, effective_read_TBps=round(80.1e9 / t / 1e12, 2)), flush=True)
