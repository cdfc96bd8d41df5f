import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
rng = np.random.default_rng(1)
L, chi, D = 48, 128, 82
cb = [min(2 ** (i + 1), 2 ** (L - 1 - i), chi) for i in range(L - 1)]
db = [min(4 ** (i + 1), 4 ** (L - 1 - i), D) for i in range(L - 1)]
psi = qil.SignalMPS.alloc(cb, dtype=np.float64, ctx=ctx)
psi.fill_random(3); W = qil.SingleSiteMPO.alloc(db, dtype=np.complex128, ctx=ctx); W.fill_random(4)
for nb in (64, 1024, 4096):
    bits = rng.integers(0, 2, size=(nb, L)).astype(np.uint8)
    qil.apply_coefficient_batch(W, psi, bits[:16])
    t0 = time.perf_counter(); v = qil.apply_coefficient_batch(W, psi, bits); t = time.perf_counter() - t0
    print(dict(mode=os.environ.get("QIL_LAZY_GEMM_MIN", "default"), sites=L, chi=chi, D=D, queries=nb, seconds=round(t, 4),
               us_per_query=round(1e6 * t / nb, 1), checksum=float(np.abs(v).sum())), flush=True)
