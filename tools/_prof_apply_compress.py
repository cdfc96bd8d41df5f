import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
def sat(L, cap, base): return [int(min(base ** (i + 1), base ** (L - 1 - i), cap)) for i in range(L - 1)]
L, chi, D = 48, 15, 89
psi = qil.ZTMPS.alloc(sat(L, chi, 2), dtype=np.float64).fill_random(1)
W = qil.PairedSiteMPO.alloc(sat(L, D, 4), dtype=np.complex128).fill_random(2)
for _ in range(3):
    t0 = time.perf_counter(); out = qil.apply_compress(W, psi, maxdim=64, tol=1e-8); ctx.synchronize()
    print("apply_compress: %.3f s" % (time.perf_counter() - t0), flush=True)
