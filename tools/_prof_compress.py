import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
L, chi = 24, int(sys.argv[1]) if len(sys.argv) > 1 else 64
cb = [min(2 ** (i + 1), 2 ** (L - 1 - i), chi) for i in range(L - 1)]
for rep in range(3):
    psi = qil.SignalMPS.alloc(cb, dtype=np.float64, ctx=ctx); psi.fill_random(3 + rep)
    ctx.synchronize(); t0 = time.perf_counter()
    qil.compress(psi, maxdim=chi // 2, tol=1e-10); ctx.synchronize()
    print("compress chi=%d -> %d: %.2f ms" % (chi, max(psi.bond_dims), 1e3 * (time.perf_counter() - t0)), flush=True)
