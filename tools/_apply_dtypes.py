import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
def sat(L, cap, base): return [int(min(base ** (i + 1), base ** (L - 1 - i), cap)) for i in range(L - 1)]
L, chi, D = 24, 64, 128
for wdt, adt in ((np.float64, np.float64), (np.complex128, np.float64), (np.float64, np.complex128), (np.complex128, np.complex128)):
    psi = qil.SignalMPS.alloc(sat(L, chi, 2), dtype=adt).fill_random(1)
    W = qil.SingleSiteMPO.alloc(sat(L, D, 4), dtype=wdt).fill_random(2)
    out = qil.apply(W, psi); ctx.synchronize()
    esz = 8 if (wdt == np.float64 and adt == np.float64) else 16
    c = [1] + sat(L, chi, 2) + [1]; d = [1] + sat(L, D, 4) + [1]
    nbytes = sum(esz * c[i] * d[i] * 2 * c[i + 1] * d[i + 1] for i in range(L))
    ctx.timer_start()
    for _ in range(5):
        qil.apply(W, psi, out=out)
    ms = ctx.timer_stop() / 5
    print(dict(W=np.dtype(wdt).name, A=np.dtype(adt).name, out_GB=round(nbytes / 1e9, 2), ms=round(ms, 3), GBps=round(nbytes / ms / 1e6, 0)), flush=True)
    del out, psi, W
