import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
def sat(L, chi, base=2): return [int(min(base ** (i + 1), base ** (L - 1 - i), chi)) for i in range(L - 1)]
L, chi = 20, 256
psi = qil.SignalMPS.alloc(sat(L, chi), dtype=np.float64).fill_random(5)
qil.compress(psi, maxdim=chi // 2, tol=1e-10)
psi = qil.SignalMPS.alloc(sat(L, chi), dtype=np.float64).fill_random(5)
os.environ["QIL_SVD_DEBUG"] = "1"
t0 = time.perf_counter(); qil.compress(psi, maxdim=chi // 2, tol=1e-10); ctx.synchronize()
print("total", time.perf_counter() - t0)
