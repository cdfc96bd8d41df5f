import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
def sat(L, chi, base=2): return [int(min(base ** (i + 1), base ** (L - 1 - i), chi)) for i in range(L - 1)]
for dt in (np.float64, np.complex128):
    for chi in (128, 256, 512):
        L = 24
        times = []
        for rep in range(2):
            psi = qil.SignalMPS.alloc(sat(L, chi), dtype=dt).fill_random(5)
            ctx.synchronize(); t0 = time.perf_counter()
            qil.compress(psi, maxdim=chi // 2, tol=1e-10); ctx.synchronize()
            times.append(time.perf_counter() - t0)
        print(f"compress {np.dtype(dt).name} chi {chi}->{chi//2} 24 sites: {min(times)*1e3:.1f} ms bonds {max(psi.bond_dims)} norm-1 {abs(qil.norm(psi)-1):.1e}", flush=True)
