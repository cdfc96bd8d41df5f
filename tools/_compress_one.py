"""One compress! configuration (for rocprofv3): python tools/_compress_one.py [chi] [f64|c64] [reps]."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
chi = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = np.complex128 if len(sys.argv) > 2 and sys.argv[2] == "c64" else np.float64
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
def sat(L, chi, base=2): return [int(min(base ** (i + 1), base ** (L - 1 - i), chi)) for i in range(L - 1)]
times = []
for rep in range(reps):
    psi = qil.SignalMPS.alloc(sat(24, chi), dtype=dt).fill_random(5)
    ctx.synchronize(); t0 = time.perf_counter()
    qil.compress(psi, maxdim=chi // 2, tol=1e-10); ctx.synchronize()
    times.append(time.perf_counter() - t0)
print(f"compress {np.dtype(dt).name} chi {chi}->{chi//2}: " + " ".join(f"{t*1e3:.1f}" for t in times) + " ms", flush=True)
