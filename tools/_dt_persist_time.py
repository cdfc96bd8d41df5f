import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
os.environ["QIL_DT_PROFILE"] = "1"
for lo in (0.25, 1.0, 2.0):
    for nb in (1, 64):
        sig = np.linspace(lo, 16.0, nb) if nb > 1 else np.array([lo])
        qil.build_dt_mpo_batch(24, sig); ctx.synchronize()
        t0 = time.perf_counter(); Ws = qil.build_dt_mpo_batch(24, sig); ctx.synchronize()
        print(f"n=24 sigma from {lo} batch {nb}: {(time.perf_counter()-t0)*1e3:.1f} ms  max bond {max(max(W.bond_dims) for W in Ws)}", flush=True)
