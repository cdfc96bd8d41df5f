"""Persistent DT builder: launch time against the number of damping values in the launch (is the launch as long as its
slowest chain, or do the chains slow each other down?).  QIL_DT_PROFILE=1 prints per-value cycle totals."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for nb in (1, 2, 8, 16, 32, 64, 128, 256):
    sig = np.linspace(2.0, 16.0, nb) if nb > 1 else np.array([2.0])
    qil.build_dt_mpo_batch(n, sig); ctx.synchronize()
    t0 = time.perf_counter(); Ws = qil.build_dt_mpo_batch(n, sig); ctx.synchronize()
    print(f"n={n} batch {nb}: {(time.perf_counter()-t0)*1e3:.1f} ms", flush=True)
