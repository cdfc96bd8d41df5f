"""One-off: signal_ztmps for nb signals of 2^n samples one after another and as one batch.  gpurun -- python tools/_encode_batch_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
for n, nb, kw in ((20, 8, dict(method="rsvd", k=15, p=5, q=2, cutoff=1e-12)), (20, 8, dict(method="rsvd", k=50, p=5, q=2, cutoff=1e-12)),
                  (16, 16, dict(method="svd", cutoff=1e-12))):
    rng = np.random.default_rng(n)
    t = np.arange(2 ** n) / 2 ** n
    xs = [np.sin(2 * np.pi * (2 + j) * t) * np.exp(-(1 + 0.3 * j) * t) + (0.05 * rng.standard_normal(2 ** n) if kw["method"] == "svd" else 0)
          for j in range(nb)]
    for rep in range(2):
        ctx.synchronize(); t0 = time.perf_counter(); one = [qil.signal_ztmps(x, **kw) for x in xs]; ctx.synchronize(); t1 = time.perf_counter() - t0
        ctx.synchronize(); t0 = time.perf_counter(); bat = qil.signal_ztmps_batch(xs, **kw); ctx.synchronize(); tb = time.perf_counter() - t0
    print(f"signal_ztmps n={n} x {nb} signals {kw}: one after another {t1*1e3:.1f} ms, one batch {tb*1e3:.1f} ms ({t1/tb:.2f} x), bonds {max(bat[0].bond_dims)}", flush=True)
