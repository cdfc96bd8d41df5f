"""One-off: single signal_ztmps(:rsvd) at n = 16..26 with the encoder's sub-trees sequential vs concurrent."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
for n in (16, 20, 24, 26):
    t = np.arange(2 ** n) / 2 ** n
    x = np.sin(2 * np.pi * 5 * t) * np.exp(-3 * t) + 0.5 * np.cos(2 * np.pi * 11 * t)
    y = np.random.default_rng(n).standard_normal(2 ** n)
    for name, sig, kw in (("structured k=15", x, dict(k=15, p=5, q=2, cutoff=1e-12)), ("random k=50", y, dict(k=50, p=5, q=2, cutoff=1e-12))):
        res = {}
        for depth in ("0", "1", "3"):
            os.environ["QIL_ENCODE_PAR_DEPTH"] = depth
            for rep in range(3):
                ctx.synchronize(); t0 = time.perf_counter(); psi = qil.signal_ztmps(sig, method="rsvd", **kw); ctx.synchronize(); dt = time.perf_counter() - t0
            res[depth] = dt * 1e3
        print(f"n={n} {name}: sequential {res['0']:.2f} ms, depth 1 {res['1']:.2f} ms, depth 3 {res['3']:.2f} ms (bonds {max(psi.bond_dims)})", flush=True)
