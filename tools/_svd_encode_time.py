import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
rng = np.random.default_rng(3)
for n in [int(a) for a in sys.argv[1:]] or (16, 18, 20):
    x = rng.standard_normal(2 ** n)
    t0 = time.perf_counter(); psi = qil.signal_mps(x, method="svd"); ctx.synchronize()
    t = time.perf_counter() - t0
    err = np.abs(qil.mps_to_vector(psi) - x).max()
    print(dict(n=n, seconds=round(t, 3), maxbond=max(psi.bond_dims), err=float(err)), flush=True)
