"""zT MPOs for a sweep of damping values: per-value product + compression chains on one context vs. on
worker contexts (threads).  gpurun -- python tools/_zt_sweep_time.py [n] [values]"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nv = int(sys.argv[2]) if len(sys.argv) > 2 else 16
wrs = np.linspace(0.25, 16.0, nv)
ctx = qil.default_context()
qil.build_zt_mpo_batch(n, wrs[:2])
for workers in (1, 8, 8, 16, 16):
    t0 = time.perf_counter(); Ws = qil.build_zt_mpo_batch(n, wrs, workers=workers); ctx.synchronize()
    dt = time.perf_counter() - t0
    print(json.dumps({"case": "zt_sigma_batch_build", "n": n, "values": nv, "workers": workers, "seconds": round(dt, 4),
                      "max_bond": int(max(max(W.bond_dims) for W in Ws))}), flush=True)
    del Ws
