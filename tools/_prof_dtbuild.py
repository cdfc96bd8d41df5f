import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
sig = np.linspace(0.25, 16.0, 64)
Ws = qil.build_dt_mpo_batch(24, sig); qil.default_context().synchronize()
t0 = time.perf_counter(); Ws = qil.build_dt_mpo_batch(24, sig); qil.default_context().synchronize()
print("batch build n=24 nb=64: %.3f s" % (time.perf_counter() - t0))
