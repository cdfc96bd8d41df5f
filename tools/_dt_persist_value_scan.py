import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import qilaplace_jl_amd as qil
ctx = qil.default_context()
for sig in ([2.0, 16.0], [4.0, 8.0], [0.25, 1.0], [16.0, 32.0]):
    qil.build_dt_mpo_batch(24, np.array(sig)); ctx.synchronize()
    Ws = qil.build_dt_mpo_batch(24, np.array(sig)); ctx.synchronize()
    print(sig, [max(W.bond_dims) for W in Ws], flush=True)
