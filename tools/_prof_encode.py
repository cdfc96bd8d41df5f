import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
rng = np.random.default_rng(3)
x = rng.standard_normal(2 ** 24)
for _ in range(3):
    psi = qil.signal_mps(x, method="rsvd", k=50, p=5, q=2)
