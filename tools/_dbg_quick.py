import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
rng = np.random.default_rng(0)
np.set_printoptions(precision=2, linewidth=200)
M = rng.standard_normal((32, 4)) @ np.diag([1, 1e-2, 1e-7, 1e-10]) @ rng.standard_normal((4, 32))
for l in (16, 17, 20):
    Y = M @ rng.standard_normal((32, l))
    Q, R = qil.qr_positive(Y)
    G = Q.T @ Q
    print(l, "diag(QtQ)", np.diag(G))
    off = np.abs(G - np.diag(np.diag(G)))
    print("   max offdiag %.2e" % off.max(), "Rdiag", np.diag(R))
