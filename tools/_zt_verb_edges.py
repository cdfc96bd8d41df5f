import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qilaplace_jl_amd as qil, oracle as O
from helpers import dense_mpo
ctx = qil.default_context()
W = qil.build_zt_mpo(1, 0.7); R = O.build_zt_mpo(1, 0.7)
print("n=1", W.bond_dims, np.abs(dense_mpo(W.to_host()) - dense_mpo(R.data)).max())
for n, kw in ((2, dict(cutoff=0.0, maxdim=None)), (3, dict(maxdim=1)), (5, dict(cutoff=1e-6)), (7, dict(cutoff=1e-14))):
    W = qil.build_zt_mpo(n, 1.1, **kw); H = qil.zt_mpo_tensors(n, 1.1, kw.get("cutoff", 1e-14), kw.get("maxdim", 1000))
    print(n, kw, W.bond_dims == [t.shape[3] for t in H[:-1]], np.abs(dense_mpo(W.to_host()) - dense_mpo(H)).max() if n <= 5 else "")
Ws = qil.build_zt_mpo_batch(8, np.linspace(0.1, 20, 300))
print("300 values at n=8:", len(Ws), max(max(w.bond_dims) for w in Ws), ctx.unowned_bytes())
W40 = qil.build_zt_mpo(40, 2 * np.pi)
print("n=40 max bond", max(W40.bond_dims), len(W40.to_host()))
try:
    qil.build_zt_mpo(140, 1.0)
    print("n=140 built")
except Exception as e:
    print("n=140:", type(e).__name__, str(e)[:120])
try:
    qil.build_zt_mpo_batch(4, [])
except Exception as e:
    print("empty batch:", type(e).__name__, str(e)[:100])
print("unowned", ctx.unowned_bytes())
