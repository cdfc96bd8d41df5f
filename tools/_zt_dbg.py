import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
wr = 2 * np.pi
host = qil.zt_mpo_tensors(n, wr)
print("host        ", [t.shape[3] for t in host[:-1]])
Wb = qil.build_zt_mpo_batch(n, [wr])[0]
print("dev-assisted", Wb.bond_dims)
# host DT, device compose + compress
dt = qil.PairedSiteMPO(qil.dt_mpo_tensors(n, wr))
Q = qil.PairedSiteMPO(qil.zt_qft_chain_tensors(n))
W = qil.apply(dt, Q); print("product     ", W.bond_dims)
qil.mpo_compress(W, "down", 1e-14, 1000)
print("hostDT+dev  ", W.bond_dims)
# device DT, host compose+compress
import importlib
B = importlib.import_module(qil.__name__ + ".builders")
ddt = qil.build_dt_mpo_batch(n, [wr])[0]
print("devDT bonds ", ddt.bond_dims, " hostDT bonds", dt.bond_dims)
dd = [ddt.site(i) for i in range(2 * n)]
Wh = B._compress_lr([B._compose(a, b) for a, b in zip(dd, qil.zt_qft_chain_tensors(n))], 1e-14, 1000)
print("devDT+host  ", [t.shape[3] for t in Wh[:-1]])
# singular values at the biggest bond, host path
prod = [B._compose(a, b) for a, b in zip(qil.dt_mpo_tensors(n, wr), qil.zt_qft_chain_tensors(n))]
out = list(prod); L = len(out)
for i in range(L - 1):
    a, _, _, b = out[i].shape
    Qm, Rm = np.linalg.qr(out[i].reshape(a * 4, b)); out[i] = Qm.reshape(a, 2, 2, Qm.shape[1]); out[i + 1] = np.tensordot(Rm, out[i + 1], axes=([1], [0]))
for i in range(L - 1, 0, -1):
    a0, b1 = out[i - 1].shape[0], out[i].shape[3]
    core = np.tensordot(out[i - 1], out[i], axes=([3], [0])).reshape(a0 * 4, 4 * b1)
    U, S, Vh = np.linalg.svd(core, full_matrices=False)
    P = S * S; tot = P.sum(); r = B._keep(S, 1e-14, 1000)
    tail = np.cumsum(P[::-1])[::-1] / tot
    print("bond", i, "kept", r, "tail weights around cut:", ["%.2e" % t for t in tail[max(0, r - 2): r + 3]])
    out[i] = Vh[:r].reshape(r, 2, 2, b1); out[i - 1] = (U[:, :r] * S[:r]).reshape(a0, 2, 2, r)
