import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
rng = np.random.default_rng(5)
for (m, n, cplx, rank) in [(8192, 3, 0, 0), (100000, 16, 0, 0), (100001, 7, 1, 0), (1 << 20, 48, 0, 0), (1 << 18, 40, 1, 0),
                           (200000, 16, 0, 5), (1 << 23, 2, 0, 0), (300000, 33, 0, 9)]:
    if rank:
        A = rng.standard_normal((m, rank)) @ rng.standard_normal((rank, n))
    else:
        A = rng.standard_normal((m, n))
    if cplx: A = A + 1j * rng.standard_normal((m, n))
    t0 = time.perf_counter(); Q, R = qil.qr_positive(A); t = time.perf_counter() - t0
    rec = np.abs(Q @ R - A).max() / np.abs(A).max()
    G = Q.conj().T @ Q
    d = np.real(np.diag(G)); kept = d > 0.5
    orth = np.abs(G - np.diag(kept.astype(float))).max()
    print(dict(m=m, n=n, cplx=cplx, rank=rank, kept=int(kept.sum()), rec=float(rec), orth=float(orth),
               diag_min=float(np.real(np.diag(R)).min()), lower=float(np.abs(np.tril(R, -1)).max()), s=round(t, 3)), flush=True)
