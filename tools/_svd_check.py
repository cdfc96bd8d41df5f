import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
rng = np.random.default_rng(11)
cases = [(300, 300, 0, "rand"), (500, 260, 1, "rand"), (260, 700, 0, "rand"), (1000, 1000, 0, "rand"), (1024, 1024, 1, "rand"),
         (600, 600, 0, "graded"), (512, 512, 1, "graded"), (700, 300, 0, "rank40"), (2048, 2048, 0, "rand"), (4096, 4096, 0, "rand"),
         (3000, 257, 0, "rand"), (320, 320, 0, "dup")]
if len(sys.argv) > 1:
    cases = [c for c in cases if str(c[0]) in sys.argv[1:]]
for (m, n, cplx, kind) in cases:
    r0 = min(m, n)
    A = rng.standard_normal((m, n))
    if cplx: A = A + 1j * rng.standard_normal((m, n))
    if kind == "graded":
        U0, _ = np.linalg.qr(A); V0, _ = np.linalg.qr(rng.standard_normal((n, n)) + (1j * rng.standard_normal((n, n)) if cplx else 0))
        A = (U0[:, :r0] * np.logspace(0, -24, r0)) @ V0[:, :r0].conj().T
    elif kind == "rank40":
        A = rng.standard_normal((m, 40)) @ rng.standard_normal((40, n))
    elif kind == "dup":
        A[:, 100:200] = A[:, 0:100]
    Af = np.asfortranarray(A)
    t0 = time.perf_counter(); U, S, Vh = qil.svd_trunc(Af, cutoff=None); t = time.perf_counter() - t0
    t1 = time.perf_counter(); Sref = np.linalg.svd(A, compute_uv=False); tl = time.perf_counter() - t1
    k = len(S)
    rec = np.abs((U * S) @ Vh - A).max() / np.abs(A).max()
    big = Sref[:k] > 1e-13 * Sref[0]
    serr = np.abs(S - Sref[:k]).max() / Sref[0]
    srel = (np.abs(S - Sref[:k])[big] / Sref[:k][big]).max()
    live = S > 1e-13 * S[0]
    ou = np.abs(U[:, live].conj().T @ U[:, live] - np.eye(live.sum())).max()
    ov = np.abs(Vh[live] @ Vh[live].conj().T - np.eye(live.sum())).max()
    print(dict(m=m, n=n, cplx=cplx, kind=kind, k=k, rec=float(rec), serr=float(serr), srel=float(srel), orthU=float(ou), orthV=float(ov),
               s=round(t, 3), lapack_s=round(tl, 3)), flush=True)
