import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
n = 20; N = 2 ** n
j = np.arange(N, dtype=np.float64)
x = np.sin(2 * np.pi * 5.0 * j / N) * np.exp(-3.0 * j / N) + 0.5 * np.cos(2 * np.pi * 11.0 * j / N)
psi = qil.signal_ztmps(x, method="rsvd", k=20, p=5, q=2, cutoff=1e-12)
W = qil.build_zt_mpo(psi, 2 * np.pi)
full = W * psi
fast = qil.apply_compress(W, psi, maxdim=64, tol=1e-9)
ks = np.arange(256); ls = np.arange(256)
for name, obj in (("materialised product (bond %d)" % max(full.bond_dims), full), ("compressed (bond %d)" % max(fast.bond_dims), fast)):
    qil.coefficient_grid(obj, ks[:8], ls[:8])
    t0 = time.perf_counter(); chi = qil.coefficient_grid(obj, ks, ls); t = time.perf_counter() - t0
    print(dict(case=name, path="dense block read-out (qil_mps_block)", queries=chi.size, seconds=round(t, 4), q_per_s=int(chi.size / t)), flush=True)
    pk = np.random.default_rng(0).permutation(256)        # any non-contiguous index set takes the per-query path
    t0 = time.perf_counter(); chi2 = qil.coefficient_grid(obj, ks[pk], ls[pk]); t = time.perf_counter() - t0
    print(dict(case=name, path="batched chains / GEMM", queries=chi2.size, seconds=round(t, 4), q_per_s=int(chi2.size / t),
               max_diff=float(np.abs(chi2 - chi[np.ix_(pk, pk)]).max())), flush=True)
