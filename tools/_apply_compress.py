import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
def sat(L, cap, base): return [int(min(base ** (i + 1), base ** (L - 1 - i), cap)) for i in range(L - 1)]
for (L, chi, D, paired) in ((24, 8, 8, False), (48, 15, 18, True), (48, 15, 89, True)):
    mps_cls = qil.ZTMPS if paired else qil.SignalMPS
    mpo_cls = qil.PairedSiteMPO if paired else qil.SingleSiteMPO
    psi = mps_cls.alloc(sat(L, chi, 2), dtype=np.float64).fill_random(1)
    W = mpo_cls.alloc(sat(L, D, 4), dtype=np.complex128).fill_random(2)
    out = qil.apply(W, psi); ctx.synchronize()
    t0 = time.perf_counter(); qil.compress(out, maxdim=64, tol=1e-8); ctx.synchronize()
    print(dict(sites=L, chi=chi, D=D, bond_before=chi * D, seconds_compress=round(time.perf_counter() - t0, 4), bonds_after=max(out.bond_dims)), flush=True)
print("--- fused apply_compress on the same shapes")
for (L, chi, D, paired) in ((48, 15, 18, True), (48, 15, 89, True)):
    mps_cls = qil.ZTMPS if paired else qil.SignalMPS
    mpo_cls = qil.PairedSiteMPO if paired else qil.SingleSiteMPO
    psi = mps_cls.alloc(sat(L, chi, 2), dtype=np.float64).fill_random(1)
    W = mpo_cls.alloc(sat(L, D, 4), dtype=np.complex128).fill_random(2)
    qil.apply_compress(W, psi, maxdim=64, tol=1e-8); ctx.synchronize()
    t0 = time.perf_counter(); out = qil.apply_compress(W, psi, maxdim=64, tol=1e-8); ctx.synchronize()
    print(dict(sites=L, chi=chi, D=D, seconds_apply_compress=round(time.perf_counter() - t0, 4), bonds_after=max(out.bond_dims)), flush=True)
print("--- genuine zT pipeline, n=20 structured signal")
n = 20; N = 2 ** n
j = np.arange(N, dtype=np.float64)
x = np.sin(2 * np.pi * 5.0 * j / N) * np.exp(-3.0 * j / N) + 0.5 * np.cos(2 * np.pi * 11.0 * j / N)
psi = qil.signal_ztmps(x, method="rsvd", k=20, p=5, q=2, cutoff=1e-12)
W = qil.build_zt_mpo(psi, 2 * np.pi)
t0 = time.perf_counter(); full = W * psi; ctx.synchronize(); t_apply = time.perf_counter() - t0
t0 = time.perf_counter(); slow = full.copy(); qil.compress(slow, maxdim=64, tol=1e-8); ctx.synchronize(); t_comp = time.perf_counter() - t0
t0 = time.perf_counter(); fast = qil.apply_compress(W, psi, maxdim=64, tol=1e-8); ctx.synchronize(); t_fused = time.perf_counter() - t0
ks, ls = np.arange(0, 40, 3), np.arange(0, 24, 2)
ref = qil.coefficient_grid(full, ks, ls)
e_s = np.abs(qil.coefficient_grid(slow, ks, ls) - ref).max() / np.abs(ref).max()
e_f = np.abs(qil.coefficient_grid(fast, ks, ls) - ref).max() / np.abs(ref).max()
print(dict(n=n, mps_bond=max(psi.bond_dims), mpo_bond=max(W.bond_dims), product_bond=max(full.bond_dims),
           seconds_apply=round(t_apply, 4), seconds_compress=round(t_comp, 4), seconds_fused=round(t_fused, 4),
           bonds_slow=max(slow.bond_dims), bonds_fused=max(fast.bond_dims), err_slow=float(e_s), err_fused=float(e_f)))
