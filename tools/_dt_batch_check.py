import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
def int_to_bits(v, n, order="msb"):
    b = [(v >> (n - 1 - i)) & 1 for i in range(n)]
    return b if order == "msb" else b[::-1]
n = 8; N = 2 ** n
j = np.arange(N); x = np.sin(2*np.pi*3*j/N)*np.exp(-2.0*j/N) + 0.3
psi = qil.signal_ztmps(x, cutoff=1e-14)
sig = np.linspace(0.25, 16.0, 9)
Wd = qil.build_dt_mpo_batch(psi, sig)
xh = x / np.linalg.norm(x)
for name, Ws in (("device", Wd), ("host", [qil.build_dt_mpo(psi, s) for s in sig])):
    worst = 0
    for W, s_ in zip(Ws, sig):
        if name == "device":
            W = qil.PairedSiteMPO(W.to_host(), sites=psi.site_ids)
        out = W * psi
        for k in (0, 1, 5, 77, 200):
            bits = np.array([[b for pair in zip(int_to_bits(int(k), n, "lsb"), int_to_bits(jj, n)) for b in pair] for jj in range(N)])
            ref = psi.amplitude * xh * np.exp(-s_ * k * np.arange(N) / N) / np.sqrt(N)
            worst = max(worst, np.abs(qil.coefficient_batch(out, bits) - ref).max())
    print(name, "worst abs err", worst, "bonds", Ws[0].bond_dims)
for nn, nb in ((24, 64), (24, 8), (16, 64)):
    sig = np.linspace(0.25, 16.0, nb)
    t0 = time.perf_counter(); Ws = qil.build_dt_mpo_batch(nn, sig); qil.default_context().synchronize()
    t1 = time.perf_counter() - t0
    t0 = time.perf_counter(); Ws = qil.build_dt_mpo_batch(nn, sig); qil.default_context().synchronize()
    print("batch build n=%d nb=%d: %.3f s (first %.3f) maxbond %d" % (nn, nb, time.perf_counter() - t0, t1, max(Ws[0].bond_dims)))
