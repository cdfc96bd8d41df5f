"""signal_ztmps(:rsvd) of STRUCTURED signals generated in HBM (rank far below the sketch width: the library's use case):
n = 30 k = 128 and n = 24 k = 15 / 50; QIL_RSVD_DEBUG=1 prints the root split's stages and the deflated rank."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import qilaplace_jl_amd as qil
ctx = qil.default_context()
for n, k in ((30, 128), (24, 15), (24, 50)):
    N = 2 ** n
    jd = torch.arange(N, dtype=torch.float64, device="cuda")
    xd = torch.sin(2 * np.pi * 5.0 * jd / N) * torch.exp(-3.0 * jd / N) + 0.5 * torch.cos(2 * np.pi * 11.0 * jd / N)
    del jd
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        psi = qil.signal_ztmps(xd, method="rsvd", k=k, p=5, q=2, cutoff=1e-12, maxdim=k)
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    js = np.random.default_rng(1).integers(0, N, size=64)
    bits = np.zeros((64, 2 * n), dtype=np.uint8)
    jb = (js[:, None] >> np.arange(n - 1, -1, -1)[None, :]) & 1
    bits[:, 0::2] = jb
    bits[:, 1::2] = jb
    rec = qil.coefficient_batch(psi, bits)
    xs = xd[torch.as_tensor(js, device="cuda")].cpu().numpy()
    print(f"n={n} k={k}: encode " + " ".join(f"{t*1e3:.1f}" for t in ts) + f" ms, bonds max {max(psi.bond_dims)}, reconstruction err {np.abs(rec - xs).max() / np.abs(xs).max():.2e}", flush=True)
    del xd, psi
    torch.cuda.empty_cache()
