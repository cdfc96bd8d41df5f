import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import qilaplace_jl_amd as qil
ctx = qil.default_context()
for n in (30, 48, 64, 100, 128, 200, 256):
    t0 = time.perf_counter(); W = qil.qft_mpo_device(n); ctx.synchronize(); t1 = time.perf_counter() - t0
    Wh = qil.qft_mpo_tensors(n) if n <= 64 else None
    print(f"QFT n={n}: {t1*1e3:.1f} ms, max bond {max(W.bond_dims)}, equal host bonds: {None if Wh is None else W.bond_dims == [t.shape[3] for t in Wh[:-1]]}", flush=True)
for n in (30, 48, 64, 100, 128):
    t0 = time.perf_counter(); Q = qil.zt_qft_chain_device(n); ctx.synchronize(); t1 = time.perf_counter() - t0
    Qh = qil.zt_qft_chain_tensors(n) if n <= 40 else None
    print(f"paired chain n={n}: {t1*1e3:.1f} ms, sites {len(Q.bond_dims)+1}, max bond {max(Q.bond_dims)}, equal host: {None if Qh is None else Q.bond_dims == [t.shape[3] for t in Qh[:-1]]}", flush=True)
try:
    qil.zt_qft_chain_device(129)
    print("n=129 paired: fell back to the generic route (L > 256)")
except Exception as e:
    print("n=129:", type(e).__name__, str(e)[:100])
