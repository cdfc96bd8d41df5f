"""qil_apply_compress_batch on the (operator, state) pairs of a damping sweep (cfg4-shaped): nb zT (or DT) operators of
linspace(0.25, 16, nb) x ONE encoded n = 24 signal, maxdim 64, tol 1e-8 -- against one pair alone.
gpurun -- python tools/_apply_compress_batch64.py [nb] [zt|dt] [repetitions of the batch, default 3]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
import bench
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 64
kind = sys.argv[2] if len(sys.argv) > 2 else "zt"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
ctx = qil.default_context()
psi = qil.signal_ztmps(bench.truncate_signal(24), method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
sig = np.linspace(0.25, 16.0, nb)
t0 = time.perf_counter()
Ws = qil.build_zt_mpo_batch(psi, sig) if kind == "zt" else qil.build_dt_mpo_batch(psi, sig)
ctx.synchronize()
t_build = time.perf_counter() - t0
for rep in range(2):
    ctx.synchronize(); t0 = time.perf_counter(); one = qil.apply_compress(Ws[nb // 2], psi, maxdim=64, tol=1e-8); ctx.synchronize(); t1 = time.perf_counter() - t0
tb = []
for rep in range(reps):
    ctx.synchronize(); t0 = time.perf_counter(); outs = qil.apply_compress_batch(Ws, psi, maxdim=64, tol=1e-8); ctx.synchronize(); tb.append(time.perf_counter() - t0)
P = max(max(c * d for c, d in zip(psi.bond_dims, W.bond_dims)) for W in Ws)
print(f"apply_compress_batch n=24 paired, {nb} {kind} operators (D <= {max(max(W.bond_dims) for W in Ws)}, built in {t_build:.2f} s) x chi {max(psi.bond_dims)}, "
      f"product bond <= {P}, maxdim 64: one pair {t1*1e3:.1f} ms, batch {' '.join('%.1f' % (t*1e3) for t in tb)} ms = {min(tb)/t1:.2f} x one pair, "
      f"{nb/min(tb):.1f} pairs/s; bonds {max(max(o.bond_dims) for o in outs)}", flush=True)
