"""Persistent complex chain builder (csrc/qil_build_chain.hip) against the host chains and the oracle: dense operators (n <= 6),
bond dimensions (n <= 24), timing of build_qft_mpo / the paired QFT chain at n = 24 (persistent kernel, generic device route, host)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import qilaplace_jl_amd as qil
import oracle as O
from helpers import dense_mpo
ctx = qil.default_context()
for n in (1, 2, 3, 4, 5, 6):
    W = qil.qft_mpo_device(n)
    ref = dense_mpo(O.build_qft_mpo(n).data)
    Q = qil.zt_qft_chain_device(n)
    Qh = qil.zt_qft_chain_tensors(n)
    print(f"n={n}: QFT dense err {np.abs(dense_mpo(W.to_host()) - ref).max():.2e} bonds {W.bond_dims} (oracle {O.build_qft_mpo(n).bond_dims}); "
          f"paired chain dense err {np.abs(dense_mpo(Q.to_host()) - dense_mpo(Qh)).max():.2e} bonds {Q.bond_dims} host {[t.shape[3] for t in Qh[:-1]]}", flush=True)
for n in (8, 12, 16, 20, 24):
    for cutoff in (1e-14, 1e-15):
        W = qil.qft_mpo_device(n, cutoff=cutoff, maxdim=None)
        Wh = qil.qft_mpo_tensors(n, cutoff, None)
        Q = qil.zt_qft_chain_device(n, cutoff=cutoff, maxdim=None)
        Qh = qil.zt_qft_chain_tensors(n, cutoff, None)
        print(f"n={n} cutoff={cutoff:g}: QFT bonds equal host {W.bond_dims == [t.shape[3] for t in Wh[:-1]]} max {max(W.bond_dims)}; "
              f"paired chain equal host {Q.bond_dims == [t.shape[3] for t in Qh[:-1]]} max {max(Q.bond_dims)}", flush=True)
n = 24
def timeit(f, reps=3):
    f(); ctx.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f(); ctx.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts) * 1e3
print(f"n=24 build_qft_mpo: persistent {timeit(lambda: qil.qft_mpo_device(n)):.1f} ms, generic device route {timeit(lambda: qil.qft_mpo_device(n, persistent=False)):.1f} ms, "
      f"host numpy {timeit(lambda: qil.qft_mpo_tensors(n)):.1f} ms")
import qilaplace_jl_amd.builders as B
def host_chain():
    B._ZT_Q_CACHE.clear(); return qil.zt_qft_chain_tensors(n)
print(f"n=24 paired QFT chain: persistent {timeit(lambda: qil.zt_qft_chain_device(n)):.1f} ms, generic device route {timeit(lambda: qil.zt_qft_chain_device(n, persistent=False)):.1f} ms, "
      f"host numpy {timeit(host_chain):.1f} ms")
for n in (6, 8, 12, 16):
    print(f"n={n} build_qft_mpo: persistent {timeit(lambda: qil.qft_mpo_device(n)):.2f} ms, host numpy + upload {timeit(lambda: qil.build_qft_mpo(n, device=False)):.2f} ms")
