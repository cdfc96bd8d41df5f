"""cfg4 step (bench.py --workload dt_sweep_n24_s64) split into its host-visible parts: build_dt_mpo_batch (one launch of the
persistent builder + handles + copy-out), apply_coefficient_sweep (apply + 1024 read-outs per value), release of the 64 operators."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
import bench
ctx = qil.default_context()
n, nsig = 24, 64
psi = qil.signal_ztmps(bench.truncate_signal(n), method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
sig = np.linspace(0.25, 16.0, nsig)
bits = np.random.default_rng(7).integers(0, 2, size=(1024, 2 * n)).astype(np.uint8)
for rep in range(4):
    ctx.synchronize(); t0 = time.perf_counter()
    Ws = qil.build_dt_mpo_batch(psi, sig); ctx.synchronize(); t1 = time.perf_counter()
    res = qil.apply_coefficient_sweep(Ws, psi, bits); ctx.synchronize(); t2 = time.perf_counter()
    del Ws; ctx.synchronize(); t3 = time.perf_counter()
    r2 = qil.damping_sweep(psi, sig, bits); ctx.synchronize(); t4 = time.perf_counter()
    print(f"build_dt_mpo_batch {1e3*(t1-t0):.1f} ms, apply_coefficient_sweep {1e3*(t2-t1):.1f} ms, release {1e3*(t3-t2):.1f} ms; damping_sweep as a whole {1e3*(t4-t3):.1f} ms", flush=True)
