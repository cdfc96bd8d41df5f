"""Persistent DT builder vs the launch-per-step builder and the oracle: operators, bond dims, timing."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import qilaplace_jl_amd as qil
import oracle as O
from helpers import dense_mpo

ctx = qil.default_context()
wrs = [0.0, 0.75, 1.0, 2.0, 5.0, 2 * np.pi]
for n in (1, 2, 3, 4, 5, 6):
    os.environ["QIL_DT_BUILDER"] = "persistent"
    Ws = qil.build_dt_mpo_batch(n, wrs)
    err = 0.0
    for W, w in zip(Ws, wrs):
        ref = dense_mpo(O.build_dt_mpo(n, w).data)
        err = max(err, np.abs(dense_mpo(W.to_host()) - ref).max())
    print(f"n={n} dense err vs oracle {err:.2e} bonds {Ws[-1].bond_dims}", flush=True)
for n in (8, 10, 12, 16):
    for cutoff in (1e-14, 1e-15):
        os.environ["QIL_DT_BUILDER"] = "persistent"
        Wp = qil.build_dt_mpo_batch(n, wrs, cutoff=cutoff, maxdim=None)
        os.environ["QIL_DT_BUILDER"] = "launches"
        Wl = [qil.build_dt_mpo_batch(n, [w], cutoff=cutoff, maxdim=None)[0] for w in wrs]
        same = [a.bond_dims == b.bond_dims for a, b in zip(Wp, Wl)]
        print(f"n={n} cutoff={cutoff:g} bond dims equal launches: {same} max {max(Wp[-1].bond_dims)}", flush=True)
        if not all(same):
            for a, b in zip(Wp, Wl):
                if a.bond_dims != b.bond_dims:
                    print("  P", a.bond_dims); print("  L", b.bond_dims)
# operator comparison through a state at n = 10
n = 10
x = O.generate_signal(n, kind="sin_decay", freq=[1.0, 2.5], decay_rate=[0.08, 0.03])
psi = qil.signal_ztmps(x, cutoff=1e-14)
bits = np.random.default_rng(3).integers(0, 2, size=(512, 2 * n)).astype(np.uint8)
os.environ["QIL_DT_BUILDER"] = "persistent"
Wp = qil.build_dt_mpo_batch(psi, wrs)
os.environ["QIL_DT_BUILDER"] = "launches"
Wl = qil.build_dt_mpo_batch(psi, wrs)
for a, b in zip(Wp, Wl):
    ca, cb = qil.coefficient_batch(a * psi, bits), qil.coefficient_batch(b * psi, bits)
    print(f"  n=10 coefficient diff persistent vs launches {np.abs(ca - cb).max():.2e} (scale {np.abs(cb).max():.2e})")
os.environ["QIL_DT_BUILDER"] = "persistent"
for nb in (1, 8, 64, 256):
    sig = np.linspace(0.25, 16.0, nb)
    qil.build_dt_mpo_batch(24, sig); ctx.synchronize()
    os.environ["QIL_DT_PROFILE"] = "1" if nb == 64 else ""
    if nb != 64: os.environ.pop("QIL_DT_PROFILE")
    t0 = time.perf_counter(); Ws = qil.build_dt_mpo_batch(24, sig); ctx.synchronize()
    dt = time.perf_counter() - t0
    os.environ.pop("QIL_DT_PROFILE", None)
    print(f"n=24 batch {nb}: {dt*1e3:.1f} ms  max bond {max(max(W.bond_dims) for W in Ws)}", flush=True)
os.environ["QIL_DT_BUILDER"] = "launches"
sig = np.linspace(0.25, 16.0, 64)
qil.build_dt_mpo_batch(24, sig); ctx.synchronize()
t0 = time.perf_counter(); qil.build_dt_mpo_batch(24, sig); ctx.synchronize()
print(f"n=24 batch 64 launches: {(time.perf_counter()-t0)*1e3:.1f} ms")
