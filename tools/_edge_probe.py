"""Unusual-parameter probe of the truncating entry points against the oracle (small problems, gauge-invariant comparison):
cutoff / tol = 0, maxdim = 1, no cap, sketches wider than the operand.  Prints one line per case; nothing here is timed.
(r06: the same kind of probe found the cutoff = 0 bug of the device DT builders.)"""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import qilaplace_jl_amd as qil
import oracle as O
from helpers import random_mps_data, random_mpo_data, saturated_profile, dense_mpo

rng = np.random.default_rng(606)
bad = 0


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def report(name, err, tol, extra=""):
    global bad
    ok = err <= tol
    bad += not ok
    print(("ok  " if ok else "BAD ") + f"{name}: {err:.2e} (tol {tol:.0e}) {extra}", flush=True)


L = 10
for dt in (np.float64, np.complex128):
    a = random_mps_data(saturated_profile(L, 12), rng, dtype=dt)
    bits = rng.integers(0, 2, size=(256, L))
    ref = O.coefficient_batch(O.SignalMPS(a, amplitude=1.3), bits)
    for kw in (dict(maxdim=None, tol=0.0), dict(maxdim=10 ** 9, tol=1e-300), dict(maxdim=1, tol=1e-12), dict(maxdim=3, tol=0.0)):
        psi = qil.SignalMPS(a, amplitude=1.3)
        po = O.SignalMPS([t.copy() for t in a], amplitude=1.3)
        try:
            qil.compress(psi, **{k: v for k, v in kw.items() if v is not None})
            O.compress(po, **{k: v for k, v in kw.items() if v is not None})
            report(f"compress {np.dtype(dt).name} {kw}", rel(qil.coefficient_batch(psi, bits), O.coefficient_batch(po, bits)), 1e-9,
                   f"bonds {max(psi.bond_dims)} / {max(po.bond_dims)}")
        except Exception as e:                                # noqa: BLE001
            print("EXC ", "compress", kw, type(e).__name__, str(e)[:100]); bad += 1
    for direction in ("left", "right"):
        psi = qil.SignalMPS(a, amplitude=1.3)
        qil.canonicalize(psi, direction, cutoff=0.0)
        report(f"canonicalize {direction} cutoff=0 {np.dtype(dt).name}", rel(qil.coefficient_batch(psi, bits), ref), 1e-12)
# MPO compression with cutoff 0 keeps the operator
w = random_mpo_data(saturated_profile(6, 9, base=4), rng)
for direction in ("down", "up"):
    W = qil.SingleSiteMPO([t.copy() for t in w])
    qil.mpo_compress(W, direction, cutoff=0.0, maxdim=None)
    report(f"mpo_compress {direction} cutoff=0", float(np.abs(dense_mpo(W.to_host()) - dense_mpo(w)).max() / np.abs(dense_mpo(w)).max()), 1e-12, f"bonds {W.bond_dims}")
    W = qil.SingleSiteMPO([t.copy() for t in w])
    qil.mpo_compress(W, direction, cutoff=1e-30, maxdim=1)
    report(f"mpo_compress {direction} maxdim=1 runs", 0.0, 1.0, f"bonds {W.bond_dims}")
# encoders
n = 9
x = rng.standard_normal(2 ** n) * np.exp(-0.01 * np.arange(2 ** n))
for kw in (dict(method="svd", cutoff=0.0), dict(method="svd", cutoff=1e-300, maxdim=10 ** 6), dict(method="rsvd", k=600, p=10, q=1, cutoff=0.0),
           dict(method="rsvd", k=40, p=0, q=0, cutoff=1e-15)):
    psi = qil.signal_mps(x, **kw)
    report(f"signal_mps {kw}", float(np.abs(qil.mps_to_vector(psi) - x).max() / np.abs(x).max()), 1e-10, f"bonds {max(psi.bond_dims)}")
zt = qil.signal_ztmps(x, cutoff=0.0)
js = rng.integers(0, 2 ** n, size=128)
bz = np.zeros((128, 2 * n), dtype=np.uint8)                       # main bits = copy bits = j (elsewhere the paired state is exactly 0)
bz[:, 0::2] = bz[:, 1::2] = (js[:, None] >> np.arange(n - 1, -1, -1)[None, :]) & 1
report("signal_ztmps cutoff=0 (coefficients at main = copy = j vs x_j)", float(np.abs(qil.coefficient_batch(zt, bz) - x[js]).max() / np.abs(x).max()), 1e-12)
# fused route at the extremes
a = random_mps_data(saturated_profile(8, 6), rng)
w = random_mpo_data(saturated_profile(8, 5, base=4), rng)
bits = rng.integers(0, 2, size=(128, 8))
for kw in (dict(maxdim=1, tol=1e-10), dict(maxdim=10 ** 6, tol=0.0), dict(maxdim=4, tol=1e-3)):
    out = qil.apply_compress(qil.SingleSiteMPO(w), qil.SignalMPS(a), **kw)
    po = O.apply(O.SingleSiteMPO(w), O.SignalMPS(a))
    full = O.coefficient_batch(po, bits)
    O.compress(po, **kw)
    e_h, e_o = rel(qil.coefficient_batch(out, bits), full), rel(O.coefficient_batch(po, bits), full)
    report(f"apply_compress {kw}: error vs exact product {e_h:.2e} (oracle route {e_o:.2e})", max(e_h - 2.0 * e_o - 1e-9, 0.0), 1e-9, f"bonds {max(out.bond_dims)} / {max(po.bond_dims)}")
# plain SVDs
A = rng.standard_normal((70, 40)) @ np.diag(np.logspace(0, -18, 40)) @ rng.standard_normal((40, 55))
for kw in (dict(cutoff=0.0), dict(cutoff=None), dict(cutoff=1e-30, maxdim=5), dict(cutoff=1e-12, mindim=30)):
    U, S, Vh = qil.svd_trunc(A, **kw)
    Uo, So, Vo = O.svd_trunc(A, **kw) if hasattr(O, "svd_trunc") else (None, np.linalg.svd(A, compute_uv=False)[:len(S)], None)
    report(f"svd_trunc {kw}", float(np.abs((U * S) @ Vh - (A if kw.get("maxdim") is None else (Uo * So) @ Vo if Uo is not None else A)).max() / np.abs(A).max()), 1e-10 if kw.get("cutoff") in (0.0, None, 1e-12) else 1.0,
           f"rank {len(S)} / {len(So)}")
print("edge probe:", bad, "bad")
