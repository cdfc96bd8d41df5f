"""One-off fuzz: compress!(apply(W, psi)) and apply_compress on random products whose bonds (D chi = 100..500) are
rank-deficient the way real product bonds are, against the CPU oracle's compress of the same product: bond dimensions,
amplitude and sampled coefficients.  gpurun -- python tools/_fuzz_product_compress.py [cases]"""
import os, sys, time
import numpy as np
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tests"))
import qilaplace_jl_amd as qil
import oracle as O
from helpers import random_mps_data, random_mpo_data, saturated_profile
ncase = int(sys.argv[1]) if len(sys.argv) > 1 else 24
bad = 0
for case in range(ncase):
    rng = np.random.default_rng(7000 + case)
    L = int(rng.integers(8, 13))
    chi = int(rng.choice([8, 12, 16, 24, 32]))
    D = int(rng.choice([6, 8, 12, 16]))
    adt = np.complex128 if rng.random() < 0.4 else np.float64
    wdt = np.complex128 if rng.random() < 0.6 else np.float64
    a = random_mps_data(saturated_profile(L, chi), rng, dtype=adt)
    w = random_mpo_data(saturated_profile(L, D, base=4), rng, dtype=wdt)
    maxdim = int(rng.choice([8, 16, 32, 64, 128] if os.environ.get("QIL_FUZZ_WIDE") else [8, 16, 32, 64]))
    tol = float(rng.choice([1e-6, 1e-8, 1e-10]))
    psi, W = qil.SignalMPS(a, amplitude=1.7), qil.SingleSiteMPO(w)
    ref = O.apply(O.SingleSiteMPO(w), O.SignalMPS([t.copy() for t in a], amplitude=1.7))
    bits = rng.integers(0, 2, size=(128, L))
    before = O.coefficient_batch(ref, bits)
    scale = np.abs(before).max()
    prod = W * psi
    pb = max(prod.bond_dims)
    qil.compress(prod, maxdim=maxdim, tol=tol)
    O.compress(ref, maxdim=maxdim, tol=tol)
    got, want = qil.coefficient_batch(prod, bits), O.coefficient_batch(ref, bits)
    e1 = np.abs(got - want).max() / scale
    zc = os.environ.get("QIL_FUZZ_ZIP")            # cap of the zip-up under test: "plus16" = maxdim + 16, "f125" = max(maxdim + 16, 1.25 maxdim); default: the library's
    zip_cap = None if not zc else (maxdim + 16 if zc == "plus16" else max(maxdim + 16, int(np.ceil(1.25 * maxdim))))
    fused = qil.apply_compress(W, psi, maxdim=maxdim, tol=tol, zip_maxdim=zip_cap)
    e2 = np.abs(qil.coefficient_batch(fused, bits) - want).max() / scale
    etr = np.abs(want - before).max() / scale                     # the truncation's own error: fused may differ by that much
    # verdict: compress!(apply) against the oracle (the parity claim), and -- since r02 (sketched zip-up + variational
    # sweep) -- the fused route too: same bond dimensions, and no further from the oracle's truncated state than
    # the truncation's own error (so at most 2x that from the exact product)
    ok = prod.bond_dims == ref.bond_dims and abs(prod.amplitude - ref.amplitude) < 1e-8 * ref.amplitude and e1 < 1e-7 \
        and fused.bond_dims == ref.bond_dims and e2 <= etr + 1e-9
    if not ok:
        bad += 1
    print("ok " if ok else "BAD", dict(case=case, L=L, chi=chi, D=D, product_bond=pb, maxdim=maxdim, tol=tol, adt=adt.__name__, wdt=wdt.__name__,
          bonds_equal=prod.bond_dims == ref.bond_dims, e_compress="%.1e" % e1, e_fused="%.1e" % e2, e_trunc="%.1e" % etr), flush=True)
print("product-compress fuzz: %d cases, %d bad" % (ncase, bad))
