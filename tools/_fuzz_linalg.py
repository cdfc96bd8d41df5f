"""One-off wide fuzz of the device linear algebra behind the hot path (GEMM, QR) against numpy: shapes around every
tile / panel / LDS-fit boundary, all op combinations, both dtypes.  gpurun -- python tools/_fuzz_linalg.py [seeds]"""
import os, sys
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dims = [1, 2, 3, 15, 16, 17, 31, 32, 33, 55, 63, 64, 65, 96, 97, 127, 128, 129, 133, 143, 144, 145, 255, 256, 257, 300, 511, 512,
        513, 1000, 1023, 1024, 1025, 1170, 2047, 2048, 2049, 3000, 4099]
bad = 0
def cx(rng, shape, c):
    A = rng.standard_normal(shape)
    return A + 1j * rng.standard_normal(shape) if c else A
for seed in range(nseeds):
    rng = np.random.default_rng(seed)
    # ---- GEMM
    m, n, k = (int(rng.choice(dims[:34])) for _ in range(3))
    opA, opB = str(rng.choice(list("NTHC"))), str(rng.choice(list("NTHC")))
    ca, cb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    A = cx(rng, (m, k) if opA in "NC" else (k, m), ca)
    B = cx(rng, (k, n) if opB in "NC" else (n, k), cb)
    f = {"N": lambda X: X, "T": lambda X: X.T, "H": lambda X: X.conj().T, "C": lambda X: X.conj()}
    ref = f[opA](A) @ f[opB](B)
    got = qil.gemm(A, B, opA, opB)
    e = np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-300)
    if e > 1e-12 * max(1, np.sqrt(k)):
        bad += 1; print("GEMM BAD", (seed, m, n, k, opA, opB, ca, cb), "%.2e" % e, flush=True)
    # ---- QR
    n2 = int(rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 48, 55, 64, 65, 100, 133, 144, 200, 256, 300]))
    m2 = max(n2, int(rng.choice(dims)))
    c2 = bool(rng.integers(0, 2))
    kind = str(rng.choice(["rand", "rand", "lowrank", "dupcols", "zerocol"]))
    A = cx(rng, (m2, n2), c2)
    if kind == "lowrank" and n2 > 3:
        r = max(1, n2 // 3); A = A[:, :r] @ cx(rng, (r, n2), c2)
    elif kind == "dupcols" and n2 > 3:
        A[:, n2 // 2:] = A[:, :n2 - n2 // 2]
    elif kind == "zerocol":
        A[:, n2 // 2] = 0
    Q, R = qil.qr_positive(A)
    G = Q.conj().T @ Q; d = np.real(np.diag(G))
    e1 = np.abs(Q @ R - A).max() / max(np.abs(A).max(), 1e-300)
    e2 = np.abs(G - np.diag(d)).max()
    e3 = np.abs(np.where(d > 0.5, d - 1, d)).max()
    ok = e1 < 1e-11 and e2 < 1e-11 and e3 < 1e-12 and np.abs(np.tril(R, -1)).max() <= 1e-13 * max(np.abs(R).max(), 1e-300) and np.real(np.diag(R)).min() >= 0
    if kind == "rand": ok = ok and (d > 0.5).all()
    if not ok:
        bad += 1; print("QR BAD", (seed, m2, n2, c2, kind), "recon %.2e offd %.2e diag %.2e kept %d" % (e1, e2, e3, (d > 0.5).sum()), flush=True)
print("fuzz done: %d seeds, %d bad" % (nseeds, bad))
