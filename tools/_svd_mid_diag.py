import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil
ctx = qil.default_context()
rng = np.random.default_rng(0)
for (m, n, kind) in [(512, 256, "graded"), (512, 256, "random"), (256, 128, "graded"), (1008, 256, "graded")]:
    if kind == "graded":
        U, _ = np.linalg.qr(rng.standard_normal((m, n))); V, _ = np.linalg.qr(rng.standard_normal((n, n)))
        A = (U * np.logspace(0, -12, n)) @ V.T
    else:
        A = rng.standard_normal((m, n))
    qil.svd_trunc(A, cutoff=1e-20)
    os.environ["QIL_SVD_DEBUG"] = "1"
    t0 = time.perf_counter(); qil.svd_trunc(A, cutoff=1e-20); dt = time.perf_counter() - t0
    os.environ.pop("QIL_SVD_DEBUG")
    t0 = time.perf_counter()
    for _ in range(5): qil.svd_trunc(A, cutoff=1e-20)
    print(f"=== {m}x{n} {kind}: {(time.perf_counter()-t0)/5*1e3:.2f} ms per svd_trunc (host operands incl. PCIe)", flush=True)
