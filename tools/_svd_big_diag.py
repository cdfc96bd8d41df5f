import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
rng = np.random.default_rng(5)
for (m, n, cx) in [(1335, 2670, True), (1335, 2670, False), (900, 1800, True)]:
    A = rng.standard_normal((m, n)) + (1j * rng.standard_normal((m, n)) if cx else 0)
    qil.svd_trunc(A, cutoff=1e-12)
    os.environ["QIL_SVD_DEBUG"] = "1"
    t0 = time.perf_counter(); qil.svd_trunc(A, cutoff=1e-12); dt = time.perf_counter() - t0
    os.environ.pop("QIL_SVD_DEBUG")
    print(f"=== {m}x{n} {'c64' if cx else 'f64'}: {dt*1e3:.0f} ms incl. PCIe", flush=True)
