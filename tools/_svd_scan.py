import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
rng = np.random.default_rng(4)
for (m, n) in [(120, 120), (270, 270), (540, 270), (256, 128), (400, 400)]:
    A = np.asfortranarray(rng.standard_normal((m, n)))
    qil.svd_trunc(A, cutoff=None)
    t0 = time.perf_counter()
    for _ in range(5): qil.svd_trunc(A, cutoff=None)
    print((m, n), round(1e3 * (time.perf_counter() - t0) / 5, 2), "ms", flush=True)
