import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
rng = np.random.default_rng(4)
for n in (100, 112, 118, 120, 122, 127, 128, 129, 133, 135, 144, 160, 192, 200, 256, 270):
    A = np.asfortranarray(rng.standard_normal((n, n)))
    qil.svd_trunc(A, cutoff=None)
    t0 = time.perf_counter()
    for _ in range(5): qil.svd_trunc(A, cutoff=None)
    print(n, round(1e3 * (time.perf_counter() - t0) / 5, 2), "ms", flush=True)
