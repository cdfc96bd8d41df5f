"""SVD time over a grid of shapes / dtypes (host-call timing, incl. ~0.6 ms of PCIe + Python): looks for dispatch anomalies."""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
rng = np.random.default_rng(4)
def t_svd(m, n, c):
    A = np.asfortranarray(rng.standard_normal((m, n)) + (1j * rng.standard_normal((m, n)) if c else 0))
    qil.svd_trunc(A, cutoff=None)
    t0 = time.perf_counter()
    for _ in range(4): qil.svd_trunc(A, cutoff=None)
    return 1e3 * (time.perf_counter() - t0) / 4
base = t_svd(8, 8, 0)
print("baseline 8x8: %.2f ms" % base)
for c in (0, 1):
    for n in (16, 32, 48, 64, 80, 96, 112, 128, 160):
        row = []
        for mult in (1, 2, 4, 8, 32):
            m = n * mult
            row.append("%6.2f" % (t_svd(m, n, c) - base))
        print("cplx=%d n=%4d  m = n x (1, 2, 4, 8, 32): " % (c, n) + " ".join(row), flush=True)
