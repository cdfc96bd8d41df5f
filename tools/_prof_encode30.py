"""signal_ztmps(:rsvd, k=128, p=5, q=2) of 2^30 i.i.d. normal samples generated in HBM; argv[1] = repetitions (default 2).
QIL_RSVD_DEBUG=1 prints the stages of the root split."""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import qilaplace_jl_amd as qil
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
g = torch.Generator(device="cuda"); g.manual_seed(30)
xd = torch.randn(2 ** 30, dtype=torch.float64, device="cuda", generator=g)
torch.cuda.synchronize()
for _ in range(reps):
    t0 = time.perf_counter()
    psi = qil.signal_ztmps(xd, method="rsvd", k=128, p=5, q=2, cutoff=1e-12, maxdim=128)
    qil.default_context().synchronize()
    print("encode n=30 random k=128: %.3f s" % (time.perf_counter() - t0), flush=True)
    del psi
