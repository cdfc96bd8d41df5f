#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
cat > /tmp/fused_only.py <<'PY'
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import qilaplace_jl_amd as qil
ctx = qil.default_context()
n, N = 24, 2 ** 24
j = np.arange(N, dtype=np.float64)
x = np.sin(2 * np.pi * 5.0 * j / N) * np.exp(-3.0 * j / N) + 0.5 * np.cos(2 * np.pi * 11.0 * j / N)
rng = np.random.default_rng(1001)
x = x + sum(0.1 * rng.random() * np.sin(40.0 * (rng.random() - 0.5) * j / N) for _ in range(6))
psi = qil.signal_ztmps(x, method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
W = qil.build_zt_mpo(psi, 2 * np.pi)
mode = sys.argv[1]
for rep in range(3):
    ctx.synchronize(); t0 = time.perf_counter()
    if mode == "fused":
        f = qil.apply_compress(W, psi, maxdim=64, tol=1e-8)
    else:
        f = qil.compress(W * psi, maxdim=64, tol=1e-8)
    ctx.synchronize(); print(mode, (time.perf_counter() - t0) * 1e3, "ms", flush=True)
PY
for m in fused exact; do
rm -rf /tmp/prof_$m; mkdir -p /tmp/prof_$m
rocprofv3 --kernel-trace --stats -d /tmp/prof_$m --output-format csv -- python3 /tmp/fused_only.py $m > /tmp/prof_$m/log 2>&1
f=$(find /tmp/prof_$m -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/${m}_kernel_stats.csv
grep " ms" /tmp/prof_$m/log
done
