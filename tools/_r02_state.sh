#!/bin/bash
R=$GRAFT_REPO_ROOT
for c in -1 4; do
echo "QIL_GEMM_CFG=$c"
QIL_GEMM_CFG=$c python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import qilaplace_jl_amd as qil
for (m, n, k, ta, tb) in [(32768, 133, 32768, "T", "N"), (32768, 133, 32768, "N", "N"), (16384, 133, 16384, "T", "N"), (32768, 144, 32768, "T", "N")]:
    ms = qil.gemm_device_time(m, n, k, np.float64, ta, tb, reps=3)
    print(m, n, k, ta, tb, round(ms, 3), "ms", round(2.0 * m * n * k / ms / 1e9, 1), "TFLOP/s", flush=True)
PY
done
