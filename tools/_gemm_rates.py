"""f64 / c64 MFMA GEMM rates on device-resident operands (qil_gemm_device_time): the shapes the chains and read-outs meet."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import qilaplace_jl_amd as qil
shapes = [(4096, 4096, 4096), (64, 16384, 8192), (32, 8192, 8192), (48, 8192, 8192), (1008, 1008, 1008), (256, 256, 256), (32768, 133, 32768)]
for dt, name, fl in ((np.float64, "f64", 2.0), (np.complex128, "c64", 8.0)):
    for (m, n, k) in shapes:
        if name == "c64" and m * k > 2 ** 29:
            continue
        ms = qil.gemm_device_time(m, n, k, dtype=dt, reps=5)
        print(f"{name} {m} x {n} x {k}: {ms:.3f} ms = {fl * m * n * k / ms / 1e9:.1f} TFLOP/s (conventional flop count)", flush=True)
