"""Closed-form error of the cfg4 damping sweep per damping value (HIP, n = 24, 64 values): how smooth is it in sigma?
(sizes the per-sigma bound of tests/test_gpu_parity.py::test_config4_damping_sweep_full_size leg b)"""
import json
import sys

import numpy as np

sys.path.insert(0, ".")
import qilaplace_jl_amd as qil   # noqa: E402
import bench_configs             # noqa: E402

n, N = 24, 2 ** 24
x = bench_configs.cfg4_signal(n)
psi = qil.signal_ztmps(x, method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
sig = np.linspace(0.25, 16.0, 64)
bits, kk, jj = qil.damping_sample_bits(n, 1024, seed=7)
got = qil.damping_sweep(psi, sig, bits)
peak = np.abs(x).max() / np.sqrt(N)
errs = [float(np.abs(got[r] - x[jj] * np.exp(-s * kk * jj / N) / np.sqrt(N)).max() / peak) for r, s in enumerate(sig)]
print(json.dumps({"case": "cfg4_closed_form_error_by_sigma", "sigma": [float(s) for s in sig], "err_rel_peak": errs}))
