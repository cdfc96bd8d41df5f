import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
print("real", qil.gemm_device_time(4096, 4096, 4096, np.float64, reps=3))
print("cplx", qil.gemm_device_time(4096, 4096, 4096, np.complex128, reps=3))
