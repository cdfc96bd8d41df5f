import sys, os, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil
print("f64_4096", qil.gemm_device_time(4096, 4096, 4096, np.float64, reps=3))
print("c64_4096", qil.gemm_device_time(4096, 4096, 4096, np.complex128, reps=3))
print("c64_coeff", qil.gemm_device_time(64, 16384, 8192, np.complex128, reps=3))
print("f64_sketch", qil.gemm_device_time(32768, 133, 32768, np.float64, "T", "N", reps=3))
