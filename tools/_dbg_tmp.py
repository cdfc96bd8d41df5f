import sys; sys.path.insert(0,'.')
import numpy as np, qilaplace_jl_amd as qil
Ws = qil.build_dt_mpo_batch(24, [2*np.pi])
