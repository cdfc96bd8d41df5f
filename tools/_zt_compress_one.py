"""The final compression of build_zt_mpo alone (zt_transformer.jl:104 on the bond-136 product at n = 24), for kernel traces:
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 $R/tools/_zt_compress_one.py 24 3
  QIL_TIMELINE_MARKER=mpo_compose_site python3 tools/_chain_timeline.py DIR 0.5 0.52"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = qil.default_context()
dts = qil.build_dt_mpo_batch(n, [2 * np.pi])
Q = qil.zt_qft_chain_device(n, dts[0].site_ids)
for rep in range(reps):
    P = qil.apply(dts[0], Q)
    ctx.synchronize()
    b0 = P.bond_dims
    t0 = time.perf_counter()
    qil.mpo_compress(P, "down", 1e-14, 1000)
    ctx.synchronize()
    print("zt product compress n=%d: bond %d -> %d: %.2f ms" % (n, max(b0), max(P.bond_dims), 1e3 * (time.perf_counter() - t0)), flush=True)
    if rep == 0:
        print("product bonds:", b0)
        print("final bonds:  ", P.bond_dims, flush=True)
