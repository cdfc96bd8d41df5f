"""Evidence that the two persistent builders of qil_build_zt_mpo_batch overlap (DT halves on the context's stream, paired QFT chain on a
worker stream): start / end of dt_build_persistent and chain_build_persistent from a rocprofv3 kernel trace of three builds.
  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 $R/tools/_zt_overlap_trace.py run
  python3 tools/_zt_overlap_trace.py DIR"""
import csv
import glob
import os
import sys

if sys.argv[1] == "run":
    sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import numpy as np
    import qilaplace_jl_amd as qil
    for _ in range(3):
        W = qil.build_zt_mpo(24, 2 * np.pi)
        qil.default_context().synchronize()
        del W
    sys.exit(0)
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
dt = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows if "dt_build_persistent" in r["Kernel_Name"]]
ch = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?")) for r in rows if "chain_build_persistent" in r["Kernel_Name"]]
for i, (d, c) in enumerate(zip(dt, ch)):
    t0 = min(d[0], c[0])
    ov = max(0, min(d[1], c[1]) - max(d[0], c[0]))
    print(f"build {i}: dt_build_persistent {(d[0]-t0)/1e6:7.3f} .. {(d[1]-t0)/1e6:7.3f} ms (queue {d[2]}), chain_build_persistent {(c[0]-t0)/1e6:7.3f} .. "
          f"{(c[1]-t0)/1e6:7.3f} ms (queue {c[2]}): overlap {ov/1e6:.3f} ms = {100.0*ov/max(c[1]-c[0],1):.1f} % of the chain kernel")
