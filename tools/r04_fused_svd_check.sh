#!/bin/bash
# after a change of the one-workgroup in-LDS Jacobi (jacobi_fused_k): where it is on the critical path + the tests that cover it
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for k in dt zt; do timeout 600 python3 tools/_apply_compress_batch64.py 64 $k 2>&1 | tail -1; done
timeout 300 python3 tools/_compress_one.py 64 f64 3 2>&1 | tail -1
timeout 300 python3 tools/_compress_one.py 64 c64 3 2>&1 | tail -1
timeout 300 python3 tools/_compress_one.py 128 c64 3 2>&1 | tail -1
timeout 300 python3 tools/_apply_compress_one.py 2>&1 | tail -1
timeout 300 python3 tools/_exact_compress_time.py 3 2>&1 | tail -1
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3
