#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q > $O/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -4 $O/r03_pytest_gpu.log
timeout 300 python3 tools/_compress_time.py 2>&1 | tail -6
timeout 200 python3 tools/_apply_compress_batch_time.py 2>&1 | tail -3
