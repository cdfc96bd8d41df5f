#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -6 $O/r03_pytest_gpu.log
echo "--- lockstep on"; QIL_BATCH_DEBUG=1 timeout 200 python3 tools/_compress_concurrent.py 8 256 2>&1 | grep -v "slot" | tail -2
echo "--- lockstep off"; QIL_BATCH_LOCKSTEP=0 timeout 200 python3 tools/_compress_concurrent.py 8 256 2>&1 | tail -1
timeout 300 python3 tools/_compress_time.py 2>&1 | tail -6
