#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "--- default"; timeout 200 python3 tools/_compress_concurrent.py 8 256 2>&1 | tail -1
for q in 2 8 16; do echo "--- GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 200 python3 tools/_compress_concurrent.py 8 256 2>&1 | tail -1; done
echo "--- workers 4"; QIL_BATCH_WORKERS=4 timeout 200 python3 tools/_compress_concurrent.py 8 256 2>&1 | tail -1
echo "--- HIP_FORCE_DEV_KERNARG"; HIP_FORCE_DEV_KERNARG=1 timeout 200 python3 tools/_compress_concurrent.py 8 256 2>&1 | tail -1
echo "--- 16 chains"; timeout 200 python3 tools/_compress_concurrent.py 16 256 2>&1 | tail -1
