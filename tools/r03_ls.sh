#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python3 -m pytest tests -m gpu -x -q -k "batch" 2>&1 | tail -5
echo "--- lockstep on"; QIL_BATCH_DEBUG=1 timeout 200 python3 tools/_compress_concurrent.py 8 256 2>&1 | grep -v "slot" | tail -4
echo "--- lockstep off"; QIL_BATCH_LOCKSTEP=0 timeout 200 python3 tools/_compress_concurrent.py 8 256 2>&1 | tail -1
