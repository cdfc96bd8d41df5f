#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for p in 50 200 1000 5000; do echo "--- patience $p"; QIL_LOCKSTEP_PATIENCE_US=$p QIL_BATCH_DEBUG=1 timeout 200 python3 tools/_compress_concurrent.py 8 256 2>&1 | grep -v "slot" | tail -2 | cut -c1-330; done
echo "--- 16 chains lockstep"; timeout 200 python3 tools/_compress_concurrent.py 16 256 2>&1 | tail -1
echo "--- 4 chains lockstep"; timeout 200 python3 tools/_compress_concurrent.py 4 256 2>&1 | tail -1
echo "--- 8 chains chi 64"; timeout 200 python3 tools/_compress_concurrent.py 8 64 2>&1 | tail -1
echo "--- 8 chains chi 64 off"; QIL_BATCH_LOCKSTEP=0 timeout 200 python3 tools/_compress_concurrent.py 8 64 2>&1 | tail -1
