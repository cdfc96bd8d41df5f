#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python3 -m pytest tests -m gpu -x -q -k "batch" 2>&1 | tail -3
for nb in 8 16 32 64; do for m in -1; do echo "--- $nb chains, QIL_BATCH_LOCKSTEP=$m"; QIL_BATCH_LOCKSTEP=$m timeout 300 python3 tools/_compress_concurrent.py $nb 256 2>&1 | tail -1 | cut -c1-130; done; done
echo "--- 32 chains chi 64 auto"; timeout 200 python3 tools/_compress_concurrent.py 32 64 2>&1 | tail -1 | cut -c1-130
echo "--- 32 chains chi 64 off"; QIL_BATCH_LOCKSTEP=0 timeout 200 python3 tools/_compress_concurrent.py 32 64 2>&1 | tail -1 | cut -c1-130
