#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for nb in 8 16 32; do echo "--- $nb chains syncflag on"; timeout 300 python3 tools/_compress_concurrent.py $nb 256 2>&1 | tail -1 | cut -c1-110; echo "--- $nb chains syncflag off"; QIL_LOCKSTEP_NO_SYNCFLAG=1 timeout 300 python3 tools/_compress_concurrent.py $nb 256 2>&1 | tail -1 | cut -c1-110; done
