#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for ts in 4 2 1; do echo "--- tile $ts"; QIL_CHOL_TILE=$ts timeout 300 python3 tools/_compress_time.py 2>&1 | tail -6; done
