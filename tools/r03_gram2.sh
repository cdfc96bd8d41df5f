#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
echo "--- gram on"; timeout 300 python3 tools/_compress_time.py 2>&1 | tail -6
echo "--- gram off"; QIL_SVD_GRAM=0 timeout 300 python3 tools/_compress_time.py 2>&1 | tail -6
timeout 900 python3 -m pytest tests -m gpu -x -q -k "svd or compress or canonic or fuzz or trunc or gauge or rsvd or signal or config5" 2>&1 | tail -4
