#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 900 python3 -m pytest tests -m gpu -x -q -k "qr or svd or compress or canonic or fuzz or trunc or gauge or rsvd or signal or mpo_compress or zt or dt" 2>&1 | tail -6
echo "--- chol on"; timeout 300 python3 tools/_compress_time.py 2>&1 | tail -6
echo "--- chol off"; QIL_QR_CHOL=0 timeout 300 python3 tools/_compress_time.py 2>&1 | tail -6
