#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 900 python3 -m pytest tests -m gpu -x -q -k "svd or compress or canonic or fuzz or trunc or gauge or rsvd or signal" > $O/r03_pytest_gram.log 2>&1; echo "pytest rc=$?"
tail -6 $O/r03_pytest_gram.log
echo "--- gram on"; timeout 300 python3 tools/_compress_time.py 2>&1 | tail -8
echo "--- gram off"; QIL_SVD_GRAM=0 timeout 300 python3 tools/_compress_time.py 2>&1 | tail -8
echo "--- debug one"; QIL_SVD_DEBUG=1 timeout 100 python3 tools/_compress_one.py 256 f64 1 2>&1 | grep -v "^\[svd-cert\]" | head -60
