#!/bin/bash
# QIL_SVD_GRAM = 0 / 1 / 2 (vector rounds / Gram rounds for f64 / for c64 too): exact route of the bond-1008 product and compress! chi 128..512
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for g in 0 1 2; do
  echo "QIL_SVD_GRAM=$g"
  QIL_SVD_GRAM=$g timeout 300 python3 tools/_exact_compress_time.py 4 2>&1 | tail -1
  QIL_SVD_GRAM=$g timeout 300 python3 tools/_compress_time.py 2>&1 | grep complex
done
