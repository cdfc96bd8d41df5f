#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for sh in 0 1 2; do echo "--- chol shape $sh (0: 2x2 tiles 32x32 threads, 1: 4x4 tiles 16x16 threads [64 cols], 2: 2x2 tiles 16x16 threads [32 cols])"; QIL_CHOL_SHAPE=$sh timeout 300 python3 tools/_compress_time.py 2>&1 | tail -6; done
QIL_CHOL_SHAPE=1 timeout 600 python3 -m pytest tests -m gpu -x -q -k "qr or svd or compress or canonic or fuzz or gauge" 2>&1 | tail -2
