#!/bin/bash
# Subset of _alt_paths.sh: the knobs of the QR / SVD paths only (skips the torch-importing test, whose first import on a
# fresh box takes minutes).
run() { echo "== $*"; env "$@" timeout 600 python -m pytest tests -m gpu -q -x -k "not device_resident" 2>&1 | tail -1; }
run QIL_BJ_MIN=128
run QIL_RT_MIN=17
run QIL_RT_MIN=100000
run QIL_SVD_BLOCK_ROUNDS=0
run QIL_SVD_BB=4
run QIL_QR_LDS=0
run QIL_SVD_NEGLIGIBLE=0 QIL_MPO_GAUGE_QR=0
run QIL_SVD_QR_RATIO=2
run QIL_QR_PANEL=16
run QIL_BJ_TWO_SIDED=0 QIL_BJ_INNER=2
run QIL_SVD_A_LDS=0 QIL_JACOBI_EARLY=0
run QIL_TSQR_MIN_ROWS=8192 QIL_TSQR_MIN_CHUNK=2048
