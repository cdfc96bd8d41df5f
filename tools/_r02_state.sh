#!/bin/bash
run() { echo "== $*"; env "$@" timeout 1200 python -m pytest tests -m gpu -q -x -k "mpo_compress or zt or svd_trunc_low_rank" 2>&1 | grep -E "^(FAILED|E  )|passed|failed" | head -6 | cut -c1-200; }
run QIL_SVD_NEGLIGIBLE=0
run QIL_MPO_GAUGE_QR=0
run A=1
