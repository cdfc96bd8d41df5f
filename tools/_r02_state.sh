#!/bin/bash
# The GPU parity suite under tuning knobs that force the alternative code paths.
run() { echo "== $*"; env "$@" timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -2; }
run QIL_BJ_MIN=128
run QIL_COEFF_GEMM_MINCHI=1 QIL_LAZY_GEMM_MIN=1
run QIL_GEMM_CFG=0
run QIL_GEMM_CFG=3
run QIL_GEMM_CFG=1
run QIL_APPLY_VARIANT=0
run QIL_APPLY_VARIANT=2
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
run QIL_SVD_FUSED_GLOBAL=1
run QIL_GEMM_XCD=0 QIL_GEMM_CFG=4
run QIL_TSQR_MIN_ROWS=8192 QIL_TSQR_MIN_CHUNK=2048
run QIL_SYSTEM_HIP=1
