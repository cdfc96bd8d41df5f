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
run QIL_SVD_NEGLIGIBLE=0
run QIL_SVD_QR_RATIO=2
run QIL_QR_PANEL=16
run QIL_BJ_TWO_SIDED=0 QIL_BJ_INNER=2
run QIL_SVD_A_LDS=0 QIL_JACOBI_EARLY=0
run QIL_SVD_FUSED_GLOBAL=1
run QIL_GEMM_XCD=0 QIL_GEMM_CFG=4
run QIL_TSQR_MIN_ROWS=8192 QIL_TSQR_MIN_CHUNK=2048
# QIL_SYSTEM_HIP=1 is not in the list: with the system HIP runtime loaded first, the one test that imports torch (cfg5 generates
# its 2^30-sample signal in HBM through torch) finds no GPU in torch's own runtime -- the reason the shim preloads torch's copy
# round 2
run QIL_SVD_LEFT_MODE=32
run QIL_SVD_LEFT=0
run QIL_SVD_LEFT_QR2=1 QIL_SVD_LOWRANK=0
run QIL_QR_HH=0 QIL_TSQR_NOFIT_ROWS=100000
run QIL_GEMM_SPLIT_MIN_K=256 QIL_GEMM_FILL_SPLIT=0
run QIL_BATCH_WORKERS=1 QIL_ENCODE_PAR_DEPTH=0 QIL_SWEEP_CONCURRENT=0
run QIL_BATCH_WORKERS=3 QIL_ENCODE_PAR_DEPTH=5
run QIL_DT_BUILDER=launches QIL_ZIP_SKETCH=0
run QIL_DT_DCAP=24
run QIL_SVD_CERT=0
run QIL_QR_FUSED_MAX_N=100000 QIL_SVD_LEFT_MIN=97
run QIL_BATCH_COMBINE=1
run QIL_BATCH_COMBINE=1 QIL_BATCH_COMBINE_WAIT_US=1000
run QIL_SVD_LEFT_QR2_GRADE=0
run QIL_SVD_LEFT_QR2_GRADE=1e8
run QIL_SVD_LEFT_QR2_GRADE_MAX=1e300
