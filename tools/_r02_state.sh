#!/bin/bash
R=$GRAFT_REPO_ROOT
for v in 128 96 192; do
echo "QIL_GEMM_TINY_SPLIT_K=$v"
QIL_GEMM_TINY_SPLIT_K=$v python $R/tools/_truncate_block.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v,1) for k,v in d.items() if k in ('fused_apply_compress_ms','exact_compress_ms','compress_chi256_to_128_24_sites_ms')})"
QIL_GEMM_TINY_SPLIT_K=$v python $R/tools/_compress_time.py 2>&1 | grep compress | head -4
done
