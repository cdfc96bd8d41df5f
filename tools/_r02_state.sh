#!/bin/bash
R=$GRAFT_REPO_ROOT
t() { echo "== $*"; env "$@" python $R/tools/_truncate_block.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v,1) for k,v in d.items() if k in ('fused_apply_compress_ms','exact_compress_ms','compress_chi256_to_128_24_sites_ms')})"; env "$@" python $R/tools/_prof_compress.py 64 2>&1 | tail -1; }
t QIL_QR_HH_SINGLE_MIN=17
t QIL_QR_HH_SINGLE_MIN=100
t QIL_QR_HH_SINGLE_MIN=9
python -m pytest $R/tests -x -q -m gpu -k "qr or svd" 2>&1 | tail -2
