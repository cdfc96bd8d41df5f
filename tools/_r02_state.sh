#!/bin/bash
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests -x -q -m gpu -k "qr or svd or compress or canonical or trunc" 2>&1 | tail -2
python $R/tools/_truncate_block.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:round(v,1) for k,v in d.items() if k in ('fused_apply_compress_ms','exact_compress_ms')})"
python $R/tools/_compress_time.py 2>&1 | grep compress
