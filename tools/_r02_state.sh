#!/bin/bash
R=$GRAFT_REPO_ROOT
$R/tools/micro/jacobi_round_cost.bin
for m in 64 32; do
  echo "== QIL_SVD_LEFT_MODE=$m"
  QIL_SVD_LEFT_MODE=$m python $R/tools/_compress_time.py 2>&1 | grep compress
done
python -m pytest $R/tests -x -q -m gpu -k "svd or compress or canonical or trunc" 2>&1 | tail -3
