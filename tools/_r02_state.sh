#!/bin/bash
R=$GRAFT_REPO_ROOT
$R/tools/micro/jacobi_round_cost.bin | grep -v "mode=2"
python $R/tools/_compress_time.py 2>&1 | grep compress
python -m pytest $R/tests -x -q -m gpu -k "svd or compress or canonical or trunc" 2>&1 | tail -3
