#!/bin/bash
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests -x -q -m gpu -k "qr or svd or compress or canonical or trunc or rsvd or encode or signal" 2>&1 | tail -3
python $R/tools/_compress_time.py 2>&1 | grep compress
python $R/tools/_prof_encode30.py 2>&1 | tail -1
python $R/tools/_truncate_block.py 2>&1 | tail -3
