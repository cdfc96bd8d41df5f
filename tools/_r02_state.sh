#!/bin/bash
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests -x -q -m gpu -k "concurrent_subtrees or rsvd or encode or signal" 2>&1 | tail -3
for d in 0 1 2 3 4; do echo "QIL_ENCODE_PAR_DEPTH=$d"; QIL_ENCODE_PAR_DEPTH=$d python $R/tools/_prof_encode30.py 2>&1 | tail -1; done
