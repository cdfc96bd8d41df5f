#!/bin/bash
# exact compress!(apply) of the bond-1008 zT product: per-site SVD phases (QIL_SVD_DEBUG) and kernel classes of one repetition
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R; mkdir -p $O
timeout 300 python3 tools/_exact_svd_paths.py > $O/r04_exact_svd_paths.txt 2>&1
grep -c "svd-left" $O/r04_exact_svd_paths.txt
timeout 600 bash tools/r03_exact_timeline.sh > $O/r04_exact_timeline.txt 2>&1; cat $O/r04_exact_timeline.txt | head -60
