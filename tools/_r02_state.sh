#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof; mkdir -p $O/prof
rocprofv3 --kernel-trace --stats -d $O/prof/c256 --output-format csv -- python3 $R/tools/_prof_compress.py 256 > $O/prof/c256.log 2>&1
f=$(find $O/prof/c256 -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" $O/c256_kernel_stats.csv
grep compress $O/prof/c256.log
rm -rf $O/prof
