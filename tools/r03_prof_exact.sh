#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O/prof
rocprofv3 --kernel-trace --stats -d $O/prof/e --output-format csv -- python3 $R/tools/_exact_compress_time.py 3 > $O/prof/e.log 2>&1
f=$(find $O/prof/e -name '*kernel_stats.csv' | head -1)
tail -1 $O/prof/e.log; python3 $R/tools/_kstats.py $f 30
rm -rf $O/prof
cd $R; QIL_SVD_DEBUG=1 python3 tools/_exact_compress_time.py 1 2>&1 | grep -v "^\[qr\]" | tail -n 900 > $O/r03_exact_debug.log; wc -l $O/r03_exact_debug.log
