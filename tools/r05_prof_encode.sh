#!/bin/bash
# rocprofv3 kernel stats of signal_ztmps(:rsvd, k=128) of 2^30 i.i.d. samples generated in HBM (two encodes)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/prof; mkdir -p $O/prof
rocprofv3 --kernel-trace --stats -d $O/prof/c --output-format csv -- python3 $R/tools/_prof_encode30.py > $O/prof/c.log 2>&1
f=$(find $O/prof/c -name '*kernel_stats.csv' | head -1)
tail -2 $O/prof/c.log; python3 $R/tools/_kstats.py $f 40
cp $f $O/r05_kernel_stats_encode_n30.csv
rm -rf $O/prof
