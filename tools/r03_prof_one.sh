#!/bin/bash
# rocprofv3 kernel stats of one compress! configuration: bash tools/r03_prof_one.sh <chi> <f64|c64> [rows]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O/prof
rocprofv3 --kernel-trace --stats -d $O/prof/c --output-format csv -- python3 $R/tools/_compress_one.py $1 $2 3 > $O/prof/c.log 2>&1
f=$(find $O/prof/c -name '*kernel_stats.csv' | head -1)
tail -1 $O/prof/c.log; python3 $R/tools/_kstats.py $f ${3:-30}
cp $f $O/r03_kernel_stats_compress_chi$1_$2.csv
rm -rf $O/prof
