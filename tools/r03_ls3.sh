#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
QIL_BATCH_DEBUG=1 timeout 200 python3 tools/_compress_concurrent.py 8 256 2>&1 | grep -v "slot" | tail -2
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/b --output-format csv -- python3 $R/tools/_batch_occupancy.py run 8 256 > $R/gpurun_out/prof/b.log 2>&1
f=$(find $R/gpurun_out/prof/b -name '*kernel_stats.csv' | head -1)
tail -2 $R/gpurun_out/prof/b.log; python3 $R/tools/_kstats.py $f 16
rm -rf $R/gpurun_out/prof
