#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof; mkdir -p $O/prof
rocprofv3 --kernel-trace --stats -d $O/prof/tb --output-format csv -- python3 $R/tools/_truncate_block.py > $O/prof/tb.log 2>&1
echo rc=$?
f=$(find $O/prof/tb -name '*kernel_stats.csv' | head -1)
echo "stats: $f"
[ -n "$f" ] && cp "$f" $O/r02_kernel_stats_truncate_block.csv
grep -n "Check failed\|F2026\|terminate\|Segmentation\|Aborted" $O/prof/tb.log | head -5 | cut -c1-300
find $O/prof/tb -type f | head
rm -rf $O/prof
