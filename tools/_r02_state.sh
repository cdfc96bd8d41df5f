#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
for i in 1 2 3 4 5; do
rm -rf /tmp/prof_tb; mkdir -p /tmp/prof_tb
rocprofv3 --kernel-trace --stats -d /tmp/prof_tb --output-format csv -- python3 $R/tools/_truncate_block.py > /tmp/prof_tb/log 2>&1
rc=$?
f=$(find /tmp/prof_tb -name '*kernel_stats.csv' | head -1)
echo "run $i rc=$rc stats=${f:+yes}"
grep -n "Check failed\|F2026\|terminate called\|Segmentation\|Aborted\|core dumped" /tmp/prof_tb/log | head -3 | cut -c1-250
[ $rc -ne 0 -o -z "$f" ] && cp /tmp/prof_tb/log $O/tb_fail_$i.log
done
grep -B4 -A16 "Check failed\|terminate called\|\*\*\* Aborted\|\*\*\* SIG" /tmp/prof_tb/log | head -60 | cut -c1-220
