#!/bin/bash
# kernel-by-kernel timeline of one compress! chain: bash tools/r03_timeline.sh <chi> <f64|c64> [from to]
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/tl && mkdir -p $O/tl
rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 $R/tools/_compress_one.py $1 $2 2 > $O/tl.log 2>&1
tail -1 $O/tl.log
python3 $R/tools/_chain_timeline.py $O/tl ${3:-0.5} ${4:-0.56} > $O/timeline_$1_$2.txt
rm -rf $O/tl
head -1 $O/timeline_$1_$2.txt
