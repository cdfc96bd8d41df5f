#!/bin/bash
# kernel classes + idle time of ONE exact compress! of the bond-1008 product (the window after the last apply kernel)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
rm -rf $O/tl && mkdir -p $O/tl
rocprofv3 --kernel-trace --output-format csv -d $O/tl -- python3 $R/tools/_exact_compress_time.py 2 > $O/tl.log 2>&1
tail -1 $O/tl.log
QIL_TIMELINE_MARKER=site_apply python3 $R/tools/_chain_timeline.py $O/tl 0.0 0.0 | head -3
rm -rf $O/tl
