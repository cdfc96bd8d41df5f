#!/bin/bash
# kernel trace of the 64-pair apply_compress batch: operands per table launch by kernel class
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; K=${1:-dt}
for try in 1 2 3; do
  rm -rf $O/pb && mkdir -p $O/pb
  rocprofv3 --kernel-trace -d $O/pb --output-format csv -- python3 $R/tools/_apply_compress_batch64.py 64 $K > $O/pb.log 2>&1
  if grep -q pairs/s $O/pb.log; then break; fi
done
tail -1 $O/pb.log
python3 $R/tools/_batch_combine_stats.py $O/pb 40
rm -rf $O/pb
