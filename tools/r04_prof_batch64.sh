#!/bin/bash
# rocprofv3 --kernel-trace --stats of the 64-pair apply_compress batch (dt | zt operators)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; K=${1:-dt}
rm -rf $O/pb && mkdir -p $O/pb
for try in 1 2 3; do
  rocprofv3 --kernel-trace --stats -d $O/pb --output-format csv -- python3 $R/tools/_apply_compress_batch64.py 64 $K > $O/pb.log 2>&1
  f=$(find $O/pb -name '*kernel_stats.csv' | head -1)
  if [ -n "$f" ]; then cp "$f" $O/r04_kernel_stats_apply_compress_batch64_$K.csv; break; fi
done
tail -1 $O/pb.log
python3 $R/tools/_kstats.py $O/r04_kernel_stats_apply_compress_batch64_$K.csv 2>/dev/null | head -40
rm -rf $O/pb
