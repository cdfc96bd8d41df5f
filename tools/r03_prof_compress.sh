#!/bin/bash
# rocprofv3 kernel stats of one compress! chi 256 -> 128 (gram rounds on / off)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O/prof
for g in 1 0; do
  export QIL_SVD_GRAM=$g
  rocprofv3 --kernel-trace --stats -d $O/prof/c$g --output-format csv -- python3 $R/tools/_compress_one.py 256 f64 3 > $O/prof/c$g.log 2>&1
  f=$(find $O/prof/c$g -name '*kernel_stats.csv' | head -1)
  echo "== gram=$g"; tail -1 $O/prof/c$g.log; head -14 $f | cut -c1-200
done
rm -rf $O/prof
