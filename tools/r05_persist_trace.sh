#!/bin/bash
# Kernel-trace comparison for VERDICT r04 item 3: compress! chi 256 -> 128 (f64) and the exact compress!(apply) with per-round launches
# and with the persistent all-sweeps kernel (QIL_SVD_PERSIST=1, code of commit "persistent all-sweeps SVD kernels").
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
for P in 0 1; do
  export QIL_SVD_PERSIST=$P
  rm -rf $O/prof; mkdir -p $O/prof
  rocprofv3 --kernel-trace --stats -d $O/prof/c --output-format csv -- python3 $R/tools/_compress_one.py 256 f64 3 > $O/prof/c.log 2>&1
  f=$(find $O/prof/c -name '*kernel_stats.csv' | head -1)
  echo "== compress! chi 256 -> 128 f64, 3 repetitions, QIL_SVD_PERSIST=$P"; tail -1 $O/prof/c.log; python3 $R/tools/_kstats.py $f 8
  rm -rf $O/prof; mkdir -p $O/prof
  rocprofv3 --kernel-trace --stats -d $O/prof/c --output-format csv -- python3 $R/tools/_exact_compress_time.py 3 > $O/prof/c.log 2>&1
  f=$(find $O/prof/c -name '*kernel_stats.csv' | head -1)
  echo "== exact compress!(apply) of the bond-1008 product, 3 repetitions, QIL_SVD_PERSIST=$P"; tail -1 $O/prof/c.log; python3 $R/tools/_kstats.py $f 8
done
rm -rf $O/prof
