#!/bin/bash
# per-queue statistics of a traced lock-step batch: bash tools/r03_batch_trace.sh <chains> <chi>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
for try in 1 2 3 4; do
  rm -rf $O/p2
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/_batch_occupancy.py run $1 $2 $3 > $O/p2.log 2>&1
  if grep -q batch_ms $O/p2.log; then
    grep batch_ms $O/p2.log
    python3 $R/tools/_batch_occupancy.py analyse $O/p2
    python3 $R/tools/_batch_trace_stats.py $O/p2
    break
  fi
done
rm -rf $O/p2
