#!/bin/bash
# batch occupancy (mean CU share from dispatch timestamps) and operands per table launch of compress_batch, 8 / 32 / 64 chains
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
: > $O/r04_batch_occupancy.jsonl
for nb in 8 32 64; do
  for try in 1 2 3 4; do
    rm -rf $O/p2
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/_batch_occupancy.py run $nb 256 > $O/p2.log 2>&1
    if grep -q batch_ms $O/p2.log; then
      grep batch_ms $O/p2.log >> $O/r04_batch_occupancy.jsonl
      python3 $R/tools/_batch_occupancy.py analyse $O/p2 >> $O/r04_batch_occupancy.jsonl
      [ $nb = 64 ] && python3 $R/tools/_batch_combine_stats.py $O/p2 14 > $O/r04_batch_combine_64.txt
      break
    fi
  done
done
rm -rf $O/p2
cat $O/r04_batch_occupancy.jsonl; cat $O/r04_batch_combine_64.txt
