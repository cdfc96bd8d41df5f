#!/bin/bash
# Everything the round's committed evidence consists of, for the CURRENT library binary, in the order that keeps the bench
# line's `roofline.traffic` keyed to that binary (run on the GPU box from the repo root; outputs land in gpurun_out/, copy
# gpurun_out/r02_* to profiles/ afterwards):
#   1. PMC passes (WRITE_SIZE, FETCH_SIZE; separate processes) -> r02_pmc_traffic.json, copied into profiles/ on the box so
#      that the bench runs below pick it up
#   2. rocprofv3 --kernel-trace --stats summaries (tools/profile_r02.sh)
#   3. bench lines (default workload, sigma sweep), compress! timings, batch occupancy from a kernel trace
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 600 python3 tools/collect_pmc.py > $O/collect_pmc.log 2>&1 || echo "collect_pmc failed"
cp $O/r02_pmc_traffic.json $O/r02_pmc_write_site_apply.csv $O/r02_pmc_fetch_site_apply.csv $R/profiles/ 2>/dev/null
timeout 900 bash tools/profile_r02.sh > $O/profile_r02.log 2>&1
cd $R
timeout 600 python3 bench.py > $O/r02_bench_default.json 2> $O/bench_default.err
timeout 600 python3 bench.py --workload dt_sweep_n24_s64 > $O/r02_bench_sweep.json 2> $O/bench_sweep.err
timeout 300 python3 tools/_compress_time.py 2>/dev/null > $O/r02_compress_times.txt
timeout 200 python3 tools/_compress_concurrent.py 8 256 2>/dev/null | tail -1 >> $O/r02_compress_times.txt
# (the tracer crashes on this multi-threaded workload about one run in three, whatever the library: retry)
( cd /tmp && export TMPDIR=/tmp
  for try in 1 2 3 4; do
    rm -rf $O/p2
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/_batch_occupancy.py run 8 256 > $O/p2.log 2>&1
    if grep -q batch_ms $O/p2.log; then
      grep batch_ms $O/p2.log > $O/r02_batch_occupancy.jsonl
      python3 $R/tools/_batch_occupancy.py analyse $O/p2 >> $O/r02_batch_occupancy.jsonl
      break
    fi
  done
  rm -rf $O/p2 )
tail -c 1500 $O/r02_bench_default.json; echo; tail -c 600 $O/r02_bench_sweep.json; echo; cat $O/r02_compress_times.txt $O/r02_batch_occupancy.jsonl
