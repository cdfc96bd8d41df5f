#!/bin/bash
# Everything the round's committed evidence consists of, for the CURRENT library binary, in the order that keeps the bench
# line's `roofline.traffic` and `truncate.roofline` keyed to that binary (run on the GPU box from the repo root; outputs land
# in gpurun_out/, copy gpurun_out/r03_* to profiles/ afterwards):
#   1. PMC passes: WRITE_SIZE / FETCH_SIZE of the apply kernel (tools/collect_pmc.py) and the f64 MFMA counters of the truncate
#      half (tools/collect_pmc_truncate.py); both json files are copied into profiles/ ON THE BOX so the bench runs pick them up
#   2. rocprofv3 --kernel-trace --stats summaries (tools/profile_r03.sh)
#   3. bench lines (default workload, sigma sweep, 2 ranks over gloo for both), compress! timings, batches, batch occupancy
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 600 python3 tools/collect_pmc.py > $O/collect_pmc.log 2>&1 || echo "collect_pmc failed"
timeout 900 python3 tools/collect_pmc_truncate.py > $O/collect_pmc_truncate.log 2>&1 || echo "collect_pmc_truncate failed"
cp $O/r03_pmc_traffic.json $O/r03_pmc_write_site_apply.csv $O/r03_pmc_fetch_site_apply.csv $O/r03_pmc_truncate.json $R/profiles/ 2>/dev/null
timeout 900 bash tools/profile_r03.sh > $O/profile_r03.log 2>&1
cd $R
timeout 900 python3 bench.py > $O/r03_bench_default.json 2> $O/bench_default.err
timeout 900 python3 bench.py --workload dt_sweep_n24_s64 > $O/r03_bench_sweep.json 2> $O/bench_sweep.err
QIL_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 50 --no-cpu-baseline --no-truncate > $O/r03_bench_gpus2_gloo_apply.json 2> $O/gloo_apply.err
QIL_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 10 --workload dt_sweep_n24_s64 --no-cpu-baseline > $O/r03_bench_gpus2_gloo_sweep.json 2> $O/gloo_sweep.err
timeout 300 python3 tools/_compress_time.py 2>/dev/null > $O/r03_compress_times.txt
for nb in 8 16 32 64; do timeout 300 python3 tools/_compress_concurrent.py $nb 256 2>/dev/null | tail -1 >> $O/r03_compress_times.txt; done
timeout 300 python3 tools/_compress_concurrent.py 32 256 same 2>/dev/null | tail -1 | sed 's/^/32 IDENTICAL chains (no divergence between the chains of a group): /' >> $O/r03_compress_times.txt
QIL_BATCH_LOCKSTEP=0 timeout 300 python3 tools/_compress_concurrent.py 8 256 2>/dev/null | tail -1 | sed 's/^/QIL_BATCH_LOCKSTEP=0: /' >> $O/r03_compress_times.txt
QIL_READBACK=0 timeout 300 python3 tools/_compress_concurrent.py 8 256 2>/dev/null | tail -1 | sed 's/^/QIL_READBACK=0: /' >> $O/r03_compress_times.txt
timeout 200 python3 tools/_apply_compress_batch_time.py 2>/dev/null | tail -1 >> $O/r03_compress_times.txt
timeout 200 python3 tools/_exact_compress_time.py 3 2>/dev/null | tail -1 >> $O/r03_compress_times.txt
cat /sys/fs/cgroup/cpu.max 2>/dev/null | sed 's/^/cgroup cpu.max of this box (quota period, us): /' >> $O/r03_compress_times.txt
./tools/micro/gram_round_cost.bin > $O/r03_gram_round_cost.txt 2>&1
timeout 300 bash tools/r03_timeline.sh 256 f64 > /dev/null 2>&1; cp $O/timeline_256_f64.txt $O/r03_timeline_256_f64.txt
timeout 300 bash tools/r03_batch_trace.sh 32 256 > $O/r03_batch_trace_32.txt 2>&1
# (the tracer crashes on this multi-threaded workload about one run in three, whatever the library: retry)
: > $O/r03_batch_occupancy.jsonl
( cd /tmp && export TMPDIR=/tmp
  for nb in 8 32 64; do
  for try in 1 2 3 4; do
    rm -rf $O/p2
    timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/p2 -- python3 $R/tools/_batch_occupancy.py run $nb 256 > $O/p2.log 2>&1
    if grep -q batch_ms $O/p2.log; then
      grep batch_ms $O/p2.log >> $O/r03_batch_occupancy.jsonl
      python3 $R/tools/_batch_occupancy.py analyse $O/p2 >> $O/r03_batch_occupancy.jsonl
      break
    fi
  done
  done
  rm -rf $O/p2 )
tail -c 1200 $O/r03_bench_default.json; echo; tail -c 900 $O/r03_bench_sweep.json; echo; cat $O/r03_compress_times.txt $O/r03_batch_occupancy.jsonl
