#!/bin/bash
# Everything the round's committed evidence consists of, for the CURRENT library binary, in the order that keeps the bench
# line's `roofline.traffic`, `truncate.roofline` and `truncate.roofline_batch64` keyed to that binary (run on the GPU box from the
# repo root; outputs land in gpurun_out/, copy gpurun_out/r04_* to profiles/ afterwards):
#   1. PMC passes: WRITE_SIZE / FETCH_SIZE of the apply kernel (tools/collect_pmc.py) and the f64 MFMA counters of the truncate
#      half incl. the 64-pair batch (tools/collect_pmc_truncate.py); the json files are copied into profiles/ ON THE BOX so the
#      bench runs pick them up
#   2. rocprofv3 --kernel-trace --stats summaries (tools/profile_r04.sh)
#   3. bench lines: default workload, sigma sweep, 2 ranks over gloo for both, the RCCL ("nccl") path in a world of one for both
#   4. compress! timings, batches (compress_batch 8..64, apply_compress_batch of 64 DT / zT pairs), batch occupancy, builders
#   5. the GPU suite's log
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 600 python3 tools/collect_pmc.py > $O/collect_pmc.log 2>&1 || echo "collect_pmc failed"
timeout 1500 python3 tools/collect_pmc_truncate.py > $O/collect_pmc_truncate.log 2>&1 || echo "collect_pmc_truncate failed"
cp $O/r04_pmc_traffic.json $O/r04_pmc_write_site_apply.csv $O/r04_pmc_fetch_site_apply.csv $O/r04_pmc_truncate.json $R/profiles/ 2>/dev/null
timeout 1500 bash tools/profile_r04.sh > $O/profile_r04.log 2>&1
cd $R
timeout 900 python3 bench.py > $O/r04_bench_default.json 2> $O/bench_default.err
timeout 900 python3 bench.py --workload dt_sweep_n24_s64 > $O/r04_bench_sweep.json 2> $O/bench_sweep.err
QIL_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 50 --no-cpu-baseline --no-truncate > $O/r04_bench_gpus2_gloo_apply.json 2> $O/gloo_apply.err
QIL_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 10 --workload dt_sweep_n24_s64 --no-cpu-baseline > $O/r04_bench_gpus2_gloo_sweep.json 2> $O/gloo_sweep.err
QIL_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --gpus 1 --steps 50 --no-cpu-baseline --no-truncate > $O/r04_bench_rccl_n1_apply.json 2> $O/rccl_apply.err
QIL_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --gpus 1 --steps 10 --workload dt_sweep_n24_s64 --no-cpu-baseline > $O/r04_bench_rccl_n1_sweep.json 2> $O/rccl_sweep.err
T=$O/r04_compress_times.txt
timeout 300 python3 tools/_compress_time.py 2>/dev/null > $T
for nb in 8 16 32 64; do timeout 300 python3 tools/_compress_concurrent.py $nb 256 2>/dev/null | tail -1 >> $T; done
timeout 300 python3 tools/_compress_concurrent.py 32 256 same 2>/dev/null | tail -1 | sed 's/^/32 IDENTICAL chains (no divergence between the chains of a group): /' >> $T
QIL_BATCH_LOCKSTEP=0 timeout 300 python3 tools/_compress_concurrent.py 8 256 2>/dev/null | tail -1 | sed 's/^/QIL_BATCH_LOCKSTEP=0: /' >> $T
QIL_CPU_BUDGET=2 timeout 300 python3 tools/_compress_concurrent.py 64 256 2>/dev/null | tail -1 | sed 's/^/QIL_CPU_BUDGET=2 (one launcher, one group): /' >> $T
timeout 200 python3 tools/_apply_compress_batch_time.py 2>/dev/null | tail -1 >> $T
for k in zt dt; do for nb in 8 64; do timeout 600 python3 tools/_apply_compress_batch64.py $nb $k 2>/dev/null | tail -1 >> $T; done; done
timeout 200 python3 tools/_exact_compress_time.py 3 2>/dev/null | tail -1 >> $T
timeout 300 python3 tools/_exact_vs_oracle.py 2>/dev/null | tail -3 >> $T
cat /sys/fs/cgroup/cpu.max 2>/dev/null | sed 's/^/cgroup cpu.max of this box (quota period, us): /' >> $T
python3 -c "import qilaplace_jl_amd as q; print('qil_host_cpu_budget:', q.host_cpu_budget())" >> $T 2>/dev/null
timeout 600 bash tools/r04_batch_occupancy.sh > /dev/null 2>&1
QIL_DT_PROFILE=1 timeout 300 python3 tools/_dt_persist_value_scan.py 2>&1 | grep -v amdgpu.ids > $O/r04_dt_persist_profile.txt
timeout 600 python3 tools/_chain_persist_check.py 2>&1 | grep -v amdgpu.ids > $O/r04_chain_builder.txt
timeout 300 python3 tools/_zt_build_breakdown.py 24 2>&1 | grep -v amdgpu.ids | tail -1 >> $O/r04_chain_builder.txt
timeout 1500 python3 -m pytest tests -m gpu -q > $O/r04_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/r04_pytest_gpu.log
tail -c 1500 $O/r04_bench_default.json; echo; tail -c 900 $O/r04_bench_sweep.json; echo; cat $T $O/r04_batch_occupancy.jsonl; tail -4 $O/r04_chain_builder.txt
