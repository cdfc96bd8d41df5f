#!/bin/bash
# Everything a round's committed evidence consists of, for the CURRENT library binary, in the order that keeps the bench line's
# `roofline.traffic`, `truncate.roofline*` and the counted-MFMA figures of the configs block keyed to that binary.  Run on the
# GPU box from the repo root:   QIL_ROUND=r06 bash tools/collect_round.sh      (outputs: gpurun_out/<round>_*, copy to profiles/)
#   1. PMC passes (separate rocprofv3 --pmc processes): WRITE_SIZE / FETCH_SIZE of the apply kernel (tools/collect_pmc.py), f64 MFMA
#      counters of the truncate half, the 64-pair batch, the 64-query read-out and the n = 30 encode (tools/collect_pmc_truncate.py);
#      the json files are copied into profiles/ ON THE BOX so that the bench runs below pick them up
#   2. rocprofv3 --kernel-trace --stats summaries of six workloads
#   3. bench lines: default (with the configs block), damping sweep, 2 ranks over gloo for both, the RCCL path in a world of one
#      through torch.distributed ("nccl") and through the library's own communicator ("cabi")
#   4. compress! timings and batches, DT builder profile, chain builders
#   5. the GPU suite's log
R=${GRAFT_REPO_ROOT:-/root/repo}
export QIL_ROUND=${QIL_ROUND:-r06}
P=$QIL_ROUND
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 1800 python3 tools/collect_pmc.py > $O/collect_pmc.log 2>&1 || echo "collect_pmc failed"
timeout 2400 python3 tools/collect_pmc_truncate.py > $O/collect_pmc_truncate.log 2>&1 || echo "collect_pmc_truncate failed"
cp $O/${P}_pmc_traffic.json $O/${P}_pmc_write_site_apply.csv $O/${P}_pmc_fetch_site_apply.csv $O/${P}_pmc_truncate.json $R/profiles/ 2>/dev/null
(
  cd /tmp && export TMPDIR=/tmp
  mkdir -p $O/prof
  run() {  # name, then the python command line
    name=$1; shift
    for try in 1 2 3; do
      rm -rf $O/prof/$name
      rocprofv3 --kernel-trace --stats -d $O/prof/$name --output-format csv -- "$@" > $O/prof/$name.log 2>&1
      f=$(find $O/prof/$name -name '*kernel_stats.csv' | head -1)
      if [ -n "$f" ]; then cp "$f" $O/${P}_kernel_stats_$name.csv; break; fi
    done
  }
  run zt_n24_chi64_D128 python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-truncate --no-configs
  run dt_sweep_n24_s64 python3 $R/bench.py --workload dt_sweep_n24_s64 --steps 3 --warmup 1 --no-cpu-baseline
  run compress_chi256 python3 $R/tools/_compress_one.py 256 f64 3
  run exact_compress python3 $R/tools/_exact_compress_time.py 3
  run coefficient_batch_cfg3 python3 $R/tools/_coeff_cfg3.py 3
  run encode_n30 python3 $R/tools/_prof_encode30.py 2
  run zt_compress_n24 python3 $R/tools/_zt_compress_one.py 24 3
  rm -rf $O/prof
)
cd $R
timeout 900 python3 bench.py > $O/${P}_bench_default.json 2> $O/bench_default.err
timeout 900 python3 bench.py --workload dt_sweep_n24_s64 > $O/${P}_bench_sweep.json 2> $O/bench_sweep.err
timeout 900 python3 bench.py --workload dt_sweep_n24_weak --no-cpu-baseline > $O/${P}_bench_sweep_weak.json 2> $O/bench_sweep_weak.err
QIL_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 50 --no-cpu-baseline --no-truncate --no-configs > $O/${P}_bench_gpus2_gloo_apply.json 2> $O/gloo_apply.err
QIL_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 8 --steps 5 --workload dt_sweep_n24_s64 --no-cpu-baseline > $O/${P}_bench_gpus8_gloo_sweep_strong.json 2> $O/gloo8_sweep.err
QIL_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 8 --steps 5 --workload dt_sweep_n24_weak --no-cpu-baseline > $O/${P}_bench_gpus8_gloo_sweep_weak.json 2> $O/gloo8_weak.err
QIL_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 8 --steps 20 --workload qft_n20_chi32_D64 --no-cpu-baseline --no-truncate --no-configs > $O/${P}_bench_gpus8_gloo_apply_cfg2.json 2> $O/gloo8_apply.err
QIL_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --gpus 1 --steps 50 --no-cpu-baseline --no-truncate --no-configs > $O/${P}_bench_rccl_n1_apply.json 2> $O/rccl_apply.err
QIL_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --gpus 1 --steps 10 --workload dt_sweep_n24_s64 --no-cpu-baseline > $O/${P}_bench_rccl_n1_sweep.json 2> $O/rccl_sweep.err
QIL_BENCH_BACKEND=cabi QIL_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --gpus 1 --steps 50 --no-cpu-baseline --no-truncate --no-configs > $O/${P}_bench_cabi_n1_apply.json 2> $O/cabi_apply.err
QIL_BENCH_BACKEND=cabi QIL_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --gpus 1 --steps 10 --workload dt_sweep_n24_s64 --no-cpu-baseline > $O/${P}_bench_cabi_n1_sweep.json 2> $O/cabi_sweep.err
T=$O/${P}_compress_times.txt
timeout 300 python3 tools/_compress_time.py 2>/dev/null > $T
for nb in 8 16 32 64; do timeout 300 python3 tools/_compress_concurrent.py $nb 256 2>/dev/null | tail -1 >> $T; done
QIL_CPU_BUDGET=2 timeout 300 python3 tools/_compress_concurrent.py 64 256 2>/dev/null | tail -1 | sed 's/^/QIL_CPU_BUDGET=2 (one launcher, one group): /' >> $T
for k in zt dt; do for nb in 8 64; do timeout 600 python3 tools/_apply_compress_batch64.py $nb $k 2>/dev/null | tail -1 >> $T; done; done
timeout 200 python3 tools/_exact_compress_time.py 3 2>/dev/null | tail -1 >> $T
timeout 200 python3 tools/_apply_compress_one.py 2>/dev/null | tail -1 >> $T
cat /sys/fs/cgroup/cpu.max 2>/dev/null | sed 's/^/cgroup cpu.max of this box (quota period, us): /' >> $T
python3 -c "import qilaplace_jl_amd as q; print('qil_host_cpu_budget:', q.host_cpu_budget())" >> $T 2>/dev/null
QIL_DT_PROFILE=1 timeout 300 python3 tools/_dt_persist_value_scan.py 2>&1 | grep -v amdgpu.ids > $O/${P}_dt_persist_profile.txt
timeout 600 python3 tools/_chain_persist_check.py 2>&1 | grep -v amdgpu.ids > $O/${P}_chain_builder.txt
timeout 300 python3 tools/_zt_build_time.py 24 30 2>&1 | grep case > $O/${P}_zt_build_time.txt
timeout 600 bash tools/_svd_sort_ab.sh > $O/${P}_svd_sort_ab.txt 2>&1
timeout 1500 python3 -m pytest tests -m gpu -q > $O/${P}_pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/${P}_pytest_gpu.log
tail -c 1500 $O/${P}_bench_default.json; echo; tail -c 900 $O/${P}_bench_sweep.json; echo; cat $T; tail -4 $O/${P}_chain_builder.txt
