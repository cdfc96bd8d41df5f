#!/bin/bash
# First GPU pass of round 3: the parity suite, both bench workloads with the new CPU baselines, and the 2-rank gloo runs
# (bench.py --gpus 2 on a 1-GPU box: the parent spawns the ranks before any GPU call; both ranks use device 0).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -5 $O/r03_pytest_gpu.log
timeout 900 python3 bench.py > $O/r03_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 900 python3 bench.py --workload dt_sweep_n24_s64 > $O/r03_bench_sweep.json 2> $O/bench_sweep.err; echo "sweep rc=$?"
QIL_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 50 --no-cpu-baseline --no-truncate > $O/r03_bench_gpus2_gloo_apply.json 2> $O/gloo_apply.err; echo "gloo apply rc=$?"
QIL_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 2 --steps 10 --workload dt_sweep_n24_s64 --no-cpu-baseline > $O/r03_bench_gpus2_gloo_sweep.json 2> $O/gloo_sweep.err; echo "gloo sweep rc=$?"
for f in r03_bench_default r03_bench_sweep r03_bench_gpus2_gloo_apply r03_bench_gpus2_gloo_sweep; do echo "== $f"; tail -c 2500 $O/$f.json; echo; done
tail -3 $O/bench_default.err $O/bench_sweep.err $O/gloo_apply.err $O/gloo_sweep.err
