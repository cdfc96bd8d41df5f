#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -q > $O/r03_pytest_gpu.log 2>&1; echo "pytest rc=$?"
tail -8 $O/r03_pytest_gpu.log
timeout 900 python3 bench.py > $O/r03_bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
timeout 900 python3 bench.py --workload dt_sweep_n24_s64 > $O/r03_bench_sweep.json 2> $O/bench_sweep.err; echo "sweep rc=$?"
for f in r03_bench_default r03_bench_sweep; do echo "== $f"; tail -c 3000 $O/$f.json; echo; done
tail -3 $O/bench_default.err $O/bench_sweep.err
