#!/bin/bash
# Round 4, first GPU call: the GPU suite on the tightened tolerances, the RCCL (backend "nccl") code path in a world of ONE rank
# for both workloads, the plain single-process lines beside them, the DT builder's in-kernel profile.
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
cd $R
mkdir -p $O
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/r04_pytest_gpu.log 2>&1; tail -3 $O/r04_pytest_gpu.log
QIL_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --gpus 1 --steps 50 --no-cpu-baseline --no-truncate > $O/r04_bench_rccl_n1_apply.json 2> $O/rccl_apply.err; echo "rccl apply rc $?"
QIL_BENCH_FORCE_DIST=1 timeout 600 python3 bench.py --gpus 1 --steps 10 --workload dt_sweep_n24_s64 --no-cpu-baseline > $O/r04_bench_rccl_n1_sweep.json 2> $O/rccl_sweep.err; echo "rccl sweep rc $?"
timeout 600 python3 bench.py --gpus 1 --steps 50 --no-cpu-baseline --no-truncate > $O/r04_bench_plain_n1_apply.json 2> $O/plain_apply.err
timeout 600 python3 bench.py --gpus 1 --steps 10 --workload dt_sweep_n24_s64 --no-cpu-baseline > $O/r04_bench_plain_n1_sweep.json 2> $O/plain_sweep.err
for f in rccl_n1_apply plain_n1_apply rccl_n1_sweep plain_n1_sweep; do echo $f; tail -n1 $O/r04_bench_$f.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('collective_backend'), d['config'].get('ranks_reported_by_collective_backend'))"; done
tail -5 $O/rccl_apply.err $O/rccl_sweep.err
QIL_DT_PROFILE=1 timeout 300 python3 tools/_dt_persist_time.py > $O/r04_dt_persist_profile_before.txt 2>&1; tail -30 $O/r04_dt_persist_profile_before.txt
python3 -c "import qilaplace_jl_amd as q; print('cpu budget', q.host_cpu_budget())"; cat /sys/fs/cgroup/cpu.max
