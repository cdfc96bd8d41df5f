#!/bin/bash
# Round-4 rocprofv3 kernel-trace summaries (run on the GPU box from the repo root; outputs under gpurun_out/):
#   bash tools/profile_r04.sh
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O/prof
run() {  # name, then the python command line
  name=$1; shift
  for try in 1 2 3; do
    rm -rf $O/prof/$name
    rocprofv3 --kernel-trace --stats -d $O/prof/$name --output-format csv -- "$@" > $O/prof/$name.log 2>&1
    f=$(find $O/prof/$name -name '*kernel_stats.csv' | head -1)
    if [ -n "$f" ]; then cp "$f" $O/r04_kernel_stats_$name.csv; break; fi
  done
}
run zt_n24_chi64_D128 python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-truncate
run dt_sweep_n24_s64 python3 $R/bench.py --workload dt_sweep_n24_s64 --steps 3 --warmup 1 --no-cpu-baseline
run compress_chi256 python3 $R/tools/_compress_one.py 256 f64 3
run exact_compress python3 $R/tools/_exact_compress_time.py 3
run apply_compress_batch64_zt python3 $R/tools/_apply_compress_batch64.py 64 zt 3
run chain_builders python3 $R/tools/_chain_persist_check.py
rm -rf $O/prof
ls -la $O/r04_kernel_stats_*.csv
