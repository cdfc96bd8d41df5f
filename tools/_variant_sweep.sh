#!/bin/bash
# usage: tools/_variant_sweep.sh ENVVAR "v1 v2 ..."  -- runs bench.py once per value, prints kernel time
for v in $2; do
  env $1=$v timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline --queries 4 2>/dev/null | tail -1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$1=$v', 'ms/step %.3f kernel_ms %.3f achieved %.0f GB/s'%(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['achieved']))"
done
