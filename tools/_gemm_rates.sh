#!/bin/bash
# GEMM rates for the tile configurations (QIL_GEMM_CFG: -1 default, 0 = 64x64 PIPE, 1 = 64x64, 3 = 128x128)
for c in -1 0 3; do
  echo "QIL_GEMM_CFG=$c"
  QIL_GEMM_CFG=$c timeout 300 python tools/bench_aux.py 2>&1 | grep gemm_device | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['m'], d['n'], d['k'], d['dtype'], d['opA'], round(d['ms'],3), 'ms', round(d['tflops'],1), 'TF')
"
done
