#!/bin/bash
for cfg in "8192 2048" "2048 512" "1024 256" "4096 1024"; do
  set -- $cfg
  echo "TSQR_MIN_ROWS=$1 MIN_CHUNK=$2"
  QIL_TSQR_MIN_ROWS=$1 QIL_TSQR_MIN_CHUNK=$2 timeout 300 python tools/_prof_encode30.py 2>&1 | grep encode | tail -1
  QIL_TSQR_MIN_ROWS=$1 QIL_TSQR_MIN_CHUNK=$2 timeout 300 python tools/bench_aux.py 2>&1 | grep -E "rsvd_random\", \"n\": 24|ztmps_rsvd" | cut -c1-120
done
