#!/bin/bash
# VERDICT r04 item 3 (stop rule): the mid-size SVD's sweeps as per-round launches (default) against ONE persistent launch with a
# device-scope barrier between rounds (QIL_SVD_PERSIST=1): compress! chi 128 / 256 / 512 (f64, c64), the exact compress!(apply) of
# the bond-1008 product, the fused route; then the parity tests of the truncating ops under the persistent form.
for P in 0 1; do
  echo "== QIL_SVD_PERSIST=$P"
  QIL_SVD_PERSIST=$P timeout 600 python3 tools/_compress_time.py
  QIL_SVD_PERSIST=$P timeout 600 python3 tools/_exact_compress_time.py 3
  QIL_SVD_PERSIST=$P timeout 600 python3 tools/_apply_compress_one.py 2>&1 | tail -2
done
echo "== parity under QIL_SVD_PERSIST=1"
QIL_SVD_PERSIST=1 timeout 1200 python3 -m pytest tests -m gpu -q -x -k "compress or canonicalize or svd or signal or rsvd or fuzz" 2>&1 | tail -3
