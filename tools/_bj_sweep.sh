#!/bin/bash
for bjmin in 128 256 100000; do
  echo "== bj_min $bjmin"
  QIL_BJ_MIN=$bjmin QIL_SVD_DEBUG=1 timeout 600 python3 tools/_svd_check.py 300 500 260 700 1000 600 512 3000 320 2>&1 | grep -E "'rec'" | sed -e "s/'lapack_s.*//" -e "s/'rec'.*'s'/'s'/"
done
echo "== inner 1"
QIL_BJ_INNER=1 QIL_SVD_DEBUG=1 timeout 600 python3 tools/_svd_check.py 1000 2048 4096 2>&1 | grep -E "block sweeps|'rec'" | sed -e "s/'lapack_s.*//" -e "s/'cplx.*'rec'/'rec'/"
