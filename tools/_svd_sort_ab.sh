#!/bin/bash
# A/B of the norm-sorted columns before the QR of the one-factor SVD (r06): QIL_SVD_NOSORT=1 = the r05 behaviour
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in nosort sort; do
  unset QIL_SVD_NOSORT; if [ $v = nosort ]; then export QIL_SVD_NOSORT=1; fi
  echo "== $v"
  python3 tools/_zt_compress_one.py 24 3 2>/dev/null | grep "zt product" | tail -2
  python3 tools/_zt_build_time.py 24 2>/dev/null | grep -o '"case": "zt_build_verb[^,]*", "n": 24, "seconds_best": [0-9.]*'
  python3 tools/_compress_time.py 2>/dev/null | tail -6
  python3 tools/_exact_compress_time.py 3 2>/dev/null | tail -1
  python3 tools/_apply_compress_one.py 2>/dev/null | tail -1
  python3 tools/_prof_encode30.py 2 2>/dev/null | tail -1
done
