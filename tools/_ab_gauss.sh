#!/bin/bash
# A/B on one box: complex GEMM by three (default build) against four real multiplications.  The second library is built with
#   (cd qilaplace.jl_amd/csrc && make -j8 OUTDIR=$PWD/../../tools/ab/lib CXXFLAGS="<the Makefile's flags> -DQIL_GEMM_GAUSS=0")
# and selected through QILHIP_LIB; tools/ab/ is not kept in the tree.  r05 result: profiles/r05_ab_gauss.txt
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for rep in 1 2; do
  for v in gauss four; do
    if [ $v = four ]; then export QILHIP_LIB=$R/tools/ab/lib/libqilhip.so; else unset QILHIP_LIB; fi
    echo "== $v (repetition $rep)"
    python3 tools/_compress_time.py 2>/dev/null | grep complex
    python3 tools/_exact_compress_time.py 3 2>/dev/null | tail -1
    python3 tools/_apply_compress_one.py 2>/dev/null | tail -1
    python3 tools/_coeff_cfg3.py 2 2>/dev/null | tail -1
  done
done
