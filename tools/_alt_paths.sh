#!/bin/bash
# The GPU parity suite under every environment switch that selects an alternative code path that SHIPS (tools/README.md lists
# them; r05 removed the switches whose alternatives had measured slower: QIL_SVD_GRAM, QIL_QR_CHOL, QIL_READBACK,
# QIL_SVD_LOWRANK, QIL_BATCH_LOCKSTEP, QIL_BATCH_WORKERS).
run() { echo "== $*"; env "$@" timeout 1200 python3 -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -2; }
run QIL_SVD_CERT=0                       # no truncation certificate: every gauge step is an SVD
run QIL_ENCODE_PAR_DEPTH=0               # sequential RSVD bisection
run QIL_ENCODE_PAR_DEPTH=5
run QIL_DT_BUILDER=launches              # the launch-per-step DT builder (the fallback for truncated bonds > 26) for every build
run QIL_DT_DCAP=24                       # smaller in-LDS plan: more builds overflow into the fallback
run QIL_CPU_BUDGET=2                     # one launcher thread, one lock-step group (a rank that gets 2 CPUs of an 8-rank node's quota)
run QIL_SVD_NOSORT=1                     # r06: no norm-sorted columns before the QR of graded operands (the r05 one-factor SVD)
