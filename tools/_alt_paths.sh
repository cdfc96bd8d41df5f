#!/bin/bash
# The GPU parity suite under every environment switch that selects an alternative code path (tools/README.md lists them).
run() { echo "== $*"; env "$@" timeout 1200 python3 -m pytest tests -m gpu -q -x 2>&1 | tail -2; }
run QIL_SVD_GRAM=0                       # vector-ALU block rounds instead of the Gram-matrix rounds on the matrix cores
run QIL_SVD_GRAM=2                       # Gram rounds for complex operands too
run QIL_QR_CHOL=0                        # Householder / CGS2 panels only
run QIL_READBACK=0                       # read-backs as copy commands + stream synchronisation instead of the polled tickets
run QIL_SVD_CERT=0                       # no truncation certificate: every gauge step is an SVD
run QIL_SVD_LOWRANK=0                    # no certified low-rank route
run QIL_BATCH_LOCKSTEP=0                 # batches: one stream per chain
run QIL_BATCH_LOCKSTEP=1                 # batches: lock-step groups whatever the batch size
run QIL_BATCH_LOCKSTEP=0 QIL_BATCH_WORKERS=3 QIL_ENCODE_PAR_DEPTH=5
run QIL_BATCH_WORKERS=1 QIL_ENCODE_PAR_DEPTH=0
run QIL_DT_BUILDER=launches
run QIL_DT_DCAP=24
run QIL_CPU_BUDGET=2                     # one launcher thread, one lock-step group (a rank that gets 2 CPUs of an 8-rank node's quota)
