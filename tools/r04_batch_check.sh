#!/bin/bash
# lock-step launcher after a change: batches of different / identical chains against one chain, bit-identity tests
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for k in dt zt; do for nb in 8 64; do timeout 600 python3 tools/_apply_compress_batch64.py $nb $k 2>&1 | tail -1; done; done
for nb in 8 16 32 64; do timeout 300 python3 tools/_compress_concurrent.py $nb 256 2>/dev/null | tail -1; done
timeout 300 python3 tools/_compress_concurrent.py 32 256 same 2>/dev/null | tail -1
QIL_BATCH_DEBUG=1 timeout 600 python3 tools/_apply_compress_batch64.py 64 dt 2>&1 | grep -E "lock-step group" | tail -4
timeout 900 python3 -m pytest tests -m gpu -x -q -k "batch or concurrent or sweep or two_contexts" 2>&1 | tail -3
