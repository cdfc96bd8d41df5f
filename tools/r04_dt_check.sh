#!/bin/bash
# persistent DT builder after a change: operators / bond dimensions vs the launch-per-step builder and the oracle, the GPU
# tests that cover it, and the in-kernel profile of two damping values (typical and slowest)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
timeout 600 python3 tools/_dt_persist_check.py 2>&1 | grep -v amdgpu.ids | tail -40
QIL_DT_PROFILE=1 timeout 300 python3 tools/_dt_persist_value_scan.py 2>&1 | grep -v "amdgpu.ids" | grep -v "^\[qil dt persistent\] kernel" | awk 'NR%2==0 || /^\[[0-9]/'
timeout 900 python3 -m pytest tests -m gpu -x -q -k "dt_builder or config4 or damping or zt_mpo_batch or device_qft or tutorial_pins or pole_scans" 2>&1 | tail -3
