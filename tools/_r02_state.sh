#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_h; mkdir -p /tmp/prof_h
rocprofv3 --hip-trace --kernel-trace --stats -d /tmp/prof_h --output-format csv -- python3 $R/tools/_prof_compress.py 256 > /tmp/prof_h/log 2>&1
grep compress /tmp/prof_h/log
f=$(find /tmp/prof_h -name '*hip_api_stats.csv' | head -1)
echo "$f"; head -14 "$f" | cut -c1-150
