#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
rm -rf $O/prof; mkdir -p $O/prof
rocprofv3 --kernel-trace -d $O/prof/b4 --output-format csv -- python3 $R/tools/_compress_concurrent.py 4 > $O/prof/b4.log 2>&1
f=$(find $O/prof/b4 -name '*kernel_trace.csv' | head -1)
head -2 "$f"
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys())
key = lambda r: (r.get('Thread_Id'), r.get('Queue_Id'), r.get('Stream_Id'))
c = collections.Counter(key(r) for r in rows)
for k, v in sorted(c.items(), key=lambda kv: -kv[1])[:40]: print(k, v)
PY
tail -2 $O/prof/b4.log
rm -rf $O/prof
