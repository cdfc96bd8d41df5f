"""Aggregate a rocprofv3 counter_collection.csv per kernel: mean counter value per dispatch.
python tools/_pmc_agg.py <counter_collection.csv> [kernel-substring]"""
import collections
import csv
import sys
rows = csv.DictReader(open(sys.argv[1]))
sub = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(collections.Counter)
for r in rows:
    k = r["Kernel_Name"]
    if sub and sub not in k:
        continue
    k = k[:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
for k, v in agg.items():
    print(k, {c: round(x / cnt[k][c], 1) for c, x in v.items()}, "dispatches", max(cnt[k].values()))
