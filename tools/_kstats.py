"""Condensed view of a rocprofv3 kernel_stats.csv: python tools/_kstats.py <csv> [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::|qil_dev::", "", r["Name"])
    name = re.sub(r"\(.*", "", name)[:70]
    print(f"{float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['Percentage']):5.1f}% {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:8.1f} us  {name}")
