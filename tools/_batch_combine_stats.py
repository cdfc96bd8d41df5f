"""Per kernel class of a traced lock-step batch: launches, operands per table launch (grid.y of qil_kn), time.
python3 tools/_batch_combine_stats.py <dir with *kernel_trace.csv> [rows]"""
import csv, glob, os, re, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 30
agg = collections.defaultdict(lambda: [0, 0, 0.0, 0])
for r in csv.DictReader(open(f)):
    name = re.sub(r"\(anonymous namespace\)::|qil_dev::", "", r["Kernel_Name"])
    if "qil_kn<" not in name:
        continue
    name = re.sub(r"^void qil_kn<", "", name)
    name = re.sub(r">, .*", ">", name)[:64]
    n = max(1, int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"])))
    a = agg[name]
    a[0] += 1
    a[1] += n
    a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a[3] += n == 1
tot_l = sum(a[0] for a in agg.values()); tot_r = sum(a[1] for a in agg.values()); tot_t = sum(a[2] for a in agg.values())
print(f"table launches {tot_l}, requests {tot_r} ({tot_r / max(tot_l, 1):.2f} per launch), kernel time {tot_t / 1e3:.1f} ms")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][2])[:nrows]:
    print(f"{a[2] / 1e3:9.1f} ms {a[0]:7d} launches {a[1] / a[0]:5.2f} operands/launch {100.0 * a[3] / a[0]:5.1f}% single {a[2] / a[0]:8.1f} us  {name}")
