"""Where one compress! chain spends its time, launch by launch: a rocprofv3 kernel trace of tools/_compress_one.py, printed as
the sequence (start offset, duration, gap to the previous kernel's end, name) of a window of the LAST repetition, plus the
totals per phase of the window (kernel time against idle gaps = host round trips).

  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 $R/tools/_compress_one.py 256 f64 2
  python3 tools/_chain_timeline.py DIR [first_fraction last_fraction]      (window of the last repetition, default 0.5 0.56)"""
import sys, csv, glob, os, re

d = sys.argv[1]
f0, f1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.5, 0.56)
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
marker = os.environ.get("QIL_TIMELINE_MARKER", "fill_normal")     # the chain starts after the last kernel whose name contains this
fills = [i for i, x in enumerate(ev) if marker in x[2]]
sel = ev[fills[-1] + 1:]
t0, t1 = sel[0][0], sel[-1][1]
span = t1 - t0
busy = sum(e - s for s, e, _ in sel)
gaps = [sel[i][0] - sel[i - 1][1] for i in range(1, len(sel))]
print(f"last repetition: {len(sel)} launches, span {span/1e6:.2f} ms, kernel time {busy/1e6:.2f} ms, idle {sum(g for g in gaps if g > 0)/1e6:.2f} ms "
      f"(gaps > 10 us: {sum(1 for g in gaps if g > 10000)} totalling {sum(g for g in gaps if g > 10000)/1e6:.2f} ms)")
import collections
cls = collections.Counter(); num = collections.Counter()
for s_, e_, n_ in sel:
    m = re.search(r"qil_k[1n]<\(anonymous namespace\)::(\w+)", n_)
    k = m.group(1) if m else re.sub(r"\(.*", "", n_)[:40]
    cls[k] += e_ - s_; num[k] += 1
print("kernel classes of the repetition:", ", ".join(f"{k} {v/1e6:.1f} ms ({num[k]})" for k, v in cls.most_common(14)))
lo, hi = t0 + f0 * span, t0 + f1 * span
prev = None
for s, e, n in sel:
    if s >= lo and s <= hi:
        n = re.sub(r"\(anonymous namespace\)::|qil_dev::|void qil_k1<", "", n)
        n = re.sub(r"\(.*", "", n)[:60]
        gap = (s - prev) / 1e3 if prev else 0.0
        print(f"{(s - t0)/1e3:10.1f} us  {(e - s)/1e3:7.1f} us  gap {gap:6.1f}  {n}")
    prev = e
