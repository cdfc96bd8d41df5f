"""Per-queue view of a traced lock-step batch (the trace tools/_batch_occupancy.py run produces):
   python3 tools/_batch_trace_stats.py DIR
for the last batch: per hardware queue the launches, busy share and mean gap; per kernel class (single form qil_k1 / table form
qil_kn) the mean duration of the dominant kernels; and how long a table launch of the Gram round takes by grid height
(= chains in the launch)."""
import sys, csv, glob, os, re, collections
d = sys.argv[1]
f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?"), int(r["Grid_Size_Y"]) // max(1, int(r["Workgroup_Size_Y"]))) for r in rows)
fills = [i for i, x in enumerate(ev) if "fill_normal" in x[2]]
sel = ev[fills[-1] + 1:]
t0, t1 = sel[0][0], max(x[1] for x in sel)
print(f"last batch: {len(sel)} launches over {(t1 - t0) / 1e6:.1f} ms")
byq = collections.defaultdict(list)
for x in sel: byq[x[3]].append(x)
for q, xs in sorted(byq.items()):
    busy = sum(e - s for s, e, *_ in xs)
    gaps = [xs[i][0] - xs[i - 1][1] for i in range(1, len(xs))]
    pos = [g for g in gaps if g > 0]
    hist = collections.Counter(min(9, int(g / 20e3)) for g in pos)
    print("    idle gaps by length (20 us bins, last = 180 us and more):", [hist.get(b, 0) for b in range(10)], "total idle ms in gaps >= 60 us:", round(sum(g for g in pos if g >= 60e3) / 1e6, 1))
    # what runs right after the long gaps
    after = collections.Counter()
    for i in range(1, len(xs)):
        if xs[i][0] - xs[i - 1][1] >= 60e3:
            m = re.search(r"qil_k[1n]<\(anonymous namespace\)::(\w+)", xs[i][2]); b = re.search(r"qil_k[1n]<\(anonymous namespace\)::(\w+)", xs[i - 1][2])
            after[((b.group(1) if b else "?"), (m.group(1) if m else "?"))] += 1
    print("    (kernel before, kernel after) of the gaps >= 60 us:", after.most_common(8))
    big = sorted(((xs[i][0] - xs[i - 1][1], i) for i in range(1, len(xs))), reverse=True)[:8]
    for gl, i in sorted(big, key=lambda t: t[1]):
        nm = lambda n: (re.search(r"qil_k[1n]<\(anonymous namespace\)::(\w+)", n) or re.search(r"(\w+)", n)).group(1)
        print(f"    gap of {gl / 1e3:9.1f} us at {(xs[i - 1][1] - t0) / 1e6:8.2f} ms after launch {i} of {len(xs)}: {nm(xs[i - 1][2])} -> {nm(xs[i][2])}")
    print(f"  queue {q}: {len(xs)} launches, busy {busy / (t1 - t0):.2f} of the span, mean kernel {busy / len(xs) / 1e3:.1f} us, "
          f"mean idle gap {sum(pos) / max(1, len(pos)) / 1e3:.1f} us ({len(pos)} gaps), overlapping successors {sum(1 for g in gaps if g < 0)}")
cls = collections.defaultdict(list)
for s, e, n, q, gy in sel:
    m = re.search(r"qil_k([1n])<\(anonymous namespace\)::(\w+)", n)
    key = (m.group(2), m.group(1)) if m else (re.sub(r"\(.*", "", n)[:40], "-")
    cls[key].append((e - s, gy))
print("kernel classes by total time:")
for key, v in sorted(cls.items(), key=lambda kv: -sum(d for d, _ in kv[1]))[:14]:
    print(f"  {key[0]:28s} form {key[1]}: {len(v):6d} x {sum(d for d, _ in v) / len(v) / 1e3:7.1f} us")
g = collections.defaultdict(list)
for (name, form), v in cls.items():
    if name == "gram_block_round_k" and form == "n":
        for dur, gy in v: g[gy].append(dur)
for gy in sorted(g): print(f"  gram round, {gy} chains in the launch: {len(g[gy]):5d} x {sum(g[gy]) / len(g[gy]) / 1e3:6.1f} us")
