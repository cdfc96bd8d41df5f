"""How much of the chip a batch of independent truncation chains keeps busy, from a rocprofv3 kernel trace (PMC passes
serialise the kernels, so counters cannot show concurrency; dispatch timestamps can).

  workload (run under `rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/_batch_occupancy.py run [nb] [chi]`):
      warm-up, then ONE compress_batch of nb chains (chi -> chi/2, 24 sites) bracketed by two marker kernels
  analysis (`python3 tools/_batch_occupancy.py analyse DIR`):
      over the batch's time span: time-weighted mean of resident workgroups (sum over running kernels of their grid, one
      workgroup of these kernels = one CU's worth of LDS or 8-16 waves) capped at 256 CUs, the share of the span with at
      least 128 workgroups resident, and the mean number of kernels in flight."""
import os, sys, csv, glob, json
import numpy as np


def run(nb, chi, same=False):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import qilaplace_jl_amd as qil
    ctx = qil.default_context()
    def sat(L, chi, base=2): return [int(min(base ** (i + 1), base ** (L - 1 - i), chi)) for i in range(L - 1)]
    def make(i): return qil.SignalMPS.alloc(sat(24, chi), dtype=np.float64).fill_random(5 if same else 5 + i)
    # one chain alone first (warm pool, code objects).  NOTE: under rocprofv3 this multi-threaded workload dies with a SIGSEGV
    # inside the tracer's interception layer in roughly one run of three -- with this library and with the one from before
    # the batched kernels alike (A/B, 6 runs each), never without the tracer (8 of 8 untraced processes) -- re-run it.
    qil.compress(make(99), maxdim=chi // 2, tol=1e-10)
    ctx.synchronize()
    for rep in range(2):
        items = [make(i) for i in range(nb)]
        qil.compress_batch(items, maxdim=chi // 2, tol=1e-10)
    one = make(100)
    ctx.synchronize()
    qil.norm(one)                                   # marker before (a norm chain: kernels no compress launches)
    ctx.synchronize()
    items = [make(i) for i in range(nb)]
    ctx.synchronize()
    import time
    t0 = time.perf_counter()
    qil.compress_batch(items, maxdim=chi // 2, tol=1e-10)
    ctx.synchronize()
    print(json.dumps({"nb": nb, "chi": chi, "batch_ms": (time.perf_counter() - t0) * 1e3}), flush=True)


def analyse(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    ev = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        wg = 1
        for ax in "XYZ":
            wg *= max(1, int(r[f"Grid_Size_{ax}"]) // max(1, int(r[f"Workgroup_Size_{ax}"])))
        ev.append((s, e, wg, r["Kernel_Name"]))
    ev.sort()
    # the last batch: everything after the last fill_random kernel train of the run (the items of the timed batch)
    fills = [i for i, x in enumerate(ev) if "fill_normal" in x[3]]
    start = ev[fills[-1]][1]
    sel = [x for x in ev if x[0] >= start]
    t0, t1 = min(x[0] for x in sel), max(x[1] for x in sel)
    pts = []
    for s, e, wg, _ in sel:
        pts.append((s, wg, 1))
        pts.append((e, -wg, -1))
    pts.sort()
    cur_wg = cur_k = 0
    last = t0
    area_wg = area_k = span128 = span64 = busy = 0
    for t, dwg, dk in pts:
        dt = t - last
        area_wg += min(cur_wg, 256) * dt
        area_k += cur_k * dt
        if cur_wg >= 128: span128 += dt
        if cur_wg >= 64: span64 += dt
        if cur_k > 0: busy += dt
        cur_wg += dwg
        cur_k += dk
        last = t
    span = t1 - t0
    print(json.dumps({"kernels": len(sel), "span_ms": span / 1e6, "gpu_busy_share": busy / span,
                      "mean_workgroups_resident_capped_256": area_wg / span, "mean_cu_share": area_wg / span / 256,
                      "share_of_span_with_ge_128_workgroups": span128 / span,
                      "share_of_span_with_ge_64_workgroups": span64 / span, "mean_kernels_in_flight": area_k / span}))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 8, int(sys.argv[3]) if len(sys.argv) > 3 else 256,
            len(sys.argv) > 4 and sys.argv[4] == "same")
    else:
        analyse(sys.argv[2])
