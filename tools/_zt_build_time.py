"""build_zt_mpo on the device behind one C verb (qil_build_zt_mpo_batch): wall time of a single value and of a 64-value sweep,
beside the same steps called one after another from their own entries (where the verb's time goes).
gpurun -- python tools/_zt_build_time.py [n ...]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import qilaplace_jl_amd as qil   # noqa: E402


def timed(fn, reps=3):
    out = []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        r = fn()
        qil.default_context().synchronize()
        out.append(time.perf_counter() - t0)
        del r
    return min(out[1:]), sum(out[1:]) / reps, out[0]


def main():
    ns = [int(a) for a in sys.argv[1:]] or [24, 30]
    ctx = qil.default_context()
    for n in ns:
        wr = 2 * np.pi
        best, mean, first = timed(lambda: qil.build_zt_mpo(n, wr))
        rec = {"case": "zt_build_verb", "n": n, "seconds_best": best, "seconds_mean": mean, "seconds_first_call": first,
               "max_bond": max(qil.build_zt_mpo(n, wr).bond_dims)}
        for route in ("parts", "host"):
            b, m, f = timed(lambda: qil.build_zt_mpo_batch(n, [wr], qft=route)[0])
            rec[f"seconds_{route}_best"] = b
        # the steps one after another
        t0 = time.perf_counter(); dts = qil.build_dt_mpo_batch(n, [wr]); ctx.synchronize(); rec["seconds_dt"] = time.perf_counter() - t0
        t0 = time.perf_counter(); Q = qil.zt_qft_chain_device(n, dts[0].site_ids); ctx.synchronize(); rec["seconds_qft_chain"] = time.perf_counter() - t0
        t0 = time.perf_counter(); P = qil.apply(dts[0], Q); ctx.synchronize(); rec["seconds_product"] = time.perf_counter() - t0
        rec["bond_before"] = max(P.bond_dims)
        t0 = time.perf_counter(); qil.mpo_compress(P, "down", 1e-14, 1000); ctx.synchronize(); rec["seconds_compress"] = time.perf_counter() - t0
        print(json.dumps(rec), flush=True)
        if n <= 24:
            wrs = np.linspace(0.25, 16.0, 64)
            b, m, f = timed(lambda: qil.build_zt_mpo_batch(n, wrs), reps=2)
            print(json.dumps({"case": "zt_build_verb_batch64", "n": n, "seconds_best": b, "seconds_mean": m, "seconds_first_call": f}), flush=True)
            b, m, f = timed(lambda: qil.build_zt_mpo_batch(n, wrs, qft="host"), reps=1)
            print(json.dumps({"case": "zt_build_host_qft_batch64", "n": n, "seconds_best": b}), flush=True)


if __name__ == "__main__":
    main()
