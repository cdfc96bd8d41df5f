"""Where a zT MPO build spends its time: batched DT half (device), QFT half (host), MPO x MPO product,
final compression (zt_transformer.jl:41-112).  gpurun -- python tools/_zt_build_breakdown.py [n ...]"""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
import qilaplace_jl_amd as qil   # noqa: E402
from qilaplace_jl_amd import builders  # noqa: E402


def main():
    ns = [int(a) for a in sys.argv[1:]] or [24, 30]
    ctx = qil.default_context()
    for n in ns:
        builders._ZT_Q_CACHE.clear()
        t0 = time.perf_counter()
        Q_t = builders.zt_qft_chain_tensors(n, 1e-14, 1000)
        t_q = time.perf_counter() - t0
        for _ in range(2):                                   # second pass: warm pool
            t0 = time.perf_counter()
            dts = builders.build_dt_mpo_batch(n, [2 * np.pi], 1e-14, 1000, ctx)
            ctx.synchronize()
            t_dt = time.perf_counter() - t0
            Q = qil.PairedSiteMPO(Q_t, sites=dts[0].site_ids, ctx=dts[0].ctx)
            t0 = time.perf_counter()
            W = qil.apply(dts[0], Q)
            ctx.synchronize()
            t_ap = time.perf_counter() - t0
            b0 = max(W.bond_dims)
            t0 = time.perf_counter()
            W = qil.mpo_compress(W, "down", 1e-14, 1000)
            ctx.synchronize()
            t_c = time.perf_counter() - t0
        print(json.dumps({"case": "zt_build_breakdown", "n": n, "seconds_qft_host": t_q, "seconds_dt_device": t_dt,
                          "seconds_product": t_ap, "seconds_compress": t_c, "bond_before": int(b0),
                          "bond_after": int(max(W.bond_dims))}), flush=True)


if __name__ == "__main__":
    main()
