#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout 600 python3 -m pytest tests -m gpu -x -q -k "device_qft or build_zt_mpo_batch" 2>&1 | tail -15
python3 - <<'PY'
import time, numpy as np, sys
sys.path.insert(0, '.')
import qilaplace_jl_amd as qil
ctx = qil.default_context()
for n in (24, 30):
    for rep in range(2):
        ctx.synchronize(); t0 = time.perf_counter(); Q = qil.zt_qft_chain_device(n); ctx.synchronize(); td = time.perf_counter() - t0
    from qilaplace_jl_amd import builders
    builders._ZT_Q_CACHE.clear()
    t0 = time.perf_counter(); Qh = qil.zt_qft_chain_tensors(n); th = time.perf_counter() - t0
    t0 = time.perf_counter(); W = qil.build_qft_mpo(n, device=True); ctx.synchronize(); tq = time.perf_counter() - t0
    t0 = time.perf_counter(); Wh = qil.qft_mpo_tensors(n); tqh = time.perf_counter() - t0
    print(f"n={n}: paired QFT chain device {td*1e3:.1f} ms (bonds max {max(Q.bond_dims)}), host {th*1e3:.1f} ms; build_qft_mpo device {tq*1e3:.1f} ms (max bond {max(W.bond_dims)}), host {tqh*1e3:.1f} ms", flush=True)
    for qft in ("host", "device"):
        builders._ZT_Q_CACHE.clear()
        ctx.synchronize(); t0 = time.perf_counter(); Z = qil.build_zt_mpo_batch(n, [2 * np.pi], qft=qft)[0]; ctx.synchronize()
        print(f"   build_zt_mpo_batch qft={qft}: {(time.perf_counter() - t0)*1e3:.1f} ms, max bond {max(Z.bond_dims)}", flush=True)
PY
