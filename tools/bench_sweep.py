#!/usr/bin/env python3
"""BASELINE.json configs[3]: n=24, one structured paired-register signal x 64 damping values sigma
(build_dt_mpo sweep), sharded round-robin over the ranks (one process per GPU), one RCCL all_gather of
the per-sigma coefficient batches.  Launch: python tools/bench_sweep.py            (1 GPU)
        or: python -m torch.distributed.run --nproc-per-node N tools/bench_sweep.py --gpus N
Prints one JSON line on rank 0."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--n", type=int, default=24)
    ap.add_argument("--sigmas", type=int, default=64)
    ap.add_argument("--samples", type=int, default=1024)
    ap.add_argument("--workers", type=int, default=32)
    ap.add_argument("--host-build", action="store_true", help="build the MPOs with the host process pool instead of "
                    "the batched device builder")
    args = ap.parse_args()
    rank, local, world = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("LOCAL_RANK", 0), ("WORLD_SIZE", 1)))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    import qilaplace_jl_amd as qil
    ctx = qil.Context(local)
    qil.set_default_context(ctx)
    n, N = args.n, 2 ** args.n
    j = np.arange(N, dtype=np.float64)
    rng = np.random.default_rng(1001)                       # :multi_sin_exp-like structured signal (Signals.jl:64-85)
    ak = rng.random(10); ak /= np.linalg.norm(ak)
    wk = 40.0 / N * (rng.random(10) - 0.5)
    lk = -2.0 / N * rng.random(10)
    x = sum(ak[k] * np.sin(wk[k] * j) * np.exp(lk[k] * j) for k in range(10))
    t0 = time.perf_counter()
    psi = qil.signal_ztmps(x, method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
    ctx.synchronize()
    t_encode = time.perf_counter() - t0
    sig = np.linspace(0.25, 16.0, args.sigmas)
    bits, _, _ = qil.damping_sample_bits(n, args.samples, seed=7)      # non-negligible closed-form values (sweep.py)
    mine = qil.shard_items(len(sig), world, rank)
    import torch  # noqa: F401  (kept out of the timed region)
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    if args.host_build:
        tensors = qil.dt_mpo_tensors_many(n, [sig[i] for i in mine], workers=args.workers)   # host process pool
        mpos = [qil.PairedSiteMPO(W, sites=psi.site_ids, ctx=ctx) for W in tensors]
    else:
        mpos = qil.build_dt_mpo_batch(psi, [sig[i] for i in mine], ctx=ctx)                  # one device batch
    ctx.synchronize()
    t_build = time.perf_counter() - t0
    t1 = time.perf_counter()
    local_res = {}
    for i, W in zip(mine, mpos):
        out = W * psi
        local_res[i] = qil.coefficient_batch(out, bits)
        del out
    ctx.synchronize()
    t_apply = time.perf_counter() - t1
    res = qil.gather_results(local_res, len(sig), args.samples, dist, f"cuda:{local}" if dist is not None else None)
    total = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([total, t_build, t_apply], dtype=torch.float64, device=f"cuda:{local}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        total, t_build, t_apply = (float(v) for v in tt)
    if rank == 0:
        # accuracy of a few entries against the closed form  x_j e^{-wr k j/N}/sqrt(N)  (copy bits = j msb, main bits = k lsb)
        main = bits[:, 0::2].astype(np.int64)
        copy = bits[:, 1::2].astype(np.int64)
        kk = (main * (1 << np.arange(n))[None, :]).sum(1)
        jj = (copy * (1 << np.arange(n - 1, -1, -1))[None, :]).sum(1)
        err = 0.0
        for r in (0, len(sig) // 2, len(sig) - 1):
            ref = x[jj] * np.exp(-sig[r] * kk * jj / N) / np.sqrt(N)
            err = max(err, float(np.abs(res[r] - ref).max() / max(np.abs(x).max() / np.sqrt(N), 1e-300)))
        print(json.dumps({
            "case": "dt_sigma_sweep", "n": n, "sigmas": len(sig), "samples": args.samples, "n_gpus": world,
            "mps_bonds_max": max(psi.bond_dims), "seconds_total": total, "seconds_build": t_build, "builder": "host-pool" if args.host_build else "device-batch",
            "seconds_apply_and_sample": t_apply, "seconds_encode": t_encode,
            "site_contractions_per_s": len(sig) * 2 * n / total,
            "site_contractions_per_s_apply_only": len(sig) * 2 * n / t_apply,
            "max_err_vs_closed_form_rel_to_peak": err, "host_build_workers": args.workers}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
