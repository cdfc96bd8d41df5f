#!/usr/bin/env python3
"""bench.py -- MPO x MPS site-contractions/sec on MI355X (BASELINE.json metric).

A "step" is one `apply(W, psi)` over one synthetic n-qubit paired-register signal: 48 site
contractions at the metric configuration (n=24 zT layout, chi_s=64, chi_c=128, complex128 output,
80.06 GB written per step).  Inputs are resident in HBM before the timed
region; the output is a fresh device MPS each step (caching pool).  With --gpus N > 1 every rank
applies the operator to its OWN independent signal (weak scaling, no data-path collective); RCCL
is used only for the barrier, the max-over-ranks time and the final gather of the coefficient
samples.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel = site_apply_grouped, HBM-store
bound; algorithmic bytes / live HIP-event kernel time) and `cpu_baseline` (the numpy oracle timed
on this box's host cores on a bounded sample of the same workload).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md: 8.0 TB/s; 6.29 TB/s measured copy)
HBM_COPY_GBS = 6290.0

WORKLOADS = {
    # name: (sites L, paired, chi cap, D cap, description)
    "zt_n24_chi64_D128": (48, True, 64, 128,
                          "n=24 paired register (48 sites) zT-layout apply, chi_s=64, chi_c=128, "
                          "saturated bond profiles, f64 MPS x c64 MPO -> c64"),
    "qft_n24_chi64_D128": (24, False, 64, 128, "n=24 single register apply, chi_s=64, chi_c=128"),
    "qft_n20_chi32_D64": (20, False, 32, 64, "n=20 single register apply, chi_s=32, chi_c=64 (configs[1])"),
    "tiny": (12, False, 16, 32, "debug size"),
}


def profiles(L, chi, D):
    cb = [int(min(2 ** (i + 1), 2 ** (L - 1 - i), chi)) for i in range(L - 1)]
    db = [int(min(4 ** (i + 1), 4 ** (L - 1 - i), D)) for i in range(L - 1)]
    return cb, db


def algorithmic_bytes(cb, db, w_bytes=16, a_bytes=8, o_bytes=16):
    """per site  out*(Dl chil)*2*(Dr chir) [write B once] + W + A read once."""
    c = [1] + cb + [1]
    d = [1] + db + [1]
    tot = 0
    for i in range(len(c) - 1):
        tot += o_bytes * (d[i] * c[i]) * 2 * (d[i + 1] * c[i + 1])
        tot += w_bytes * d[i] * 4 * d[i + 1] + a_bytes * c[i] * 2 * c[i + 1]
    return tot


def cpu_baseline(qil, W, psi, cb, db, L, budget_s=30.0):
    """Time the numpy oracle (`oracle.apply_site`, the reference's K=2 GEMM + permute formulation,
    apply.jl:101,114,118) on this box's host cores.  Pass 1 times every distinct site shape once to
    estimate the whole apply; if the estimate fits the budget (about 10-30 s of CPU work) the oracle
    then applies ALL sites and that wall time is the baseline; otherwise the per-shape times are
    summed over the sites (stated in `sample`)."""
    import oracle as O
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count() or 1
    c = [1] + cb + [1]
    d = [1] + db + [1]
    shapes = {}
    for i in range(L):
        shapes.setdefault((d[i], d[i + 1], c[i], c[i + 1]), []).append(i)
    size = lambda s: s[0] * s[1] * s[2] * s[3]
    t_shape, spent = {}, 0.0
    for shp in sorted(shapes, key=size):
        if 32 * size(shp) > 8e9 or (spent > budget_s and t_shape):
            continue
        i = shapes[shp][0]
        Wi, Ai = W.site(i), psi.site(i)
        t0 = time.perf_counter()
        B = O.apply_site(Wi, Ai)
        dt = time.perf_counter() - t0
        del B
        t_shape[shp] = dt
        spent += dt
    big = max(t_shape, key=size)
    est = sum(t_shape.get(s, t_shape[big] * size(s) / size(big)) * len(idx) for s, idx in shapes.items())
    if len(t_shape) == len(shapes) and est <= budget_s:
        Wh, Ah = W.to_host(), psi.to_host()
        t0 = time.perf_counter()
        for Wi, Ai in zip(Wh, Ah):
            B = O.apply_site(Wi, Ai)
            del B
        total = time.perf_counter() - t0
        sample = (f"oracle.apply_site (numpy K=2 GEMM + permute) over ALL {L} sites of the same workload, "
                  f"one pass, {total:.1f} s wall")
    else:
        total = est
        sample = (f"oracle.apply_site timed once per distinct site shape ({len(t_shape)} of {len(shapes)} shapes, "
                  f"{spent:.1f} s of CPU work); full-apply time = sum over the {L} sites of their shape's time "
                  f"({total:.1f} s estimated)")
    return {"value": L / total, "unit": "site-contractions/s", "cores": int(threads), "kind": "port",
            "sample": sample}


def spawn_ranks(n):
    """One child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in its environment), started by a parent
    that never initialises the GPU.  stdout of rank 0 is relayed (its last line is the JSON result); the exit code is
    the first non-zero child code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        code = p.wait()
        rc = rc or code
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="zt_n24_chi64_D128", choices=sorted(WORKLOADS))
    ap.add_argument("--queries", type=int, default=64, help="coefficient samples for max|coeff err|")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process touches no GPU (nothing HIP- or torch-related
        # has been imported yet), starts one child per GPU and relays rank 0's JSON line
        sys.exit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    # QIL_BENCH_FORCE_DIST=1 exercises the RCCL code path (barrier, all_reduce, gather) even at N=1
    if world > 1 or os.environ.get("QIL_BENCH_FORCE_DIST") == "1":
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
        world = dist.get_world_size()
    if world != args.gpus and os.environ.get("QIL_BENCH_FORCE_DIST") != "1":
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")

    import qilaplace_jl_amd as qil
    ctx = qil.Context(local_rank)
    qil.set_default_context(ctx)

    L, paired, chi, D, desc = WORKLOADS[args.workload]
    cb, db = profiles(L, chi, D)
    mps_cls = qil.ZTMPS if paired else qil.SignalMPS
    mpo_cls = qil.PairedSiteMPO if paired else qil.SingleSiteMPO
    # synthetic, seeded, generated on the device: i.i.d. N(0,1)-scaled site tensors
    psi = mps_cls.alloc(cb, dtype=np.float64, amplitude=1.0, ctx=ctx).fill_random(20240064 + rank)
    W = mpo_cls.alloc(db, dtype=np.complex128, ctx=ctx).fill_random(777)
    abytes = algorithmic_bytes(cb, db)

    def barrier():
        ctx.synchronize()
        if dist is not None:
            import torch
            dist.barrier()
            torch.cuda.synchronize()

    out = None
    for _ in range(args.warmup):
        del out
        out = qil.apply(W, psi)
    barrier()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        del out                      # the previous result's blocks go back to the pool
        out = qil.apply(W, psi)
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    n_launch, kernel_ms = ctx.profile_read(reset=True)

    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- accuracy: sampled coefficients of the materialised W*psi vs the lazy HIP path and vs
    # the CPU oracle (lazy restatement on the same W, psi)
    rng = np.random.default_rng(12345)
    bits = rng.integers(0, 2, size=(args.queries, L)).astype(np.uint8)
    c_mat = qil.coefficient_batch(out, bits)
    c_lazy = qil.apply_coefficient_batch(W, psi, bits)
    scale = max(np.abs(c_mat).max(), 1e-300)
    err_lazy = float(np.abs(c_mat - c_lazy).max() / scale)
    err_oracle = None
    if rank == 0 and not args.no_cpu_baseline:
        import oracle as O
        Wh = O.SingleSiteMPO(W.to_host())
        ph = O.SignalMPS(psi.to_host(), amplitude=psi.amplitude)
        c_ref = O.lazy_coefficient_batch(Wh, ph, bits)
        err_oracle = float(np.abs(c_mat - c_ref).max() / max(np.abs(c_ref).max(), 1e-300))
    if dist is not None:
        import torch
        mine = torch.tensor(np.stack([c_mat.real, c_mat.imag], -1), device=f"cuda:{local_rank}")
        gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        dist.gather(mine, gathered, dst=0)           # the one RCCL data collective (KB-scale)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        k_ms = kernel_ms / max(n_launch, 1)
        achieved = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload)
            except Exception:
                traffic = None
        res = {
            "metric": "MPO×MPS site-contractions/sec + max |coeff err|, n=24 χ_s=64 χ_c=128",
            "value": L * world / (elapsed / args.steps),
            "unit": "site-contractions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "description": desc, "sites": L,
                       "mps_bonds_max": chi, "mpo_bonds_max": D,
                       "output_bytes_per_step": abytes, "parallelism": f"replicas x{world} (one signal per GPU)"},
            "max_coeff_err": err_oracle if err_oracle is not None else err_lazy,
            "coeff_err": {"materialised_vs_lazy_hip": err_lazy, "materialised_vs_cpu_oracle": err_oracle,
                          "queries": args.queries, "kind": "max relative"},
            "roofline": {"bound": "hbm", "kernel": "site_apply_grouped<c64,double>",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak": achieved / HBM_COPY_GBS,
                         "traffic": traffic, "kernel_ms": k_ms, "launches_timed": n_launch,
                         "algorithmic_bytes_per_launch": abytes},
        }
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(qil, W, psi, cb, db, L)
        try:                                   # RCCL prints a banner through C stdio: flush it first so
            import ctypes                      # the JSON line is the last line of stdout
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
