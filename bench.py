#!/usr/bin/env python3
"""bench.py -- MPO x MPS site-contractions/sec on MI355X (BASELINE.json metric).

Default workload (`zt_n24_chi64_D128`, the configuration the metric is quoted on): a "step" is one
`apply(W, psi)` over one synthetic n-qubit paired-register signal: 48 site contractions, chi_s=64,
chi_c=128, complex128 output, 80.06 GB written per step.  Inputs are resident in HBM
before the timed region; the output is a fresh device MPS each step (caching pool).  With --gpus N > 1
every rank applies the operator to its OWN independent signal (weak scaling, no data-path collective);
RCCL is used only for the barrier, the max-over-ranks time and the final gather of the coefficient samples.

`--workload dt_sweep_n24_s64` (BASELINE.json configs[3]): a step is one whole damping sweep -- 64 DT MPOs built
by the persistent device builder, applied to one encoded n=24 signal and sampled at 1024 configurations each --
with the 64 values dealt round-robin to the ranks (strong scaling) and one RCCL all_gather of the samples.
`--workload dt_sweep_n24_weak`: the same sweep with 64 values PER RANK (64 N values in all, weak scaling): the builder launch
is one chain's latency whatever the share (one workgroup per value, up to one per CU), so the strong-scaling form is expected
to give ~1.05x at 8 GPUs (`expected_speedup_at_8` in its line) while this one scales with the number of values.

`python bench.py --gpus N` without a launcher starts the N ranks itself (before anything touches the GPU).

Prints ONE JSON line (rank 0) with
  roofline      dominant kernel = site_apply_grouped (HBM-store bound): algorithmic bytes / live HIP-event time
  truncate      the other half of "apply-and-truncate": the fused apply_compress and the exact compress!(apply)
                on the zT product of an encoded n=24 signal (chi 15 x D ~89, 48 sites, maxdim 64), with an MFMA
                flop model against the f64 matrix peak
  cpu_baseline  the C++/OpenMP CPU backend behind the same C ABI (oracle/cpu, all cores and one thread) and the
                numpy port of the oracle, timed on this box's host cores on the same workload
"""
import argparse
import hashlib
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
F64_MFMA_PEAK_TFLOPS = 78.6    # v_mfma_f64_16x16x4: 2048 flop / 64 cycles / SIMD, 1024 SIMDs, 2.4 GHz

WORKLOADS = {
    # name: (sites L, paired, chi cap, D cap, description)
    "zt_n24_chi64_D128": (48, True, 64, 128,
                          "n=24 paired register (48 sites) zT-layout apply, chi_s=64, chi_c=128, "
                          "saturated bond profiles, f64 MPS x c64 MPO -> c64"),
    "qft_n24_chi64_D128": (24, False, 64, 128, "n=24 single register apply, chi_s=64, chi_c=128"),
    "qft_n20_chi32_D64": (20, False, 32, 64, "n=20 single register apply, chi_s=32, chi_c=64 (configs[1])"),
    "tiny": (12, False, 16, 32, "debug size"),
    "dt_sweep_n24_s64": (48, True, 0, 0, "n=24 signal x 64 damping values: build_dt_mpo sweep, apply, 1024 samples each "
                                        "(configs[3]); values dealt round-robin to the ranks"),
    "dt_sweep_n24_weak": (48, True, 0, 0, "n=24 signal x 64 damping values PER RANK (64 N values in linspace(0.25, 16) dealt round-robin): "
                                         "the axis of configs[3] that scales -- one builder launch per rank whatever its share"),
}


def profiles(L, chi, D):
    cb = [int(min(2 ** (i + 1), 2 ** (L - 1 - i), chi)) for i in range(L - 1)]
    db = [int(min(4 ** (i + 1), 4 ** (L - 1 - i), D)) for i in range(L - 1)]
    return cb, db


from bench_configs import algorithmic_bytes, embed_and_gauge  # noqa: E402  (SURVEY.md 8d bytes figure; cfg3's operand recipe)


def lib_sha16():
    import qilaplace_jl_amd as qil
    with open(qil.LIB_PATH, "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def pmc_traffic(workload):
    """HBM bytes per launch of the dominant kernel from the rocprofv3 PMC passes of THIS round
    (profiles/<round>_pmc_traffic.json, written by tools/collect_pmc.py from separate WRITE_SIZE / FETCH_SIZE passes
    with the guide's unit and gfx950 corrections).  Only reported when the counters were collected with the
    library binary that is running now (sha recorded next to them); otherwise null."""
    import glob
    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc_traffic.json")))
    if not paths:
        return None, None
    path = paths[-1]
    try:
        rec = json.load(open(path))
        if rec.get("lib_sha16") != lib_sha16():
            return None, f"{os.path.basename(path)} was collected with another build of libqilhip.so"
        return rec.get(workload), os.path.basename(path)
    except Exception:
        return None, None


def pmc_truncate():
    """Matrix-core counters of the truncate half (tools/collect_pmc_truncate.py), for THIS build of the library only."""
    import bench_configs
    return bench_configs.pmc_record()


# ---------------------------------------------------------------------------------------------- CPU baseline
def numpy_port_time(W, psi, cb, db, L, budget_s=20.0):
    """The numpy oracle (`oracle.apply_site`, the reference's K=2 GEMM + permute formulation, apply.jl:101,114,118):
    every distinct site shape timed once, summed over the sites (shapes whose result exceeds 8 GB scaled from the
    largest timed one)."""
    import oracle as O
    c = [1] + list(cb) + [1]
    d = [1] + list(db) + [1]
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
    total = sum(t_shape.get(s, t_shape[big] * size(s) / size(big)) * len(idx) for s, idx in shapes.items())
    return total, len(t_shape), len(shapes), spent


def cpu_baseline(W, psi, cb, db, L):
    """SURVEY.md 8d "CPU baseline beside it": (1) the C++/OpenMP backend behind the same C ABI (oracle/cpu/libqilcpu.so):
    one apply over ALL sites of the same (W, psi), all cores and one thread; (2) the numpy port of the oracle."""
    import subprocess
    from oracle.cpu_backend import CpuBackend
    lib = os.path.join(ROOT, "oracle", "cpu", "lib", "libqilcpu.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "cpu")], check=True, stdout=subprocess.DEVNULL)
    cpu = CpuBackend(lib)
    Wh, Ah = W.to_host(), psi.to_host()
    # `nproc` of a container can exceed the cores it really gets (GPU boxes: 256 visible, cgroup quota 16): no team is larger
    # than the quota; the team size is the fastest of these on the largest site
    avail = cpu_quota()
    teams = sorted({t for t in (avail, avail // 2, avail // 4, 64, 32, 16, 8, 4) if 1 <= t <= avail}, reverse=True)
    t_all, cores, nbytes = cpu.time_apply(Wh, Ah, teams, reps=2)
    t_one, _, _ = cpu.time_apply(Wh, Ah, 1, reps=1)
    t_np, ns, nshapes, spent = numpy_port_time(W, psi, cb, db, L)
    try:
        from threadpoolctl import threadpool_info
        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    return {
        "value": L / t_all, "unit": "site-contractions/s", "cores": int(cores), "kind": "port",
        "sample": (f"C++/OpenMP backend behind the same C ABI (oracle/cpu/libqilcpu.so): one apply over ALL {L} sites of the "
                   f"same (W, psi), every site written into one reusable buffer of the largest site's size, mean of 2 passes "
                   f"after a dry run: {t_all:.2f} s = {nbytes / t_all / 1e9:.1f} GB/s of output on {cores} threads, the fastest team "
                   f"size of {teams} on the largest site (nproc = {os.cpu_count()}, cgroup quota / affinity = {avail})"),
        "cpu_quota": avail,
        "gb_per_s": nbytes / t_all / 1e9,
        "single_thread": {"value": L / t_one, "seconds_per_apply": t_one, "gb_per_s": nbytes / t_one / 1e9, "cores": 1},
        "numpy_port": {"value": L / t_np, "seconds_per_apply": t_np, "cores": int(blas_threads),
                       "sample": f"oracle.apply_site (numpy K=2 GEMM + permute) timed once per distinct site shape "
                                 f"({ns} of {nshapes} shapes, {spent:.1f} s of CPU work), summed over the {L} sites"},
    }


# ---------------------------------------------------------------------------------------------- truncate block
def svd_flops(m, n):
    """Golub-Van Loan count of a thin SVD with both factors (6 m n^2 + 20 n^3, n = short side)."""
    n, m = min(m, n), max(m, n)
    return 6.0 * m * n * n + 20.0 * n ** 3


def _host_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([p.get("num_threads", 1) for p in threadpool_info()] or [1]))
    except Exception:
        return int(os.cpu_count() or 1)


def truncate_roofline(t_exact, t_one, f_exact_model, f_fused_model, t_fused):
    """f64 matrix-core work of the truncate half against the f64 MFMA peak: from the PMC passes of this build when they exist
    (counted MFMA flops / measured time, and the share of SIMD cycles the matrix pipe was busy), else the Golub-Van Loan flop
    model of r02 (stated as such)."""
    pmc, src = pmc_truncate()
    if pmc:
        ex, c2 = pmc["exact_compress_product_1008"], pmc["compress_chi256_24_sites"]
        simd_cycles = lambda t: t * 2.4e9 * 1024.0                       # 256 CUs x 4 SIMDs at 2.4 GHz
        return {"bound": "mfma", "unit": "TFLOP/s", "peak": F64_MFMA_PEAK_TFLOPS, "source": src,
                "achieved": ex["mfma_f64_flops"] / t_exact / 1e12,
                "frac": ex["mfma_f64_flops"] / t_exact / 1e12 / F64_MFMA_PEAK_TFLOPS,
                "mfma_f64_flops_per_exact_compress": ex["mfma_f64_flops"],
                "mfma_busy_share_of_simd_cycles": ex["mfma_busy_cycles"] / simd_cycles(t_exact),
                "compress_chi256": {"mfma_f64_flops": c2["mfma_f64_flops"], "achieved": c2["mfma_f64_flops"] / t_one / 1e12,
                                    "frac": c2["mfma_f64_flops"] / t_one / 1e12 / F64_MFMA_PEAK_TFLOPS,
                                    "mfma_busy_share_of_simd_cycles": c2["mfma_busy_cycles"] / simd_cycles(t_one)},
                "model": "counted: SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 and SQ_VALU_MFMA_BUSY_CYCLES per repetition (separate rocprofv3 "
                         "--pmc passes of this build, tools/collect_pmc_truncate.py) over the times measured here; the chains are "
                         "latency-bound sequences of small factorisations, so this fraction is the honest distance to the matrix peak.  "
                         "(Since r05 complex products issue THREE real MFMA multiplications instead of four -- Gauss's trick, DESIGN.md 3.4 --, "
                         "so the same complex work counts 25 % fewer MFMA flops than in the r04 records at equal time.)"}
    return {"bound": "mfma", "unit": "TFLOP/s", "peak": F64_MFMA_PEAK_TFLOPS, "source": src,
            "achieved": f_exact_model / t_exact / 1e12, "frac": f_exact_model / t_exact / 1e12 / F64_MFMA_PEAK_TFLOPS,
            "achieved_fused": f_fused_model / t_fused / 1e12,
            "model": "Golub-Van Loan thin-SVD flops 6mn^2+20n^3 of the truncated SVDs (+2mnk of the fused route's GEMMs), shapes "
                     "from the bond profiles (no PMC file for this build of the library under profiles/)"}


def cpu_quota():
    """CPUs this process may really use: min(affinity mask, cgroup quota).  The GPU boxes show 256 cores and grant 16."""
    try:
        avail = float(len(os.sched_getaffinity(0)))
    except AttributeError:
        avail = float(os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            avail = min(avail, float(q) / float(per))
    except (OSError, ValueError):
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                avail = min(avail, q / per)
        except (OSError, ValueError):
            pass
    return max(1, int(avail))


def truncate_signal(n=24):
    """The structured n-qubit signal of the truncate block (also the operand recipe of
    tests/test_gpu_parity.py::test_bench_truncate_operands_against_oracle): two damped tones + six seeded slow sines."""
    N = 2 ** n
    j = np.arange(N, dtype=np.float64)
    x = np.sin(2 * np.pi * 5.0 * j / N) * np.exp(-3.0 * j / N) + 0.5 * np.cos(2 * np.pi * 11.0 * j / N)
    rng = np.random.default_rng(1001)
    return x + sum(0.1 * rng.random() * np.sin(40.0 * (rng.random() - 0.5) * j / N) for _ in range(6))


TRUNCATE_MAXDIM, TRUNCATE_TOL = 64, 1e-8


def truncate_operands(qil, n=24):
    """(W, psi) of the truncate block: signal_ztmps(:rsvd, k=15, p=5, q=2, cutoff=1e-12) of truncate_signal(n) and the
    genuine build_zt_mpo(psi, 2 pi) -- product bond ~1008 at n = 24."""
    psi = qil.signal_ztmps(truncate_signal(n), method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
    return qil.build_zt_mpo(psi, 2 * np.pi), psi


def truncate_roofline_batch64(t_batch):
    """f64 matrix-core work of ONE 64-pair qil_apply_compress_batch (counted by the PMC pass of this build,
    tools/collect_pmc_truncate.py) over the batch time measured here."""
    pmc, src = pmc_truncate()
    if not pmc or "apply_compress_batch64_zt" not in pmc:
        return {"bound": "mfma", "unit": "TFLOP/s", "peak": F64_MFMA_PEAK_TFLOPS, "achieved": None, "frac": None, "source": src,
                "model": "no PMC pass of this build for the 64-pair batch under profiles/"}
    b = pmc["apply_compress_batch64_zt"]
    return {"bound": "mfma", "unit": "TFLOP/s", "peak": F64_MFMA_PEAK_TFLOPS, "source": src,
            "achieved": b["mfma_f64_flops"] / t_batch / 1e12, "frac": b["mfma_f64_flops"] / t_batch / 1e12 / F64_MFMA_PEAK_TFLOPS,
            "mfma_f64_flops_per_batch": b["mfma_f64_flops"], "dispatches_per_batch": b["dispatches"],
            "mfma_busy_share_of_simd_cycles": b["mfma_busy_cycles"] / (t_batch * 2.4e9 * 1024.0),
            "model": "counted: SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 per batch (3-repetition minus 1-repetition PMC run) over the batch time measured here "
                     "(complex products count three real multiplications since r05, four in the r04 records)"}


def truncate_block(qil, ctx, reps=3, cpu=True):
    """The 'truncate' of apply-and-truncate on the pipeline's own operands (cfg4-shaped): n=24 structured signal
    encoded to chi ~15, genuine zT MPO (D ~89), product bond ~1008, truncated to maxdim 64 at tol 1e-8.  Every figure is the
    mean of `reps` runs after one dry run (the reference's method: 1 dry run + 5 samples, scripts/benchmark/qft_vs_fftw.jl:126-127);
    the minimum is printed beside it."""
    n = 24
    W, psi = truncate_operands(qil, n)
    maxdim, tol = TRUNCATE_MAXDIM, TRUNCATE_TOL
    fused = qil.apply_compress(W, psi, maxdim=maxdim, tol=tol)          # warm-up (pool, code objects)
    ctx.synchronize()
    tf = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fused = qil.apply_compress(W, psi, maxdim=maxdim, tol=tol)
        ctx.synchronize()
        tf.append(time.perf_counter() - t0)
    t_fused = sum(tf) / reps
    ta, te = [], []
    prod_host = None
    for r in range(reps + 1):                                           # run 0 is the dry run of the exact route
        t0 = time.perf_counter()
        prod = W * psi
        ctx.synchronize()
        t1 = time.perf_counter()
        if cpu and r == 0:
            prod_host = (prod.to_host(), prod.amplitude)                # the SAME product for the CPU baseline below
        t2 = time.perf_counter()
        qil.compress(prod, maxdim=maxdim, tol=tol)
        ctx.synchronize()
        t3 = time.perf_counter()
        if r:
            ta.append(t1 - t0)
            te.append(t3 - t2)
    t_apply, t_exact = sum(ta) / reps, sum(te) / reps
    # batch of 8 independent chains (qil_compress_batch) against one chain alone: compress! chi 256 -> 128 on 24 sites
    def sat(L, chi):
        return [int(min(2 ** (i + 1), 2 ** (L - 1 - i), chi)) for i in range(L - 1)]

    def chains(k):
        return [qil.SignalMPS.alloc(sat(24, 256), dtype=np.float64).fill_random(5 + i) for i in range(k)]

    t_one = t_eight = None
    t_many = {}
    one_host = None
    for _ in range(2):                                                   # first round warms the pool and the worker streams
        one, eight = chains(1)[0], chains(8)
        if cpu and one_host is None:
            one_host = one.to_host()
        ctx.synchronize()
        t0 = time.perf_counter()
        qil.compress(one, maxdim=128, tol=1e-10)
        ctx.synchronize()
        t_one = time.perf_counter() - t0
        t0 = time.perf_counter()
        qil.compress_batch(eight, maxdim=128, tol=1e-10)
        ctx.synchronize()
        t_eight = time.perf_counter() - t0
        del eight
        for nbig in (16, 32, 64):
            many = chains(nbig)
            ctx.synchronize()
            t0 = time.perf_counter()
            qil.compress_batch(many, maxdim=128, tol=1e-10)
            ctx.synchronize()
            t_many[nbig] = time.perf_counter() - t0
            del many
    del one
    # The batch as the operating point of the truncate half: the 64 (operator, state) pairs of a cfg4-shaped
    # damping sweep -- 64 genuine zT operators of linspace(0.25, 16, 64) x the encoded signal -- through ONE
    # qil_apply_compress_batch (fused apply-and-truncate per pair, maxdim 64, tol 1e-8), against one pair alone
    sig64 = np.linspace(0.25, 16.0, 64)
    W64 = qil.build_zt_mpo_batch(psi, sig64)
    ctx.synchronize()
    t_pair = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        one_pair = qil.apply_compress(W64[32], psi, maxdim=maxdim, tol=tol)
        ctx.synchronize()
        if r:
            t_pair.append(time.perf_counter() - t0)
    t_b64 = []
    for r in range(reps + 1):
        t0 = time.perf_counter()
        outs64 = qil.apply_compress_batch(W64, psi, maxdim=maxdim, tol=tol)
        ctx.synchronize()
        if r:
            t_b64.append(time.perf_counter() - t0)
    bits64 = np.random.default_rng(4).integers(0, 2, size=(64, 2 * n)).astype(np.uint8)
    c_b, c_1 = qil.coefficient_batch(outs64[32], bits64), qil.coefficient_batch(one_pair, bits64)
    batch64 = {
        "op": "qil_apply_compress_batch: 64 (zT operator, signal) pairs, wr = linspace(0.25, 16, 64), n=24 paired, maxdim 64, tol 1e-8",
        "product_bond_max": int(max(max(c * d for c, d in zip(psi.bond_dims, Wk.bond_dims)) for Wk in W64)),
        "one_pair_ms": sum(t_pair) / reps * 1e3, "batch_ms": sum(t_b64) / reps * 1e3, "batch_ms_min": min(t_b64) * 1e3,
        "batch_over_one_pair": (sum(t_b64) / reps) / (sum(t_pair) / reps), "pairs_per_s": 64 / (sum(t_b64) / reps),
        "pairs_per_s_one_at_a_time": 1 / (sum(t_pair) / reps), "site_truncations_per_s": 64 * 2 * n / (sum(t_b64) / reps),
        "bonds_max": int(max(max(o.bond_dims) for o in outs64)),
        "item_32_equals_the_pair_alone": bool(np.array_equal(c_b, c_1)),
    }
    t_batch64 = sum(t_b64) / reps
    del W64, outs64, one_pair
    bits = np.random.default_rng(3).integers(0, 2, size=(256, 2 * n)).astype(np.uint8)
    c_f, c_e = qil.coefficient_batch(fused, bits), qil.coefficient_batch(prod, bits)
    c_x = qil.apply_coefficient_batch(W, psi, bits)
    scale = np.abs(c_x).max()
    P = [1] + [c * d for c, d in zip(psi.bond_dims, W.bond_dims)] + [1]
    cpu_res = None
    if cpu:
        # the reference-shaped CPU path beside every reported figure -- the numpy / LAPACK restatement of
        # compress! (oracle.compress = src/mps.jl:913-973: canonicalize!, two-site SVD sweeps, canonicalize!) on the SAME
        # downloaded tensors, timed on this box's host cores
        import oracle as O
        from threadpoolctl import threadpool_limits

        quota = cpu_quota()

        def timed_compress(host, amp, md, tl):
            # LAPACK on a 256-thread box loses to itself: the BLAS team is capped -- never above the cgroup quota, which is all
            # the process really gets --, fastest of two team sizes reported
            best = None
            for team in sorted({min(8, quota), min(32, quota)}):
                with threadpool_limits(limits=team):
                    obj = O.SignalMPS([a.copy() for a in host], amplitude=amp)
                    t0 = time.perf_counter()
                    O.compress(obj, maxdim=md, tol=tl)
                    dt_ = time.perf_counter() - t0
                if best is None or dt_ < best[0]:
                    best = (dt_, team, obj)
            return best

        t_cpu_exact, team_exact, ph = timed_compress(prod_host[0], prod_host[1], maxdim, tol)
        c_cpu = O.coefficient_batch(ph, bits)
        t_cpu_one, team_one, _ = timed_compress(one_host, 1.0, 128, 1e-10)
        cpu_res = {
            "kind": "port", "cores": int(team_exact), "cores_chi256": int(team_one), "nproc": os.cpu_count(), "cpu_quota": quota,
            "sample": "oracle.compress (numpy + LAPACK gesdd restatement of compress!, src/mps.jl:913-973) on the same downloaded "
                      "tensors: the whole bond-%d product (48 sites) and the whole chi 256 -> 128 chain (24 sites), BLAS team capped at min(8, quota) and "
                      "min(32, quota) threads, the faster run of each reported"
                      % max(P),
            "exact_compress_ms": t_cpu_exact * 1e3, "compress_chi256_to_128_24_sites_ms": t_cpu_one * 1e3,
            "value": 2 * n / t_cpu_exact, "unit": "site-truncations/s",
            "gpu_over_cpu_exact_compress": t_cpu_exact / t_exact, "gpu_over_cpu_chi256": t_cpu_one / t_one,
            "bonds_exact_max": int(max(ph.bond_dims)),
            "err_cpu_route_vs_exact_product": float(np.abs(c_cpu - c_x).max() / scale),
            "hip_vs_cpu_truncated_state": float(np.abs(c_e - c_cpu).max() / scale),
        }
    # flop models (stated, not measured): exact route = one gauge pass of truncated SVDs over the product sites
    # (P_l x 2 P_r, the later passes run at <= maxdim); fused route = the zip-up's theta SVDs (2 r x D chi) with
    # r <= 2 maxdim plus its contraction GEMMs
    f_exact = sum(svd_flops(P[i], 2 * P[i + 1]) for i in range(1, len(P) - 1))
    r = 2 * maxdim
    f_fused = sum(svd_flops(2 * min(r, P[i]), P[i + 1]) + 2.0 * 2 * min(r, P[i]) * P[i + 1] * 2 * r
                  for i in range(len(P) - 1))
    return {
        "op": "compress!(apply(W_zt, psi); maxdim=64, tol=1e-8), n=24 paired (48 sites), encoded signal x genuine zT MPO",
        "mps_bonds_max": max(psi.bond_dims), "mpo_bonds_max": max(W.bond_dims), "product_bond_max": max(P),
        "fused_apply_compress_ms": t_fused * 1e3, "exact_apply_ms": t_apply * 1e3, "exact_compress_ms": t_exact * 1e3,
        "repetitions": reps, "fused_apply_compress_ms_min": min(tf) * 1e3, "exact_apply_ms_min": min(ta) * 1e3,
        "exact_compress_ms_min": min(te) * 1e3,
        "site_truncations_per_s_fused": 2 * n / t_fused, "site_truncations_per_s_exact": 2 * n / (t_apply + t_exact),
        "bonds_fused_max": max(fused.bond_dims), "bonds_exact_max": max(prod.bond_dims),
        "err_fused_vs_exact_product": float(np.abs(c_f - c_x).max() / scale),
        "err_exact_route_vs_exact_product": float(np.abs(c_e - c_x).max() / scale),
        "compress_chi256_to_128_24_sites_ms": t_one * 1e3, "compress_batch_of_8_ms": t_eight * 1e3,
        "batch_of_8_over_single": t_eight / t_one,
        "compress_batch_of_16_ms": t_many[16] * 1e3, "batch_of_16_over_single": t_many[16] / t_one,
        "compress_batch_of_32_ms": t_many[32] * 1e3, "batch_of_32_over_single": t_many[32] / t_one,
        "compress_batch_of_64_ms": t_many[64] * 1e3, "batch_of_64_over_single": t_many[64] / t_one,
        "chains_per_s_batch_of_32": 32 / t_many[32], "chains_per_s_batch_of_64": 64 / t_many[64], "chains_per_s_single": 1 / t_one,
        "batch64": batch64,
        "cpu_baseline": cpu_res,
        "roofline": truncate_roofline(t_exact, t_one, f_exact, f_fused, t_fused),
        "roofline_batch64": truncate_roofline_batch64(t_batch64),
    }


# ---------------------------------------------------------------------------------------------- ranks
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
    tag = f"bench{os.getpid()}_{int(time.time() * 1e6) & 0xffffffff:08x}"     # per-job nonce of the C ABI's file rendezvous (sweep.Comm.from_env)
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", QIL_COMM_TAG=tag)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        code = p.wait()
        rc = rc or code
    return rc


class Ranks:
    """RANK / WORLD_SIZE from the launcher's environment; torch.distributed over RCCL when there is more than one.
    QIL_BENCH_FORCE_DIST=1 exercises the collective code path at N=1; QIL_BENCH_BACKEND=gloo (+ every rank on device 0)
    lets a 1-GPU box run the N-rank logic end to end; QIL_BENCH_BACKEND=cabi runs barrier, max-over-ranks and the gather through the
    library's own RCCL communicator (qil_comm_* / qil_gather_coefficients: what a Julia host uses) with no torch in the process."""

    def __init__(self, gpus):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.world_hint = self.world     # known before any communicator exists (sizes the weak-scaling workload)
        self.dist = None
        self.backend = os.environ.get("QIL_BENCH_BACKEND", "nccl")
        self.comm = None                 # QIL_BENCH_BACKEND=cabi: the C ABI's own RCCL communicator (qil_comm_*), no torch at all
        self.want_comm = self.backend == "cabi" and (self.world > 1 or os.environ.get("QIL_BENCH_FORCE_DIST") == "1")
        if self.backend == "gloo":
            self.local_rank = 0
        if self.backend != "cabi" and (self.world > 1 or os.environ.get("QIL_BENCH_FORCE_DIST") == "1"):
            import torch
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            if self.backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.local_rank))
            else:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            self.dist = dist
            self.world = dist.get_world_size()
        if self.world != gpus and os.environ.get("QIL_BENCH_FORCE_DIST") != "1":
            sys.exit(f"bench.py: --gpus {gpus} but the launcher started WORLD_SIZE={self.world} ranks")

    def attach(self, qil, ctx):
        """QIL_BENCH_BACKEND=cabi: create the library's communicator on the rank's context (collective; file rendezvous keyed by
        MASTER_PORT and the launcher's pid, sweep.Comm.from_env)."""
        if self.want_comm and self.comm is None:
            self.comm = qil.Comm.from_env(ctx)
            self.world = self.comm.world

    def _gather_scalars(self, values):
        n = len(values)
        out = self.comm.gather_coefficients({self.rank: np.asarray(values, dtype=np.complex128)}, self.world, n)
        return out.real                                  # (world, n)

    @property
    def backend_used(self):
        """"nccl" (= RCCL on ROCm via torch.distributed) / "gloo" when a process group is up, "rccl-cabi" for the library's own
        communicator, None for a bare single process."""
        if self.comm is not None:
            return "rccl-cabi"
        return self.backend if self.dist is not None else None

    @property
    def device(self):
        return f"cuda:{self.local_rank}" if self.backend == "nccl" else "cpu"

    def barrier(self, ctx):
        ctx.synchronize()
        if self.comm is not None:
            self._gather_scalars([0.0])                  # an all-gather of one value per rank is a barrier
        if self.dist is not None:
            import torch
            self.dist.barrier()
            if self.backend == "nccl":
                torch.cuda.synchronize()

    def max_over_ranks(self, values):
        if self.comm is not None:
            return [float(v) for v in self._gather_scalars(values).max(axis=0)]
        if self.dist is None:
            return list(values)
        import torch
        t = torch.tensor(list(values), dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return [float(v) for v in t]

    def finish(self):
        if self.comm is not None:
            self.comm.close()
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def emit(res):
    try:                                   # RCCL prints a banner through C stdio: flush it first so
        import ctypes                      # the JSON line is the last line of stdout
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    print(json.dumps(res), flush=True)


# ---------------------------------------------------------------------------------------------- workloads
def run_apply(args, rk):
    import qilaplace_jl_amd as qil
    ctx = qil.Context(rk.local_rank)
    qil.set_default_context(ctx)
    rk.attach(qil, ctx)
    rank, world = rk.rank, rk.world
    L, paired, chi, D, desc = WORKLOADS[args.workload]
    cb, db = profiles(L, chi, D)
    mps_cls = qil.ZTMPS if paired else qil.SignalMPS
    mpo_cls = qil.PairedSiteMPO if paired else qil.SingleSiteMPO
    # synthetic, seeded, generated on the device: i.i.d. N(0,1)-scaled site tensors
    psi = mps_cls.alloc(cb, dtype=np.float64, amplitude=1.0, ctx=ctx).fill_random(20240064 + rank)
    genuine = args.workload == "zt_n24_chi64_D128" and not args.random_mpo
    w_natural = None
    if genuine:
        # the GENUINE operator of the metric configuration: build_zt_mpo(24, 2 pi) at its natural bonds (~89), zero-embedded
        # and gauge-mixed to the dense chi_c = 128 profile (same operator, dense tensors: the kernel sees what a
        # random fill shows it); the accuracy leg below checks against the oracle on the NATURAL-bond operator
        w_natural = qil.build_zt_mpo(L // 2, 2 * np.pi, ctx=ctx).to_host()
        W = mpo_cls(embed_and_gauge(w_natural, db, np.random.default_rng(20240128)), ctx=ctx)
    else:
        W = mpo_cls.alloc(db, dtype=np.complex128, ctx=ctx).fill_random(777)
    abytes = algorithmic_bytes(cb, db)
    # the box's own store-only ceiling (8 GiB x 5 by four writers, < 0.1 s, same process, before the timed region): fractions
    # measured on different boxes of the pool become comparable
    box_peak, box_writer = ctx.hbm_store_peak(8 << 30 if args.workload != "tiny" else 1 << 28, 5)

    out = None
    for _ in range(args.warmup):
        del out
        out = qil.apply(W, psi)
    rk.barrier(ctx)
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        del out                      # the previous result's blocks go back to the pool
        out = qil.apply(W, psi)
    rk.barrier(ctx)
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    n_launch, kernel_ms = ctx.profile_read(reset=True)
    elapsed = rk.max_over_ranks([elapsed])[0]

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
        Wh = O.SingleSiteMPO(w_natural if genuine else W.to_host())
        ph = O.SignalMPS(psi.to_host(), amplitude=psi.amplitude)
        c_ref = O.lazy_coefficient_batch(Wh, ph, bits)
        err_oracle = float(np.abs(c_mat - c_ref).max() / max(np.abs(c_ref).max(), 1e-300))
    if rk.comm is not None:                               # the one data collective, through the C ABI (qil_gather_coefficients)
        rk.comm.gather_coefficients({rank: c_mat}, world, len(c_mat))
    if rk.dist is not None:
        import torch
        mine = torch.tensor(np.stack([c_mat.real, c_mat.imag], -1), device=rk.device)
        gathered = [torch.empty_like(mine) for _ in range(world)] if rank == 0 else None
        rk.dist.gather(mine, gathered, dst=0)           # the one RCCL data collective (KB-scale)
    readout = None
    if world == 1 and not args.no_configs and args.workload == "zt_n24_chi64_D128":
        # SURVEY.md 8(d) `coefficient_batch` roofline, on the product of the timed region itself (80 GB, still in HBM)
        import bench_configs
        readout = bench_configs.coefficient_batch_entry(qil, ctx, out)
    del out
    if rank != 0:
        return
    ms_step = elapsed / args.steps * 1e3
    k_ms = kernel_ms / max(n_launch, 1)
    achieved = abytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    traffic, tsrc = pmc_traffic(args.workload)
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
                   "output_bytes_per_step": abytes, "parallelism": f"replicas x{world} (one signal per GPU)",
                   "mpo": ("genuine build_zt_mpo(24, 2 pi), natural bonds %d, zero-embedded + gauge-mixed to 128"
                           % max(t.shape[3] for t in w_natural[:-1])) if genuine else "seeded random fill",
                   "ranks_reported_by_collective_backend": world, "collective_backend": rk.backend_used,
                   "lib_sha16": lib_sha16()},
        "max_coeff_err": err_oracle if err_oracle is not None else err_lazy,
        "coeff_err": {"materialised_vs_lazy_hip": err_lazy, "materialised_vs_cpu_oracle": err_oracle,
                      "queries": args.queries, "kind": "max relative"},
        "roofline": {"bound": "hbm", "kernel": "site_apply_grouped<c64,double>",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "frac_of_measured_copy_peak": achieved / HBM_COPY_GBS,
                     "box_store_peak": box_peak, "box_store_peak_writer": box_writer,
                     "frac_of_box_store_peak": achieved / box_peak if box_peak > 0 else None,
                     "traffic": traffic, "traffic_source": tsrc, "kernel_ms": k_ms, "launches_timed": n_launch,
                     "algorithmic_bytes_per_launch": abytes},
    }
    heavy = world == 1 and (not args.no_truncate or not args.no_configs)
    if heavy:
        del W, psi
        ctx.trim()
    if world == 1 and not args.no_truncate:
        res["truncate"] = truncate_block(qil, ctx, cpu=not args.no_cpu_baseline)
        ctx.trim()
    if world == 1 and not args.no_configs:
        # one record per BASELINE.json configuration + the other two rooflines of SURVEY.md 8(d), same process (bench_configs.py)
        import bench_configs
        res["configs"] = bench_configs.configs_block(qil, ctx, small=args.workload == "tiny", readout=readout,
                                                     log=lambda m: print("[bench] " + m, file=sys.stderr, flush=True))
    if heavy:
        psi = mps_cls.alloc(cb, dtype=np.float64, amplitude=1.0, ctx=ctx).fill_random(20240064 + rank)
        W = (mpo_cls(embed_and_gauge(w_natural, db, np.random.default_rng(20240128)), ctx=ctx) if genuine
             else mpo_cls.alloc(db, dtype=np.complex128, ctx=ctx).fill_random(777))
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(W, psi, cb, db, L)
    import bench_configs as _bc
    res["summary"] = _bc.summary(res)          # LAST key: the tail of the line (what a truncated record keeps) carries every headline figure
    emit(res)


def run_sweep(args, rk):
    """configs[3]: one encoded n=24 signal x 64 damping values; a step = one whole sweep (device build of this rank's
    share, apply + 1024 samples per value, one all_gather).  Strong scaling: the 64 values are dealt to the ranks."""
    import qilaplace_jl_amd as qil
    ctx = qil.Context(rk.local_rank)
    qil.set_default_context(ctx)
    import bench_configs
    weak = args.workload == "dt_sweep_n24_weak"
    n, N, nsamp = 24, 2 ** 24, 1024
    nsig = 64 * (rk.world_hint if weak else 1)       # weak: 64 values per rank
    x = bench_configs.cfg4_signal(n)                         # :multi_sin_exp-like structured signal (Signals.jl:64-85)
    psi = qil.signal_ztmps(x, method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
    sig = np.linspace(0.25, 16.0, nsig)
    # small k, log-uniform j: the closed form x_j exp(-sigma k j / N) is NOT negligible on >= 64 % of the samples of every
    # damping value (uniformly random bits make all of them underflow to 0.0 at n = 24)
    bits, kk, jj = qil.damping_sample_bits(n, nsamp, seed=7)
    rk.attach(qil, ctx)
    forced = (rk.dist is not None or rk.comm is not None) and rk.world == 1      # QIL_BENCH_FORCE_DIST=1: the collective path in a world of one
    dist = (rk.comm or rk.dist) if (rk.world > 1 or forced) else None
    dev = rk.device if rk.dist is not None and dist is rk.dist and rk.backend == "nccl" else None
    res = None
    for _ in range(args.warmup):
        res = qil.damping_sweep(psi, sig, bits, dist=dist, device=dev, always_gather=forced)
    rk.barrier(ctx)
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = qil.damping_sweep(psi, sig, bits, dist=dist, device=dev, always_gather=forced)
    rk.barrier(ctx)
    elapsed = time.perf_counter() - t0
    ctx.profile_enable(False)
    n_launch, kernel_ms = ctx.profile_read(reset=True)
    elapsed = rk.max_over_ranks([elapsed])[0]
    if rk.rank != 0:
        return
    err, shares, peak = bench_configs.cfg4_errors(res, x, sig, kk, jj, n)    # + the share of reference samples that are not negligible
    Ws = qil.build_dt_mpo_batch(psi, [sig[0], sig[-1]])
    ab = sum(algorithmic_bytes(psi.bond_dims, W.bond_dims, w_bytes=8, a_bytes=8, o_bytes=8) for W in Ws) / 2.0
    k_ms = kernel_ms / max(n_launch, 1)
    achieved = ab / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    # what bounds the step: the per-value builder chain (one launch of dt_build_persistent per sweep), timed here with
    # host clocks around a synchronised build of this rank's share (the kernel is >= 99 % of it, the kernel stats of the sweep, profiles/r0*_kernel_stats_dt_sweep_n24_s64.csv)
    xnorm = float(np.linalg.norm(x))
    share = [sig[i] for i in range(rk.rank, nsig, rk.world)]
    tb = []
    for _ in range(3):
        ctx.synchronize()
        t0 = time.perf_counter()
        Wb = qil.build_dt_mpo_batch(psi, share)
        ctx.synchronize()
        tb.append(time.perf_counter() - t0)
        del Wb
    t_build = min(tb)
    by_values = bench_configs.builder_launch_by_values(qil, ctx, psi, (8, nsig, 4 * nsig)) if rk.world == 1 else None
    cpu = None
    if rk.world == 1 and not args.no_cpu_baseline:
        # the reference-shaped CPU path beside the figure -- build_dt_mpo (numpy restatement of
        # dt_transformer.jl:312-412) + apply + the same 1024 samples for a bounded subset of the damping values, scaled to 64
        import oracle as O
        sub = [0, nsig // 3, 2 * nsig // 3, nsig - 1]
        ph = O.SignalMPS(psi.to_host(), amplitude=psi.amplitude)
        t_b = t_a = 0.0
        worst = 0.0
        live_cpu = 1.0
        for r in sub:
            t0 = time.perf_counter()
            Wc = O.build_dt_mpo(n, float(sig[r]))
            t_b += time.perf_counter() - t0
            t0 = time.perf_counter()
            Wchain = O.SingleSiteMPO((Wc.as_single_site_mpo() if hasattr(Wc, "as_single_site_mpo") else Wc).data)
            c = O.coefficient_batch(O.apply(Wchain, ph), bits)
            t_a += time.perf_counter() - t0
            worst = max(worst, float(np.abs(c - res[r]).max() / peak))
            live_cpu = min(live_cpu, float((np.abs(c) > 1e-6 * peak).mean()))
        t_cpu = (t_b + t_a) / len(sub) * nsig
        cpu = {"value": nsig * 2 * n / t_cpu, "unit": "site-contractions/s", "cores": _host_threads(), "nproc": os.cpu_count(),
               "kind": "port",
               "sample": f"oracle.build_dt_mpo (numpy restatement of dt_transformer.jl:312-412) + oracle.apply + {nsamp} "
                         f"coefficients for {len(sub)} of the {nsig} damping values (indices {sub}), scaled to {nsig}: "
                         f"build {t_b / len(sub):.2f} s, apply + sample {t_a / len(sub):.3f} s per value",
               "seconds_per_sweep_scaled": t_cpu, "build_seconds_per_value": t_b / len(sub),
               "hip_vs_cpu_max_err_rel_to_signal_peak": worst, "cpu_samples_above_1e-6_peak_min_share": live_cpu}
    emit({
        "metric": "MPO×MPS site-contractions/sec + max |coeff err|, n=24 damping sweep (configs[3])",
        "value": nsig * 2 * n / (elapsed / args.steps), "unit": "site-contractions/s",
        "n_gpus": rk.world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak" if weak else "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": args.workload, "description": WORKLOADS[args.workload][4], "sites": 2 * n,
                   "damping_values": nsig, "samples_per_value": nsamp, "mps_bonds_max": max(psi.bond_dims),
                   "mpo_bonds_max": max(max(W.bond_dims) for W in Ws),
                   "parallelism": f"{nsig} damping values round-robin over {rk.world} rank(s), one all_gather",
                   "ranks_reported_by_collective_backend": rk.world, "collective_backend": rk.backend_used,
                   "values_per_rank": [len(range(r, nsig, rk.world)) for r in range(rk.world)],
                   "lib_sha16": lib_sha16()},
        # parity figure = HIP against the CPU restatement of the reference's algorithm on the same operands (when that leg ran);
        # the distance to the closed form is the ALGORITHM's own MPO-truncation error at the reference's default cutoff 1e-14
        # (3.5e-5 of the peak at sigma = 0.25, the numpy oracle shows the same value to five digits)
        # ONE meaning on every run -- the distance to the closed form; the parity figure has its own key
        "max_coeff_err": err,
        "max_coeff_err_kind": "vs closed form x_j exp(-sigma k j / N) / sqrt(N), relative to the signal peak, all values x samples",
        # ... and in the reference's normalisation (unit-norm input, absolute error: test/test_dt_transformer.jl:234-235, bound 1e-7)
        "max_coeff_err_unit_norm_signal": err * peak / xnorm,
        "parity_err_vs_oracle": cpu["hip_vs_cpu_max_err_rel_to_signal_peak"] if cpu else None,
        "parity_err_vs_oracle_kind": "HIP vs CPU oracle (oracle.build_dt_mpo + apply) on 4 of the damping values, same samples, relative to the signal peak (null: CPU leg not run)",
        "coeff_err": {"vs_closed_form_rel_to_signal_peak": err, "queries": nsig * nsamp,
                      "vs_closed_form_note": "the operator's own truncation at MPO cutoff 1e-14 (identical in the numpy oracle); converges with the "
                                             "cutoff: 2.8e-7 at 1e-18, 1.4e-8 at 1e-22 (tests/test_gpu_parity.py::test_config4_damping_sweep_full_size leg c)",
                      "samples": "damping_sample_bits: 1/8 k = 0, 3/8 k in 1..3 x uniform j, 1/2 k < 64 x log-uniform j",
                      "reference_samples_above_1e-6_peak": shares},
        # The step is NOT bandwidth- or matrix-bound: it is the latency of one damping value's chain of ~3 300 dependent
        # in-LDS factorisations inside dt_build_persistent.  `roofline` keeps the contract's shape for the
        # sweep's apply launches (HBM-bound, tiny); `bound_by` says what the step really waits for.
        "bound_by": {"kernel": "dt_build_persistent (one launch per sweep, one workgroup per damping value)",
                     "ms": t_build * 1e3, "frac_of_step": t_build / (elapsed / args.steps),
                     "kind": "latency chain: zip QR + gauge QR + truncating Jacobi SVD per site and layer, all in LDS",
                     "workgroups": len(share), "cu_share": len(share) / 256.0,
                     # one launch of 8 / 64 / 256 values -- the launch is as long as its slowest chain up to one value per CU
                     "builder_launch_ms_by_values": by_values},
        # strong scaling of configs[3], stated before anybody measures it: T(N) = builder launch (one chain's latency, the same for
        # 64 / N values) + the rest of this step / N.  No scaling curve has been measured (no multi-GPU node in r01-r06).
        "expected_speedup_at_8": (None if weak or rk.world != 1 else
                                  (elapsed / args.steps) / (t_build + max(elapsed / args.steps - t_build, 0.0) / 8.0)),
        "expected_speedup_model": "T(N) = t_builder_launch + (T(1) - t_builder_launch) / N, from this run's N = 1 figures" if not weak else
                                  "weak: T(N) = T(1) (64 values per rank, one builder launch per rank)",
        "roofline": {"bound": "hbm", "kernel": "site_apply_grouped<double,double> (the sweep's apply launches; the step is "
                     "dominated by the latency-bound dt_build_persistent chain, see bound_by and DESIGN.md 3.6)",
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel_ms": k_ms, "launches_timed": n_launch,
                     "algorithmic_bytes_per_launch": ab},
        "cpu_baseline": cpu,
    })


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default: 1000 applies (12 s timed) / 40 sweeps (9 s)")
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--workload", default="zt_n24_chi64_D128", choices=sorted(WORKLOADS))
    ap.add_argument("--queries", type=int, default=4096, help="coefficient samples for max|coeff err| (BASELINE.md 3.6: >= 4096)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-truncate", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (cfg2 / cfg4 / cfg5 + read-out roofline)")
    ap.add_argument("--random-mpo", action="store_true", help="seeded random MPO instead of the embedded genuine zT MPO")
    args = ap.parse_args()
    sweep = args.workload in ("dt_sweep_n24_s64", "dt_sweep_n24_weak")
    if args.steps is None:
        args.steps = 40 if sweep else 1000
    if args.warmup is None:
        args.warmup = 2 if sweep else 3

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process touches no GPU (nothing HIP- or torch-related
        # has been imported yet), starts one child per GPU and relays rank 0's JSON line
        sys.exit(spawn_ranks(args.gpus))
    rk = Ranks(args.gpus)
    (run_sweep if sweep else run_apply)(args, rk)
    rk.finish()


if __name__ == "__main__":
    main()
