"""bench_configs.py -- the `configs` block of bench.py's default line: one driver-run record for every BASELINE.json
configuration and for the two other rooflines of SURVEY.md 8(d).

  cfg2               n=20 chi_s=32 chi_c=64 QFT apply: ms, algorithmic GB/s against the HBM spec, all 2^20 coefficients vs numpy FFT
  cfg4               n=24 x 64 damping values: ms per sweep, what bounds it, error against the closed form on samples where the
                     closed form is NOT negligible (share printed)
  cfg5               n=30 signal generated in HBM: signal_ztmps(:rsvd, k=128, p=5, q=2) encode with the bytes / flops model of
                     its root split against the HBM and f64-MFMA peaks, zT MPO build, lazy read-out, materialised apply of the
                     structured signal and of the saturated chi_s = 128 state (62 GB, HBM-resident), error against the closed form
  coefficient_batch  64 coefficients of the 80 GB cfg3 product: ms, site bytes against the HBM spec and slice flops against the
                     f64 matrix peak (measured by bench.py on the product of its own timed region, passed in)

Reference points (M2 Max, another machine; context, not a target): docs/src/benchmarking.md:162-166 (`signal_mps` random 2^24:
:svd 19.67 s, :rsvd 0.37 s), :307 (`signal_ztmps(:rsvd, k=15)` n=30 19.4-20.0 s), :309 (`apply(W_zt, psi)` chi_s=64 0.929 s).

Every device time is HIP events on the library's stream (`qil_timer_*` around a call, `qil_profile_*` around the apply kernel).
Nothing here touches oracle/: the checks are closed forms (numpy FFT, geometric series).
"""
import time

import numpy as np

HBM_PEAK_GBS = 8000.0
F64_MFMA_PEAK_TFLOPS = 78.6

# keys every entry of the block carries (tests/test_bench_contract.py; the block asserts them itself before returning)
CONFIGS_BLOCK_KEYS = {
    "cfg2": ["workload", "ms_per_apply", "kernel_ms", "site_contractions_per_s", "algorithmic_bytes", "roofline", "max_coeff_err"],
    "cfg4": ["workload", "ms_per_sweep", "site_contractions_per_s", "bound_by", "max_coeff_err", "reference_samples_above_1e-6_peak"],
    "cfg5": ["workload", "encode_ms", "encode_roofline", "zt_build_ms", "lazy_readout_ms", "max_coeff_err"],
    "coefficient_batch": ["workload", "queries", "ms", "roofline"],
    "zt_build": ["workload", "ms_single", "ms_batch64", "max_bond", "stages_ms"],
}


def pmc_record():
    """The newest profiles/r0*_pmc_truncate.json (tools/collect_pmc_truncate.py: counted f64 MFMA instructions per repetition of the
    truncate-half workloads, the 64-query read-out and the n = 30 encode) IF it was collected with the library binary that is
    running now (sha256 beside it); else (None, reason)."""
    import glob, hashlib, json, os
    root = os.path.dirname(os.path.abspath(__file__))
    paths = sorted(glob.glob(os.path.join(root, "profiles", "r0*_pmc_truncate.json")))
    if not paths:
        return None, None
    try:
        import qilaplace_jl_amd as qil
        sha = hashlib.sha256(open(qil.LIB_PATH, "rb").read()).hexdigest()[:16]
        rec = json.load(open(paths[-1]))
        if rec.get("lib_sha16") != sha:
            return None, f"{os.path.basename(paths[-1])} was collected with another build of libqilhip.so"
        return rec, os.path.basename(paths[-1])
    except Exception:                                        # noqa: BLE001
        return None, None


def pmc_traffic(key, algorithmic_bytes=None):
    """{"traffic", "traffic_over_algorithmic", "traffic_source"}: HBM bytes of workload `key` from the newest
    profiles/r0*_pmc_traffic.json (tools/collect_pmc.py: WRITE_SIZE + 2 x FETCH_SIZE, separate --pmc passes) IF it was collected with
    the library binary that is running now; traffic = None otherwise, with the reason in traffic_source."""
    import glob, hashlib, json, os
    root = os.path.dirname(os.path.abspath(__file__))
    paths = sorted(glob.glob(os.path.join(root, "profiles", "r0*_pmc_traffic.json")))
    if not paths:
        return {"traffic": None, "traffic_over_algorithmic": None, "traffic_source": None}
    try:
        import qilaplace_jl_amd as qil
        sha = hashlib.sha256(open(qil.LIB_PATH, "rb").read()).hexdigest()[:16]
        rec = json.load(open(paths[-1]))
        name = os.path.basename(paths[-1])
        if rec.get("lib_sha16") != sha:
            return {"traffic": None, "traffic_over_algorithmic": None, "traffic_source": f"{name} was collected with another build of libqilhip.so"}
        t = rec.get(key)
        return {"traffic": t, "traffic_over_algorithmic": (t / algorithmic_bytes) if (t and algorithmic_bytes) else None,
                "traffic_source": name}
    except Exception:                                        # noqa: BLE001
        return {"traffic": None, "traffic_over_algorithmic": None, "traffic_source": None}


def counted_mfma(key, ms):
    """{"mfma_f64_flops", "achieved", "frac", "source"} from the PMC record of this build for workload `key`, over the time measured here."""
    rec, src = pmc_record()
    if not rec or key not in rec:
        return {"mfma_f64_flops": None, "achieved": None, "frac": None, "source": src}
    fl = rec[key]["mfma_f64_flops"]
    ach = fl / (ms * 1e-3) / 1e12
    return {"mfma_f64_flops": fl, "achieved": ach, "unit": "TFLOP/s", "frac": ach / F64_MFMA_PEAK_TFLOPS, "source": src,
            "dispatches": rec[key].get("dispatches"),
            "mfma_busy_share_of_simd_cycles": rec[key]["mfma_busy_cycles"] / (ms * 1e-3 * 2.4e9 * 1024.0),
            "model": "counted: SQ_INSTS_VALU_MFMA_MOPS_F64 x 512 per repetition (3-repetition minus 1-repetition rocprofv3 --pmc run of "
                     "this build) over the time measured here"}


def saturated(L, cap, base=2):
    return [int(min(base ** (i + 1), base ** (L - 1 - i), cap)) for i in range(L - 1)]


def timed(ctx, fn, reps=3, warm=1):
    """(mean, min) device ms of fn() over `reps` runs after `warm` dry runs: HIP events on the library's stream around the call."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        ctx.synchronize()
        ctx.timer_start()
        fn()
        ts.append(ctx.timer_stop())
    return sum(ts) / len(ts), min(ts)


# ---------------------------------------------------------------------------------------------- coefficient_batch roofline
def readout_roofline(bond_dims, nb, ms, elem_bytes=16, complex_sites=True):
    """SURVEY.md 8(d) `coefficient_batch`: the batch reads every site tensor once (sum of elem * chi_l * 2 * chi_r bytes) and a
    query needs the product of its vector with ONE slice per site: 8 (c64) or 2 (f64) flop * chi_l * chi_r per site and query.
    The GEMM form executes both slices for every query (twice the algorithmic flops) unless the batch is bit-sorted."""
    c = [1] + list(bond_dims) + [1]
    bytes_ = sum(elem_bytes * c[i] * 2 * c[i + 1] for i in range(len(c) - 1))
    # complex products are executed as Gauss's THREE real multiplications (csrc/qil_linalg.hip): 6 flop issue per complex
    # multiply-add; the fraction of the matrix peak is quoted on those (the conventional 8 overstates the pipe's
    # utilisation by 4/3 and can exceed 1); the conventional-equivalent rate stays as a labelled extra
    per_mac = 6.0 if complex_sites else 2.0
    macs = nb * sum(c[i] * c[i + 1] for i in range(len(c) - 1))
    flops = per_mac * macs
    t = ms * 1e-3
    hbm, mfma = bytes_ / t / 1e9, flops / t / 1e12
    f_h, f_m = hbm / HBM_PEAK_GBS, mfma / F64_MFMA_PEAK_TFLOPS
    bound = "mfma" if f_m >= f_h else "hbm"
    return {"bound": bound, "achieved": mfma if bound == "mfma" else hbm, "peak": F64_MFMA_PEAK_TFLOPS if bound == "mfma" else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": max(f_h, f_m),
            "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": f_h, "algorithmic_bytes": bytes_},
            "mfma": {"achieved": mfma, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": f_m, "algorithmic_flops": flops,
                     "conventional_equivalent_tflops": (8.0 if complex_sites else 2.0) * macs / t / 1e12},
            "model": "bytes = every site tensor read once per batch; flops = one slice per site and query, 6 flop per complex "
                     "multiply-add (three real multiplications: what the matrix pipe executes; `conventional_equivalent_tflops` counts 8); "
                     "time = HIP events around the whole read-out (bit-sorted: two one-slice GEMMs + one row gather per site)",
            "traffic": None, "traffic_over_algorithmic": None, "traffic_source": None}


def coefficient_batch_entry(qil, ctx, out, nb=64, reps=3):
    """64 coefficients of a materialised product `out` (bench.py passes the 80 GB cfg3 product of its timed region)."""
    L = out.ntensors if hasattr(out, "ntensors") else len(out)
    bits = np.random.default_rng(64).integers(0, 2, size=(nb, L)).astype(np.uint8)
    mean, best = timed(ctx, lambda: qil.coefficient_batch(out, bits), reps=reps)
    cx = np.dtype(out.dtype) == np.complex128
    roof = readout_roofline(out.bond_dims, nb, mean, 16 if cx else 8, cx)
    if nb == 64 and max(out.bond_dims) == 8192:
        roof["mfma_counted"] = counted_mfma("coefficient_batch_64_cfg3", mean)      # executed (incl. tile padding), not algorithmic
        roof.update(pmc_traffic("coefficient_batch_64_cfg3", roof["hbm"]["algorithmic_bytes"]))
    return {"workload": "coefficient_batch on the materialised cfg3 product (zt_n24_chi64_D128, 80 GB)", "queries": nb,
            "ms": mean, "ms_min": best, "repetitions": reps, "product_bond_max": int(max(out.bond_dims)),
            "roofline": roof}


# ---------------------------------------------------------------------------------------------- cfg2
def embed_and_gauge(Wdata, cap_profile, rng):
    """zero-embed every bond of the genuine MPO to the nominal cap and conjugate it with a seeded random
    orthogonal gauge (G on one side, G^T on the other): the operator is exactly unchanged, every tensor is dense."""
    out = [np.asarray(w, dtype=np.complex128) for w in Wdata]
    for i in range(len(out) - 1):
        d, D = out[i].shape[3], cap_profile[i]
        assert D >= d, (i, d, D)
        G, _ = np.linalg.qr(rng.standard_normal((D, D)))
        left = np.zeros(out[i].shape[:3] + (D,), dtype=np.complex128)
        left[..., :d] = out[i]
        right = np.zeros((D,) + out[i + 1].shape[1:], dtype=np.complex128)
        right[:d] = out[i + 1]
        out[i] = left @ G
        out[i + 1] = np.tensordot(G.T, right, axes=([1], [0]))
    return out


def algorithmic_bytes(cb, db, w_bytes=16, a_bytes=8, o_bytes=16):
    """per site  out * (Dl chil) * 2 * (Dr chir) [write B once] + W + A read once."""
    c = [1] + list(cb) + [1]
    d = [1] + list(db) + [1]
    return sum(o_bytes * (d[i] * c[i]) * 2 * (d[i + 1] * c[i + 1]) + w_bytes * d[i] * 4 * d[i + 1] + a_bytes * c[i] * 2 * c[i + 1]
               for i in range(len(c) - 1))


def cfg2_entry(qil, ctx, n=20, chi=32, D=64, steps=200):
    """configs[1]: synthetic saturated MPS (seeded device fill), the GENUINE QFT MPO (device builder at cutoff 1e-24 so its own
    truncation sits below the check) zero-embedded + gauge-mixed to the dense chi_c profile; all 2^n coefficients against numpy FFT."""
    cb, db = saturated(n, chi), saturated(n, D, base=4)
    psi = qil.SignalMPS.alloc(cb, dtype=np.float64, amplitude=1.0, ctx=ctx).fill_random(20240032)
    w_nat = qil.build_qft_mpo(n, cutoff=1e-24, ctx=ctx).to_host()
    W = qil.SingleSiteMPO(embed_and_gauge(w_nat, db, np.random.default_rng(20240032)), ctx=ctx)
    out = None
    for _ in range(5):
        del out
        out = qil.apply(W, psi)
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        del out
        out = qil.apply(W, psi)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / steps
    ctx.profile_enable(False)
    nl, kms = ctx.profile_read(reset=True)
    k_ms = kms / max(nl, 1)
    ab = algorithmic_bytes(cb, db)
    x = qil.mps_to_vector(psi)
    F = np.fft.fft(x) / np.sqrt(2 ** n)
    full = qil.mps_to_vector(out, reverse=True)
    err = float(np.abs(full - F).max() / np.abs(F).max())
    ach = ab / (k_ms * 1e-3) / 1e9
    return {"workload": f"qft_n{n}_chi{chi}_D{D}", "sites": n, "ms_per_apply": wall * 1e3, "kernel_ms": k_ms, "steps": steps,
            "site_contractions_per_s": n / wall, "algorithmic_bytes": ab, "mpo_natural_bond_max": int(max(t.shape[3] for t in w_nat[:-1])),
            "roofline": {"bound": "hbm", "kernel": "site_apply_grouped<c64,double>", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "frac_wall": ab / wall / 1e9 / HBM_PEAK_GBS,
                         **(pmc_traffic(f"qft_n{n}_chi{chi}_D{D}", ab) if (n, chi, D) == (20, 32, 64) else
                            {"traffic": None, "traffic_over_algorithmic": None, "traffic_source": "reduced size: not the profiled workload"}),
                         "note": "a 1.5 GB apply of 0.2-0.3 ms: the launch's fill and drain are a visible share of it"},
            "max_coeff_err": err, "coeff_err_kind": f"all 2^{n} coefficients vs numpy.fft.fft(x) / sqrt(N), relative to max |F|"}


# ---------------------------------------------------------------------------------------------- cfg4
def cfg4_signal(n):
    """:multi_sin_exp-like structured signal (Signals.jl:64-85), the one tests/test_gpu_parity.py::test_config4_* uses."""
    N = 2 ** n
    j = np.arange(N, dtype=np.float64)
    rng = np.random.default_rng(1001)
    ak = rng.random(10)
    ak /= np.linalg.norm(ak)
    wk = 40.0 / N * (rng.random(10) - 0.5)
    lk = -2.0 / N * rng.random(10)
    return sum(ak[k] * np.sin(wk[k] * j) * np.exp(lk[k] * j) for k in range(10))


def cfg4_errors(res, x, sig, kk, jj, n):
    """max |HIP - closed form| / signal peak over all damping values, and how much of the reference is not negligible."""
    N = 2 ** n
    peak = np.abs(x).max() / np.sqrt(N)
    refs = np.stack([x[jj] * np.exp(-s * kk * jj / N) / np.sqrt(N) for s in sig])
    err = float((np.abs(res - refs).max(axis=1) / peak).max())
    live = (np.abs(refs) > 1e-6 * peak).mean(axis=1)
    big = (np.abs(refs) > 1e-2 * peak).mean(axis=1)
    shares = {"min_share_over_values": float(live.min()), "mean_share": float(live.mean()), "count": int((np.abs(refs) > 1e-6 * peak).sum()),
              "above_1e-2_peak_min_share": float(big.min()), "above_1e-2_peak_mean_share": float(big.mean())}
    return err, shares, peak


def builder_launch_by_values(qil, ctx, psi, counts, lo=0.25, hi=16.0):
    """Wall ms (synchronised, min of 2) of ONE build_dt_mpo_batch of `c` damping values in linspace(lo, hi, c), per count."""
    out = {}
    for c in counts:
        sig = np.linspace(lo, hi, int(c))
        best = None
        for _ in range(2):
            ctx.synchronize()
            t0 = time.perf_counter()
            Wb = qil.build_dt_mpo_batch(psi, sig)
            ctx.synchronize()
            dt_ = time.perf_counter() - t0
            del Wb
            best = dt_ if best is None else min(best, dt_)
        out[str(int(c))] = best * 1e3
    return out


def cfg4_entry(qil, ctx, n=24, nsig=64, nsamp=1024, steps=3):
    x = cfg4_signal(n)
    psi = qil.signal_ztmps(x, method="rsvd", k=15, p=5, q=2, cutoff=1e-12)
    sig = np.linspace(0.25, 16.0, nsig)
    bits, kk, jj = qil.damping_sample_bits(n, nsamp, seed=7)
    res = qil.damping_sweep(psi, sig, bits)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        res = qil.damping_sweep(psi, sig, bits)
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / steps
    tb = []
    for _ in range(2):
        ctx.synchronize()
        t0 = time.perf_counter()
        Wb = qil.build_dt_mpo_batch(psi, sig)
        ctx.synchronize()
        tb.append(time.perf_counter() - t0)
        mpo_bond = int(max(max(W.bond_dims) for W in Wb))
        del Wb
    err, shares, _ = cfg4_errors(res, x, sig, kk, jj, n)
    by_values = builder_launch_by_values(qil, ctx, psi, (8, nsig, 4 * nsig))
    return {"workload": f"dt_sweep_n{n}_s{nsig}", "damping_values": nsig, "samples_per_value": nsamp, "steps": steps,
            "ms_per_sweep": wall * 1e3, "site_contractions_per_s": nsig * 2 * n / wall, "mps_bonds_max": int(max(psi.bond_dims)),
            "mpo_bonds_max": mpo_bond,
            "bound_by": {"kernel": "DT builder launch (latency chain of in-LDS factorisations per damping value)",
                         "ms": min(tb) * 1e3, "frac_of_step": min(tb) / wall,
                         "builder_launch_ms_by_values": by_values,
                         "note": "one workgroup per damping value: the launch is as long as its slowest chain whatever the count up to one "
                                 "value per CU (256), so 4x the values per launch cost the same time -- the unit that scales with GPUs is the value"},
            "max_coeff_err": err, "coeff_err_kind": "vs the closed form x_j exp(-sigma k j / N) / sqrt(N), relative to the signal peak, all values x samples: "
                                                    "the reference algorithm's own MPO-truncation error at cutoff 1e-14 (the numpy oracle shows the same value; "
                                                    "HIP vs oracle 3e-12 in the sweep workload's line and in test_config4_* leg a)",
            "reference_samples_above_1e-6_peak": shares}


# ---------------------------------------------------------------------------------------------- cfg5
def zt_closed_form(terms, n, wr, kk, ll):
    """chi(k, l) = (1/N) sum_j x_j exp(-(wr k + 2 pi i l) j / N) (test/test_zt_transformer.jl:20-39) for x_j = sum_m c_m exp(lam_m j / N):
    geometric series, sum_{j<N} exp(z j / N) = expm1(z) / expm1(z / N)."""
    N = 2.0 ** n
    out = np.zeros(len(kk), dtype=np.complex128)
    for c, lam in terms:
        z = lam - wr * np.asarray(kk, dtype=np.float64) - 2j * np.pi * np.asarray(ll, dtype=np.float64)
        den = np.expm1(z / N)
        out += c * np.where(den == 0, N, np.expm1(z) / np.where(den == 0, 1.0, den))
    return out / N


STRUCTURED_TERMS = [(0.5 / 1j, -3.0 + 2j * np.pi * 5.0), (-0.5 / 1j, -3.0 - 2j * np.pi * 5.0),
                    (0.25, 2j * np.pi * 11.0), (0.25, -2j * np.pi * 11.0)]      # sin(2 pi 5 t) e^{-3t} + 0.5 cos(2 pi 11 t)


def kl_bits(n, kk, ll):
    bits = np.zeros((len(kk), 2 * n), dtype=np.uint8)
    for i in range(n):
        bits[:, 2 * i] = (np.asarray(kk) >> i) & 1
        bits[:, 2 * i + 1] = (np.asarray(ll) >> i) & 1
    return bits


def rsvd_root_model(n, k, p, q):
    """SURVEY.md 8(d) RSVD encode: the root split of the bisection is the 2^(n/2) x 2^(n - n/2) matricisation of the whole signal;
    (2 + 2q) products with l = k + p column panels: flops 2 m n l each, bytes 8 m n each (the panels are l / n of that).  The
    children hold <= (k + p) 2^(n/2) elements each (a 2^-(n/2) share of the root), so the root IS the encode's bytes and flops."""
    m, nn, l = 2 ** (n // 2), 2 ** (n - n // 2), k + p
    return (2 + 2 * q) * 2.0 * m * nn * l, (2 + 2 * q) * 8.0 * m * nn


def cfg5_entry(qil, ctx, n=30, k=128, p=5, q=2, reps=2):
    """configs[4]: the 2^n samples are produced IN HBM (torch) and never exist on the host.  Two signals: the structured one
    (closed form available: accuracy; its encoded bonds are small, so the materialised zT apply fits) and an i.i.d. normal one
    (every bond saturates at chi_s = 128: the encode's worst case, the one the roofline is quoted on)."""
    import torch
    N = 2 ** n
    dev = torch.device("cuda", ctx.device)
    jd = torch.arange(N, dtype=torch.float64, device=dev)
    xd = torch.sin(2 * np.pi * 5.0 * jd / N) * torch.exp(-3.0 * jd / N) + 0.5 * torch.cos(2 * np.pi * 11.0 * jd / N)
    del jd
    torch.cuda.synchronize()
    enc = lambda sig_dev: qil.signal_ztmps(sig_dev, method="rsvd", k=k, p=p, q=q, cutoff=1e-12, maxdim=k)
    box = {}

    def run_s():
        box["psi"] = enc(xd)

    e_mean, e_min = timed(ctx, run_s, reps=reps)
    psi = box.pop("psi")
    del xd
    torch.cuda.empty_cache()
    wr = 2 * np.pi
    ctx.synchronize()
    t0 = time.perf_counter()
    W = qil.build_zt_mpo_batch(psi, [wr], cutoff=1e-14)[0]
    ctx.synchronize()
    t_build1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    W = qil.build_zt_mpo_batch(psi, [wr], cutoff=1e-14)[0]
    ctx.synchronize()
    t_build = time.perf_counter() - t0
    rng = np.random.default_rng(5)
    nq = 64
    kk, ll = rng.integers(0, min(64, N), size=nq), rng.integers(0, min(32, N), size=nq)
    bits = kl_bits(n, kk, ll)
    box = {}

    def run_l():
        box["c"] = qil.apply_coefficient_batch(W, psi, bits)

    l_mean, l_min = timed(ctx, run_l, reps=reps)
    lazy = box["c"]
    ref = zt_closed_form(STRUCTURED_TERMS, n, wr, kk, ll)
    err_abs = float(np.abs(lazy - ref).max())
    pb = [c * d for c, d in zip(psi.bond_dims, W.bond_dims)]
    out_bytes = sum(16 * a * 2 * b for a, b in zip([1] + pb, pb + [1]))
    res = {"workload": f"zt_n{n}_rsvd_k{k}", "signal_samples": N, "signal_bytes": 8 * N, "signal": "generated in HBM (torch), never on the host",
           "encode_ms": e_mean, "encode_ms_min": e_min, "encode_signal": "structured: sin(2 pi 5 t) e^{-3t} + 0.5 cos(2 pi 11 t)",
           "mps_bonds_max": int(max(psi.bond_dims)), "mpo_bonds_max": int(max(W.bond_dims)),
           "zt_build_ms": t_build * 1e3, "zt_build_first_call_ms": t_build1 * 1e3,
           "lazy_readout_ms": l_mean, "lazy_readout_queries": nq,
           "max_coeff_err": err_abs, "max_coeff_err_rel": float(np.abs(lazy - ref).max() / np.abs(ref).max()),
           "coeff_err_kind": "lazy <k,l| W_zt psi> vs the closed-form z-transform, absolute (the reference's zT bound: 2e-7, "
                             "test/test_zt_transformer.jl:106) and relative to the largest sampled |chi|",
           "materialised_output_bytes": out_bytes}
    if out_bytes < 100e9:
        out = None

        def run_a():
            box.pop("o", None)
            box["o"] = qil.apply(W, psi)

        a_mean, a_min = timed(ctx, run_a, reps=3)
        out = box.pop("o")
        mat = qil.coefficient_batch(out, bits)
        res.update({"apply_ms": a_mean, "apply_site_contractions_per_s": 2 * n / (a_mean * 1e-3),
                    "apply_GBps": out_bytes / (a_mean * 1e-3) / 1e9,
                    "lazy_vs_materialised_rel": float(np.abs(mat - lazy).max() / np.abs(mat).max())})
        del out
    del W, psi
    ctx.trim()
    # the saturated case: i.i.d. normal samples, every bond of the bulk = chi_s
    g = torch.Generator(device=dev)
    g.manual_seed(30)
    xr = torch.randn(N, dtype=torch.float64, device=dev, generator=g)
    torch.cuda.synchronize()

    def run_r():
        box["psi"] = enc(xr)

    r_mean, r_min = timed(ctx, run_r, reps=reps)
    psi_r = box.pop("psi")
    flops, bytes_ = rsvd_root_model(n, k, p, q)
    t = r_mean * 1e-3
    hbm, mf = bytes_ / t / 1e9, flops / t / 1e12
    res["encode_random_ms"] = r_mean
    res["encode_random_ms_min"] = r_min
    res["encode_random_bonds_max"] = int(max(psi_r.bond_dims))
    res["encode_roofline"] = {
        "signal": "i.i.d. normal (bonds saturate at chi_s)", "bound": "mfma" if mf / F64_MFMA_PEAK_TFLOPS >= hbm / HBM_PEAK_GBS else "hbm",
        "hbm": {"achieved": hbm, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm / HBM_PEAK_GBS, "algorithmic_bytes": bytes_},
        "mfma": {"achieved": mf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": mf / F64_MFMA_PEAK_TFLOPS, "algorithmic_flops": flops},
        "frac": max(mf / F64_MFMA_PEAK_TFLOPS, hbm / HBM_PEAK_GBS),
        "mfma_counted": counted_mfma("encode_n30_random_k128", r_mean) if (n, k, p, q) == (30, 128, 5, 2) else None,
        **(pmc_traffic("encode_n30_random_k128", bytes_) if (n, k, p, q) == (30, 128, 5, 2) else
           {"traffic": None, "traffic_over_algorithmic": None, "traffic_source": "reduced size: not the profiled workload"}),
        "model": f"root split {2 ** (n // 2)} x {2 ** (n - n // 2)}: (2 + 2q) = {2 + 2 * q} sketch products with l = k + p = {k + p} columns, "
                 "2 m n l flops and 8 m n bytes each; time = the WHOLE encode (root + 2 n - 2 smaller splits + normalisation), HIP events"}
    del xr
    torch.cuda.empty_cache()
    ctx.trim()
    # ... and the apply at the configuration's NOMINAL chi_s: zT MPO (natural bonds) x the saturated state, materialised in HBM
    # (201 GB at D ~ 90; it must not be padded to 128 -- 406 GB > 288 GB).  Only when the device has the room.
    try:
        Wr = qil.build_zt_mpo_batch(psi_r, [wr], cutoff=1e-14)[0]
        pbr = [c * d for c, d in zip(psi_r.bond_dims, Wr.bond_dims)]
        ob = sum(16 * a * 2 * b for a, b in zip([1] + pbr, pbr + [1]))
        free = ctx.mem_info()["device_free"]
        sat = {"output_bytes": ob, "device_free_bytes": int(free), "mps_bonds_max": int(max(psi_r.bond_dims)), "mpo_bonds_max": int(max(Wr.bond_dims)),
               "product_bond_max": int(max(pbr))}
        if ob < 0.8 * free:
            ab = algorithmic_bytes(psi_r.bond_dims, Wr.bond_dims)
            outr = qil.apply(Wr, psi_r)                                   # warm-up: the pool takes the blocks from the driver
            ctx.synchronize()
            ctx.profile_enable(True)
            ctx.profile_read(reset=True)
            t0 = time.perf_counter()
            for _ in range(2):
                del outr
                outr = qil.apply(Wr, psi_r)
            ctx.synchronize()
            wall = (time.perf_counter() - t0) / 2
            ctx.profile_enable(False)
            nl, kms = ctx.profile_read(reset=True)
            k_ms = kms / max(nl, 1)
            bq = kl_bits(n, rng.integers(0, min(64, N), size=16), rng.integers(0, min(1 << 20, N), size=16))
            mat = qil.coefficient_batch(outr, bq)
            lz = qil.apply_coefficient_batch(Wr, psi_r, bq)
            sat.update({"apply_ms": wall * 1e3, "kernel_ms": k_ms, "site_contractions_per_s": 2 * n / wall, "algorithmic_bytes": ab,
                        "roofline": {"bound": "hbm", "kernel": "site_apply_grouped<c64,double>", "achieved": ab / (k_ms * 1e-3) / 1e9,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                        "materialised_vs_lazy_rel": float(np.abs(mat - lz).max() / np.abs(lz).max())})
            del outr
        else:
            sat["skipped"] = "the materialised product does not fit beside what the device holds"
        res["apply_saturated"] = sat
        del Wr
    except Exception as e:                                                 # noqa: BLE001
        res["apply_saturated"] = {"error": f"{type(e).__name__}: {e}"}
    del psi_r
    ctx.trim()
    return res


# ---------------------------------------------------------------------------------------------- build_zt_mpo on the device
def zt_build_entry(qil, ctx, n=24, nvalues=64, reps=3):
    """build_zt_mpo(n, 2 pi) and a sweep of `nvalues` damping values through ONE C verb each (qil_build_zt_mpo_batch: DT halves ||
    paired QFT chain on two streams, product, batched compression; zt_transformer.jl:41-112), wall clock, synchronised, best of
    `reps` after a dry run; beside them the stages called one after another from their own entries."""
    wr = 2 * np.pi

    def best(fn, r):
        out = []
        for _ in range(r + 1):
            ctx.synchronize()
            t0 = time.perf_counter()
            keep = fn()
            ctx.synchronize()
            out.append(time.perf_counter() - t0)
            del keep
        return min(out[1:]), out[0]

    t1, t1_first = best(lambda: qil.build_zt_mpo(n, wr, ctx=ctx), reps)
    W = qil.build_zt_mpo(n, wr, ctx=ctx)
    bond = int(max(W.bond_dims))
    del W
    sig = np.linspace(0.25, 16.0, nvalues)
    tb, _ = best(lambda: qil.build_zt_mpo_batch(n, sig, ctx=ctx), max(1, reps - 1))
    st = {}
    ctx.synchronize(); t0 = time.perf_counter(); dts = qil.build_dt_mpo_batch(n, [wr], ctx=ctx); ctx.synchronize(); st["dt_half"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); Q = qil.zt_qft_chain_device(n, dts[0].site_ids, ctx=ctx); ctx.synchronize(); st["paired_qft_chain"] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter(); P = qil.apply(dts[0], Q); ctx.synchronize(); st["product"] = (time.perf_counter() - t0) * 1e3
    b0 = int(max(P.bond_dims))
    t0 = time.perf_counter(); qil.mpo_compress(P, "down", 1e-14, 1000); ctx.synchronize(); st["compress"] = (time.perf_counter() - t0) * 1e3
    return {"workload": f"build_zt_mpo(n={n}, 2 pi) / sweep of {nvalues} damping values in linspace(0.25, 16), cutoff 1e-14",
            "ms_single": t1 * 1e3, "ms_single_first_call": t1_first * 1e3, "ms_batch64": tb * 1e3, "values": int(nvalues),
            "max_bond": bond, "product_bond_before_compression": b0, "stages_ms": st,
            "note": "single = max(DT half, paired QFT chain) + product + compression: the two persistent builders run concurrently on two "
                    "streams; nothing but 2 x 2 gate entries is computed on the host"}


# ---------------------------------------------------------------------------------------------- the block
def configs_block(qil, ctx, small=False, readout=None, log=None):
    """All entries; `small` runs the same code at sizes a test can afford (n = 12 / 10 / 16).  `readout`: the coefficient_batch entry
    measured by the caller on its own materialised product.  A failing entry is reported as {"error": ...}: the headline line must
    still print."""
    todo = {
        "cfg2": (lambda: cfg2_entry(qil, ctx, n=12, chi=16, D=32, steps=20)) if small else (lambda: cfg2_entry(qil, ctx)),
        "cfg4": (lambda: cfg4_entry(qil, ctx, n=10, nsig=8, nsamp=256, steps=1)) if small else (lambda: cfg4_entry(qil, ctx)),
        "cfg5": (lambda: cfg5_entry(qil, ctx, n=16, k=24, reps=1)) if small else (lambda: cfg5_entry(qil, ctx)),
        "zt_build": (lambda: zt_build_entry(qil, ctx, n=8, nvalues=6, reps=1)) if small else (lambda: zt_build_entry(qil, ctx)),
    }
    block = {}
    for name, fn in todo.items():
        t0 = time.perf_counter()
        try:
            block[name] = fn()
            missing = [k for k in CONFIGS_BLOCK_KEYS[name] if k not in block[name]]
            assert not missing, f"{name}: keys missing from the entry: {missing}"
        except AssertionError:
            raise
        except Exception as e:                               # noqa: BLE001
            block[name] = {"error": f"{type(e).__name__}: {e}"}
        block[name]["seconds_spent"] = time.perf_counter() - t0
        ctx.trim()
        if log:
            log(f"configs[{name}] done in {block[name]['seconds_spent']:.1f} s")
    if readout is not None:
        block["coefficient_batch"] = readout
    return block


def summary(res):
    """The figures a reader of the LAST 2 000 characters of the line needs (the driver's record keeps a tail of
    stdout): one compact object, appended as the line's last key."""
    def g(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return float(f"{d:.5g}") if isinstance(d, float) else d
    c = res.get("configs") or {}
    t = res.get("truncate") or {}
    return {
        "apply_ms": g(res, "ms_per_step"), "apply_frac_of_8TBps": g(res, "roofline", "frac"),
        "apply_frac_of_box_store_peak": g(res, "roofline", "frac_of_box_store_peak"), "box_store_peak_GBps": g(res, "roofline", "box_store_peak"),
        "apply_traffic_over_algorithmic": (g(res, "roofline", "traffic") / g(res, "roofline", "algorithmic_bytes_per_launch")
                                            if g(res, "roofline", "traffic") else None),
        "cpu_sites_per_s": g(res, "cpu_baseline", "value"), "max_coeff_err": g(res, "max_coeff_err"),
        "cfg2": {"ms": g(c, "cfg2", "ms_per_apply"), "frac": g(c, "cfg2", "roofline", "frac"), "traffic_ratio": g(c, "cfg2", "roofline", "traffic_over_algorithmic"),
                 "err": g(c, "cfg2", "max_coeff_err")},
        "cfg4": {"ms_per_sweep": g(c, "cfg4", "ms_per_sweep"), "builder_ms": g(c, "cfg4", "bound_by", "ms"), "err": g(c, "cfg4", "max_coeff_err")},
        "cfg5": {"encode_ms": g(c, "cfg5", "encode_ms"), "encode_random_ms": g(c, "cfg5", "encode_random_ms"),
                 "encode_frac": g(c, "cfg5", "encode_roofline", "frac"), "encode_traffic_ratio": g(c, "cfg5", "encode_roofline", "traffic_over_algorithmic"),
                 "zt_build_ms": g(c, "cfg5", "zt_build_ms"), "lazy_readout_ms": g(c, "cfg5", "lazy_readout_ms"),
                 "apply_saturated_frac": g(c, "cfg5", "apply_saturated", "roofline", "frac"), "err": g(c, "cfg5", "max_coeff_err")},
        "readout64": {"ms": g(c, "coefficient_batch", "ms"), "frac": g(c, "coefficient_batch", "roofline", "frac"),
                      "traffic_ratio": g(c, "coefficient_batch", "roofline", "traffic_over_algorithmic")},
        "zt_build_n24": {"ms_single": g(c, "zt_build", "ms_single"), "ms_batch64": g(c, "zt_build", "ms_batch64"), "stages_ms": g(c, "zt_build", "stages_ms")},
        "truncate": {"fused_ms": g(t, "fused_apply_compress_ms"), "exact_ms": g(t, "exact_compress_ms"),
                     "compress_chi256_ms": g(t, "compress_chi256_to_128_24_sites_ms"), "batch64_pairs_per_s": g(t, "batch64", "pairs_per_s")},
    }
