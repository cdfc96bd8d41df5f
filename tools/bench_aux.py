#!/usr/bin/env python3
"""Secondary measurements on one MI355X, one JSON line each: the reference's own published points
on this path (Apple M2 Max) re-run through the HIP path, plus the kernels
that bound encode/compress.  Not the headline bench (that is bench.py)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qilaplace_jl_amd as qil  # noqa: E402

ctx = qil.default_context()


def timed(fn, reps=5, warm=1):
    for _ in range(warm):
        fn()
    ctx.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.mean(ts)), float(np.min(ts))


def emit(**kw):
    print(json.dumps(kw), flush=True)


def sat(L, cap, base):
    return [int(min(base ** (i + 1), base ** (L - 1 - i), cap)) for i in range(L - 1)]


def apply_case(name, L, chi, D, paired, ref_s, ref_note):
    cb, db = sat(L, chi, 2), sat(L, D, 4)
    mps_cls = qil.ZTMPS if paired else qil.SignalMPS
    mpo_cls = qil.PairedSiteMPO if paired else qil.SingleSiteMPO
    psi = mps_cls.alloc(cb, dtype=np.float64).fill_random(1)
    W = mpo_cls.alloc(db, dtype=np.complex128).fill_random(2)
    holder = {}

    def run():
        holder.pop("o", None)
        holder["o"] = qil.apply(W, psi)

    mean, best = timed(run, reps=10, warm=2)
    emit(case=name, sites=L, chi_s=chi, chi_c=D, apply_ms_mean=mean * 1e3, apply_ms_min=best * 1e3,
         site_contractions_per_s=L / mean, reference_m2max_s=ref_s, reference_note=ref_note)


def main():
    which = set(sys.argv[1:]) or {"apply", "gemm", "encode", "compress", "coeff"}
    if "apply" in which:
        # BASELINE.md section 1 rows (M2 Max, ITensors CPU): same bond dimensions, synthetic tensors
        apply_case("apply_zt_n24_chi15_D89", 48, 15, 89, True, 0.161, "apply(W_zt, psi) :multi_sin_exp n=24")
        apply_case("apply_zt_n12_chi64_D91", 24, 64, 91, True, 0.929, "apply(W_zt, psi) :random n=12, 9.24 GB alloc")
        apply_case("apply_zt_n14_chi128_D91", 28, 128, 91, True, 2.80, "apply(W_zt, psi) :random n=14, 34.8 GB alloc")
        apply_case("apply_qft_n28_chi2_D8", 28, 2, 8, False, 0.756e-3, "apply(W_qft, psi) :sin n=28")
        apply_case("apply_qft_n24_chi17_D8", 24, 17, 8, False, 0.93e-3, "apply(W_qft, psi) :sin_cusp n=24")
        apply_case("apply_qft_n20_chi1024_D8", 20, 1024, 8, False, 0.628, "apply(W_qft, psi) :random n=20")
        apply_case("apply_qft_n20_chi32_D64_cfg2", 20, 32, 64, False, None, "BASELINE.json configs[1]")
    if "gemm" in which:
        # device-resident f64-MFMA GEMM rates (peak f64 matrix rate of MI355X: 78.6 TFLOP/s spec)
        for (m, n, k, dt, oa, ob, note) in (
                (4096, 4096, 4096, np.float64, "N", "N", "square"),
                (4096, 4096, 4096, np.complex128, "N", "N", "square complex"),
                (64, 16384, 8192, np.complex128, "N", "N", "coefficient_batch bulk site at cfg3 (64 queries)"),
                (4096, 55, 4096, np.float64, "T", "N", "rsvd sketch Y = M Omega, n=24 k=50"),
                (32768, 133, 32768, np.float64, "T", "N", "rsvd sketch, n=30 k=128"),
                (32768, 133, 32768, np.float64, "C", "N", "rsvd power iteration M^H Q, n=30")):
            ms = qil.gemm_device_time(m, n, k, dt, oa, ob, reps=5)
            fl = (8 if dt == np.complex128 else 2) * m * n * k
            emit(case="gemm_device", m=m, n=n, k=k, dtype=str(np.dtype(dt)), opA=oa, opB=ob, note=note, ms=ms,
                 tflops=fl / ms / 1e9, frac_of_f64_mfma_peak=fl / ms / 1e9 / 78.6)
    if "encode" in which:
        rng = np.random.default_rng(3)
        for n, kk in ((16, 50), (20, 50), (24, 50)):
            x = rng.standard_normal(2 ** n)
            mean, best = timed(lambda: qil.signal_mps(x, method="rsvd", k=kk, p=5, q=2), reps=3, warm=1)
            emit(case="signal_mps_rsvd_random", n=n, k=kk, p=5, q=2, seconds_mean=mean, seconds_min=best,
                 reference_m2max_s=0.37 if n == 24 else None)
        for n in (10, 12, 14):
            x = rng.standard_normal(2 ** n)
            mean, best = timed(lambda: qil.signal_mps(x, method="svd"), reps=2, warm=1)
            emit(case="signal_mps_svd_random", n=n, seconds_mean=mean, seconds_min=best)
        j = np.arange(2 ** 24, dtype=np.float64)
        xs = np.sin(2 * np.pi * j / 2 ** 24 * 5.0) * np.exp(-3.0 * j / 2 ** 24)
        mean, best = timed(lambda: qil.signal_ztmps(xs, method="rsvd", k=15, p=5, q=2, cutoff=1e-12), reps=3, warm=1)
        emit(case="signal_ztmps_rsvd_structured", n=24, k=15, seconds_mean=mean, seconds_min=best,
             reference_m2max_s=0.23)
    if "compress" in which:
        for chi in (16, 32, 64):
            L = 24
            psi = qil.SignalMPS.alloc(sat(L, chi, 2), dtype=np.complex128).fill_random(5)
            t0 = time.perf_counter()
            qil.compress(psi, maxdim=chi // 2, tol=1e-10)
            ctx.synchronize()
            emit(case="compress", sites=L, chi=chi, maxdim=chi // 2, seconds=time.perf_counter() - t0,
                 bonds_after=max(psi.bond_dims))
    if "coeff" in which:
        L, chi, D = 24, 32, 64
        psi = qil.SignalMPS.alloc(sat(L, chi, 2), dtype=np.float64).fill_random(1)
        W = qil.SingleSiteMPO.alloc(sat(L, D, 4), dtype=np.complex128).fill_random(2)
        out = qil.apply(W, psi)
        bits = np.random.default_rng(1).integers(0, 2, size=(1024, L))
        mean, best = timed(lambda: qil.coefficient_batch(out, bits), reps=3, warm=1)
        emit(case="coefficient_batch_materialised", sites=L, chi_out=chi * D, queries=1024, seconds=mean,
             queries_per_s=1024 / mean)
        mean, best = timed(lambda: qil.apply_coefficient_batch(W, psi, bits), reps=3, warm=1)
        emit(case="coefficient_batch_lazy", sites=L, queries=1024, seconds=mean, queries_per_s=1024 / mean)


if __name__ == "__main__":
    main()
