#!/usr/bin/env python3
"""Pole finding with the z-transform MPO -- the large example of the reference's zT tutorial
(docs/src/tutorials/zt.md:318-560 of QILaplace.jl) on the MI355X path.

    python examples/zt_pole_scan.py

A two-pole signal x_j = a^j cos(w0 j), a = 1.00015 e^{0.002 i}, w0 = 0.0061, sampled at N = 2^20 points, is encoded
into a paired-register MPS, z-transformed (N^2 = 10^12 grid points, never formed), and the poles are located by three
scans of |chi(k, l)|: coarse (stride 2^12), fine (near the unit circle, wr = 0.5) and superfine (stride 1)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil  # noqa: E402


def z_from_kl(k, l, n, wr, wi):
    N = 2 ** n
    r, th = np.exp(-wr * k / N), wi * l / N
    return complex(r * np.cos(th), -r * np.sin(th))


def report(name, chi, ks, ls, n, wr, wi, poles):
    i, j = np.unravel_index(np.argmax(np.abs(chi)), chi.shape)
    z = z_from_kl(int(ks[i]), int(ls[j]), n, wr, wi)
    err = min(abs(z - p) for p in poles)
    print(f" Predicted pole indices from {name} scan: {int(ks[i])}, {int(ls[j])}")
    print(f" Predicted pole location from {name} scan: {z.real:.6f} + {z.imag:.6f}i")
    print(f" Error from nearest analytic pole: {err:.3e}")
    return int(ks[i]), int(ls[j]), err


def main():
    n = 20
    N = 2 ** n
    a, w0 = 1.00015 * np.exp(0.002j), 0.0061
    j = np.arange(N)
    x = a ** j * np.cos(w0 * j)
    poles = [np.exp(1j * w0) / a, np.exp(-1j * w0) / a]
    t0 = time.perf_counter()
    psi = qil.signal_ztmps(x, method="rsvd", k=50, p=5, q=2, cutoff=1e-12, maxdim=128)
    print(f"signal_ztmps: bonds_main {psi.bonds_main}, bonds_copy {psi.bonds_copy}  ({time.perf_counter() - t0:.3f} s)")
    out = {}
    # coarse: every 2^12-th k and l -- one dense block read-out
    wr = wi = 2 * np.pi
    t0 = time.perf_counter()
    phi = qil.build_zt_mpo(psi, wr, cutoff=1e-12, maxdim=128) * psi
    ks = ls = np.arange(0, N, 2 ** 12)
    chi = qil.coefficient_grid(phi, ks, ls)
    out["coarse"] = report("coarse", chi, ks, ls, n, wr, wi, poles)
    print(f"   ({chi.size} points, build + apply + scan {time.perf_counter() - t0:.3f} s)")
    # fine: 128 x 128 points with r in [1 - 1.6e-4, 1], theta in [-5e-3, 9e-3], at wr = 0.5
    wr = 0.5
    t0 = time.perf_counter()
    phi = qil.build_zt_mpo(psi, wr, cutoff=1e-12, maxdim=128) * psi
    r_t = np.linspace(1 - 1.6e-4, 1.0, 128)
    ks = np.clip(np.rint((-N / wr) * np.log(r_t)).astype(np.int64), 0, N - 1)
    th = np.mod(np.linspace(-5e-3, 9e-3, 128), 2 * np.pi)
    ls = np.mod(np.rint((N / wi) * th).astype(np.int64), N)
    chi = qil.coefficient_grid(phi, ks, ls)
    out["fine"] = report("fine", chi, ks, ls, n, wr, wi, poles)
    print(f"   ({chi.size} points, build + apply + scan {time.perf_counter() - t0:.3f} s)")
    # superfine: stride 1 around the analytic positive pole
    zt = poles[0]
    kc = int(np.clip(np.rint((-N / wr) * np.log(abs(zt))), 0, N - 1))
    lc = int(np.mod(np.rint((N / wi) * np.mod(-np.angle(zt), 2 * np.pi)), N))
    ks = np.arange(kc - 24, kc + 25)
    ls = np.mod(np.arange(lc - 24, lc + 25), N)
    t0 = time.perf_counter()
    chi = qil.coefficient_grid(phi, ks, ls)
    out["superfine"] = report("superfine", chi, ks, ls, n, wr, wi, poles)
    print(f"   ({chi.size} points, scan {time.perf_counter() - t0:.3f} s)")
    return out


if __name__ == "__main__":
    main()
