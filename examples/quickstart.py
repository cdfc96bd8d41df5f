#!/usr/bin/env python3
"""The reference's README quick-start tour (README.md:95-135 of QILaplace.jl) on the MI355X path.

    python examples/quickstart.py

Everything numeric (encode, apply, compress, coefficient) runs in libqilhip.so on the GPU; the
transform MPOs come from the host-side builders, as in the reference."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qilaplace_jl_amd as qil  # noqa: E402


def sin_decay(n, freq, decay_rate):
    """generate_signal(n; kind=:sin_decay, freq, decay_rate) of the reference (closed form)."""
    freq, decay_rate = np.atleast_1d(freq).astype(float), np.atleast_1d(decay_rate).astype(float)
    N = 2 ** n
    dt = 1.0 / (np.max(np.abs(freq)) * N)
    j = np.arange(N)
    return sum(np.sin(w * dt * j) * np.exp(-l * dt * j) for w, l in zip(freq, decay_rate))


def main():
    n = 10
    N = 2 ** n
    # --- Signal -> MPS -> QFT frequency bins
    signal = sin_decay(n, [1.0, 2.5], [0.08, 0.03])
    psi = qil.signal_mps(signal, method="rsvd", cutoff=1e-9, maxdim=64)
    qil.compress(psi, maxdim=64)
    Wqft = qil.build_qft_mpo(psi, cutoff=1e-12, maxdim=128)
    spectrum = Wqft * psi                                   # apply MPO to MPS (one HIP launch)
    amplitude = qil.coefficient(spectrum, "0101010110")     # bit-reversed frequency bin
    k = int("0101010110"[::-1], 2)                          # site 1 = LSB of the frequency index
    ref = np.fft.fft(signal)[k] / np.sqrt(N)
    print(f"QFT   coefficient('0101010110') = {amplitude:.12f}   fft reference = {ref:.12f}   "
          f"|diff| = {abs(amplitude - ref):.2e}   bonds = {spectrum.bond_dims}")

    # --- Damped and z-transform workflows (paired register)
    signal = sin_decay(n, 1.0, 0.05)
    psiz = qil.signal_ztmps(signal, method="svd", cutoff=1e-12)
    damped = qil.build_dt_mpo(psiz, 0.3, maxdim=64) * psiz
    response = qil.build_zt_mpo(psiz, 0.3, maxdim=128) * psiz
    kk, ll, jj = 341, 682, 5
    lsb = lambda v: [(v >> i) & 1 for i in range(n)]
    msb = lambda v: [(v >> (n - 1 - i)) & 1 for i in range(n)]
    il = lambda a, b: [x for p in zip(a, b) for x in p]
    a_dt = qil.coefficient(damped, il(lsb(kk), msb(jj)))
    r_dt = signal[jj] * np.exp(-0.3 * kk * jj / N) / np.sqrt(N)
    a_zt = qil.coefficient(response, il(lsb(kk), lsb(ll)))
    r_zt = np.sum(signal * np.exp(-(0.3 * kk + 2j * np.pi * ll) * np.arange(N) / N)) / N
    print(f"DT    <k={kk}, j={jj}|W psi>  = {a_dt:.12f}   closed form = {r_dt:.12f}   |diff| = {abs(a_dt - r_dt):.2e}")
    print(f"zT    chi(k={kk}, l={ll})      = {a_zt:.10f}   closed form = {r_zt:.10f}   |diff| = {abs(a_zt - r_zt):.2e}")
    # Laplace values as marginals (one chain each) and a (k, l) grid in batched launches
    L = qil.laplace_values(damped, np.arange(4), dt=1.0 / N)
    grid = qil.coefficient_grid(response, np.arange(32), np.arange(32))
    print(f"Laplace values L(s_0..3) = {np.round(L.real, 6)}   |chi| grid 32x32 max = {np.abs(grid).max():.6f}")
    ok = abs(amplitude - ref) < 1e-6 and abs(a_dt - r_dt) < 1e-7 and abs(a_zt - r_zt) < 2e-7
    print("OK" if ok else "MISMATCH")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
