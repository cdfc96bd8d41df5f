"""Pins the CPU oracle against everything the reference publishes for this path:
closed-form oracles of its own tests and the printed outputs of its executed
tutorials (tests/golden/reference_pins.json)."""
import numpy as np
import pytest

import oracle as O
from oracle.analytic import int_to_bits
from helpers import dense_mpo, basis_mps, basis_ztmps, interleave, all_bits


# ---- QFT: test/test_qft_transformer.jl:331-374 (basis states vs Q_n), atol 1e-10
@pytest.mark.parametrize("n", [1, 2, 3, 4, 5])
def test_qft_mpo_is_bit_reversed_dft(n):
    W = O.build_qft_mpo(n, cutoff=1e-14, maxdim=1000)
    M = dense_mpo(W.data)                      # M[in, out]
    Q = O.qn_matrix(n)                         # Q[j, k] = exp(-2 pi i bitrev(j) k / N)/sqrt N
    assert np.abs(M - Q.T).max() < 1e-10


# ---- QFT vs FFT on a random complex signal: test_qft_transformer.jl:427-464, atol 1e-10
@pytest.mark.parametrize("n", [2, 3, 4, 5])
def test_qft_random_signal_vs_fft(n):
    rng = np.random.default_rng(100 + n)
    N = 2 ** n
    sig = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    psi = O.signal_mps(sig)
    out = O.apply(O.build_qft_mpo(n), psi, cutoff=0.0, maxdim=1000)
    fn = O.mps_to_vector(out, reverse=True)
    assert np.linalg.norm(fn - np.fft.fft(sig) / np.sqrt(N)) < 1e-10
    qn = O.mps_to_vector(out, reverse=False)
    rev = np.array([O.bitrev(i, n) for i in range(N)])
    assert np.linalg.norm(qn[rev] - fn) < 1e-12


# ---- DT: test/test_dt_transformer.jl:211-238, tolerance 1e-7 * max(1, ||.||)
@pytest.mark.parametrize("n", [1, 2, 3, 4])
@pytest.mark.parametrize("wr", [0.0, 0.75, 1.0, 2.0, 5.0])
def test_dt_mpo_basis_states(n, wr):
    N = 2 ** n
    W = O.build_dt_mpo(n, wr)
    for j in range(N):
        out = O.apply(W, basis_ztmps(j, n))
        ref = O.analytical_dt(np.eye(N)[j], wr)
        bits = [interleave(int_to_bits(k, n, "lsb"), int_to_bits(j, n)) for k in range(N)]
        got = O.coefficient_batch(out, np.array(bits))
        assert np.abs(got - ref).max() < 1e-7 * max(1.0, np.linalg.norm(ref))


# ---- zT: test/test_zt_transformer.jl:68-109, abs 2e-7
@pytest.mark.parametrize("n", [1, 2, 3, 4])
@pytest.mark.parametrize("wr", [0.0, 0.75, 1.0, 2.0, 5.0])
def test_zt_mpo_basis_states(n, wr):
    N = 2 ** n
    W = O.build_zt_mpo(n, wr)
    bits = np.array([interleave(int_to_bits(k, n, "lsb"), int_to_bits(l, n, "lsb"))
                     for k in range(N) for l in range(N)])
    for j in range(N):
        out = O.apply(W, basis_ztmps(j, n))
        ref = O.analytical_zt(np.eye(N)[j], wr=wr).reshape(-1)
        assert np.abs(O.coefficient_batch(out, bits) - ref).max() < 2e-7


# ---- MPO bond-dimension series from the reference's committed benchmark artifact
def test_mpo_maxbond_series(pins):
    want = pins["mpo_maxbond_n2_30"]
    for i, n in enumerate(range(2, 9)):
        assert max(O.build_qft_mpo(n, cutoff=1e-15, maxdim=None).bond_dims) == want["qft"][i]
        assert max(O.build_dt_mpo(n, 2 * np.pi, cutoff=1e-15, maxdim=None).bond_dims) == want["dt"][i]
        assert max(O.build_zt_mpo(n, 2 * np.pi, cutoff=1e-15, maxdim=None).bond_dims) == want["zt"][i]


def test_qft_maxbond_saturates_at_8(pins):
    want = pins["mpo_maxbond_n2_30"]["qft"]
    for n in (12, 16):
        assert max(O.build_qft_mpo(n, cutoff=1e-15).bond_dims) == want[n - 2]


# ---- signal tutorial: docs/src/tutorials/signal.md
def test_signal_tutorial(pins):
    p = pins["signal_tutorial"]
    x = O.generate_signal(4, kind="sin", dt=1 / 16, freq=[2 * np.pi, 6 * np.pi], phase=[0.2, -0.4])
    assert np.abs(x - np.array(p["x"])).max() < 1e-14
    psi = O.signal_mps(x, method="svd", cutoff=1e-14)
    assert psi.bond_dims == p["bonds"]
    got = O.coefficient_batch(psi, all_bits(4))
    assert np.abs(got - x).max() < 1e-12


# ---- DFT tutorial: docs/src/tutorials/dft.md
def test_dft_tutorial(pins):
    p = pins["dft_tutorial"]
    x = np.sin(2 * np.pi * np.arange(16) / 16)
    psi = O.signal_mps(x)
    assert psi.bond_dims == p["signal_bonds"]
    W = O.build_qft_mpo(4, cutoff=1e-14, maxdim=100)
    assert W.bond_dims == p["qft_mpo_bonds_n4"]
    out = O.apply(W, psi)
    assert out.bond_dims == p["applied_bonds"]
    err = np.linalg.norm(O.mps_to_vector(out, reverse=True) - np.fft.fft(x) / 4.0)
    assert err < 1e-13                               # published: 3.46e-15


# ---- DT tutorial: docs/src/tutorials/dt.md
def test_dt_tutorial(pins):
    p = pins["dt_tutorial"]
    n, dt, wr = p["n"], p["dt"], p["wr"]
    N = 2 ** n
    x = np.exp(-p["a"] * dt * np.arange(N))
    psiz = O.signal_ztmps(x, cutoff=1e-14, maxdim=64)
    assert psiz.bonds_copy == p["ztmps_bonds_copy"]
    assert psiz.bonds_main == p["ztmps_bonds_main"]
    W = O.build_dt_mpo(n, wr, cutoff=1e-14, maxdim=64)
    out = O.apply(W, psiz)
    assert out.as_signal_2n().bond_dims == p["applied_chain_bonds"]

    def laplace(k):
        bits = np.array([interleave(int_to_bits(k, n, "lsb"), int_to_bits(j, n)) for j in range(N)])
        return dt * np.sqrt(N) * O.coefficient_batch(out, bits).sum()

    assert abs(laplace(0) - p["L_s0"]) < 1e-14
    L = np.array([laplace(k) for k in range(N)])
    assert np.abs(L.real - np.array(p["L_rounded5"])).max() <= 0.5e-5 + 1e-9   # printed to 5 decimals in the tutorial
    ref = dt * np.sqrt(N) * O.analytical_dt(x, wr)
    assert np.abs(L - ref).max() < 1e-14


# ---- zT tutorial: docs/src/tutorials/zt.md
def test_zt_tutorial(pins):
    p = pins["zt_tutorial"]
    n = p["n"]
    N = 2 ** n
    x = np.array([p["a"] ** j * np.cos(np.pi * p["w0_over_pi"] * j) for j in range(N)])
    assert np.abs(x - np.array(p["x_rounded4"])).max() <= 0.5e-4 + 1e-9   # printed to 4 decimals in the tutorial
    psiz = O.signal_ztmps(x, cutoff=1e-14, maxdim=64)
    b2 = int_to_bits(2, n)
    assert abs(O.coefficient(psiz, interleave(b2, b2)) - p["amp_match_j2"]) < 1e-15
    W = O.build_zt_mpo(n, np.pi * p["wr_over_pi"], cutoff=1e-14, maxdim=64)
    assert W.bond_dims == p["zt_mpo_chain_bonds"]
    out = O.apply(W, psiz)
    bits = np.array([interleave(int_to_bits(k, n, "lsb"), int_to_bits(l, n, "lsb"))
                     for k in range(N) for l in range(N)])
    chi = O.coefficient_batch(out, bits).reshape(N, N)
    assert np.abs(chi.real - np.array(p["chi_rounded4_re"])).max() <= 0.5e-4 + 1e-9   # printed to 4 decimals in the tutorial
    assert np.abs(chi.imag - np.array(p["chi_rounded4_im"])).max() <= 0.5e-4 + 1e-9   # printed to 4 decimals in the tutorial
    ref = O.analytical_zt(x, wr=2 * np.pi, wi=2 * np.pi)
    assert (np.abs(chi - ref) / np.abs(ref)).max() < 1e-13


# ---- zT tutorial, large signal: docs/src/tutorials/zt.md:318-392 (68 orders of magnitude of dynamic range)
def test_zt_tutorial_big_signal_bond_structure(pins):
    p = pins["zt_tutorial_big"]
    N = 2 ** p["n"]
    j = np.arange(N)
    x = (p["a_abs"] * np.exp(1j * p["a_arg"])) ** j * np.cos(p["w0"] * j)
    zt = O.signal_ztmps(x, method="rsvd", k=p["k"], p=p["p"], q=p["q"], cutoff=p["cutoff"], maxdim=p["maxdim"])
    assert zt.bonds_main == p["bonds_main"] and zt.bonds_copy == p["bonds_copy"]


def test_zt_tutorial_pole_scans(pins):
    """docs/src/tutorials/zt.md:430-561: the three |chi(k, l)| scans of the n = 20 two-pole run.  The oracle's pipeline
    (RSVD encode -> build_zt_mpo -> apply -> coefficient) lands on the peak indices and pole errors the reference's
    executed tutorial printed."""
    p = pins["zt_tutorial_big"]
    n = p["n"]
    N = 2 ** n
    j = np.arange(N)
    a, w0 = p["a_abs"] * np.exp(1j * p["a_arg"]), p["w0"]
    x = a ** j * np.cos(w0 * j)
    poles = [np.exp(1j * w0) / a, np.exp(-1j * w0) / a]
    psi = O.signal_ztmps(x, method="rsvd", k=p["k"], p=p["p"], q=p["q"], cutoff=p["cutoff"], maxdim=p["maxdim"])
    wi = 2 * np.pi

    def scan(phi, ks, ls, wr):
        kb, lb = ((ks[:, None] >> np.arange(n)) & 1), ((ls[:, None] >> np.arange(n)) & 1)
        bits = np.empty((len(ks), len(ls), 2 * n), dtype=np.uint8)
        bits[:, :, 0::2] = kb[:, None, :]
        bits[:, :, 1::2] = lb[None, :, :]
        chi = O.coefficient_batch(phi, bits.reshape(-1, 2 * n)).reshape(len(ks), len(ls))
        i, m = np.unravel_index(np.argmax(np.abs(chi)), chi.shape)
        r, th = np.exp(-wr * ks[i] / N), wi * ls[m] / N
        z = complex(r * np.cos(th), -r * np.sin(th))
        return int(ks[i]), int(ls[m]), min(abs(z - q) for q in poles)

    def check(got, want):
        assert got[:2] == (want["peak_k"], want["peak_l"])
        assert abs(got[2] - want["pole_error"]) <= 0.5e-3 * want["pole_error"]

    wr = 2 * np.pi
    phi = O.apply(O.build_zt_mpo(n, wr, cutoff=p["mpo_cutoff"], maxdim=p["mpo_maxdim"]), psi)
    ks = np.arange(0, N, p["coarse"]["step"])
    check(scan(phi, ks, ks, wr), p["coarse"])
    wr = 0.5
    phi = O.apply(O.build_zt_mpo(n, wr, cutoff=p["mpo_cutoff"], maxdim=p["mpo_maxdim"]), psi)
    ks = np.clip(np.rint((-N / wr) * np.log(np.linspace(1 - 1.6e-4, 1.0, 128))).astype(np.int64), 0, N - 1)
    ls = np.mod(np.rint((N / wi) * np.mod(np.linspace(-5e-3, 9e-3, 128), 2 * np.pi)).astype(np.int64), N)
    check(scan(phi, ks, ls, wr), p["fine"])
    zt = poles[0]
    kc = int(np.clip(np.rint((-N / wr) * np.log(abs(zt))), 0, N - 1))
    lc = int(np.mod(np.rint((N / wi) * np.mod(-np.angle(zt), 2 * np.pi)), N))
    h = p["superfine"]["half"]
    check(scan(phi, np.arange(kc - h, kc + h + 1), np.mod(np.arange(lc - h, lc + h + 1), N), wr), p["superfine"])
