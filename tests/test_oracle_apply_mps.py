"""Oracle self-consistency tests mirroring the reference's own unit tests for the
path: test/test_apply.jl, test/test_mps.jl, test/test_signal_converters.jl,
test/test_rsvd.jl (gauge-invariant comparisons only)."""
import numpy as np
import pytest

import oracle as O
from helpers import (random_mps_data, random_mpo_data, dense_mps, dense_mpo, apply_dense,
                     all_bits, saturated_profile)


# ---------------------------------------------------------------- apply (test_apply.jl:50-217)
def test_apply_identity_pauli_random():
    rng = np.random.default_rng(7)
    n = 4
    psi = O.SignalMPS(random_mps_data([3, 3, 3], rng))
    ident = O.SingleSiteMPO.identity(n)
    out = O.apply(ident, psi)
    assert np.abs(dense_mps(out.data) - dense_mps(psi.data)).max() < 1e-12
    X = np.array([[0, 1], [1, 0]], dtype=float).reshape(1, 2, 2, 1)
    outx = O.apply(O.SingleSiteMPO([X] * n), psi)
    flipped = dense_mps(psi.data)[::-1, ::-1, ::-1, ::-1]
    assert np.abs(dense_mps(outx.data) - flipped).max() < 1e-12
    W = O.SingleSiteMPO(random_mpo_data([2, 2, 2], rng))
    outw = O.apply(W, psi)
    assert outw.bond_dims == [6, 6, 6]
    assert outw.amplitude == psi.amplitude
    assert np.abs(dense_mps(outw.data).reshape(-1) - apply_dense(W.data, psi.data)).max() < 1e-10


@pytest.mark.parametrize("wdt,adt", [(np.float64, np.float64), (np.complex128, np.float64),
                                     (np.float64, np.complex128), (np.complex128, np.complex128)])
def test_apply_dtype_combinations(wdt, adt):
    rng = np.random.default_rng(11)
    psi = O.SignalMPS(random_mps_data([2, 4, 3, 2], rng, adt))
    W = O.SingleSiteMPO(random_mpo_data([3, 5, 4, 2], rng, wdt))
    out = O.apply(W, psi)
    assert out.data[2].dtype == np.result_type(wdt, adt)
    assert np.abs(dense_mps(out.data).reshape(-1) - apply_dense(W.data, psi.data)).max() < 1e-12


def test_apply_errors():
    rng = np.random.default_rng(3)
    psi = O.SignalMPS(random_mps_data([2, 2], rng))
    with pytest.raises(ValueError, match="same number of sites"):
        O.apply(O.SingleSiteMPO.identity(4), psi)
    with pytest.raises(ValueError, match="same site indices"):
        O.apply(O.SingleSiteMPO.identity(3, sites=["a", "b", "c"]), psi)
    with pytest.raises(ValueError, match="compatible sizes"):
        O.apply(O.PairedSiteMPO.identity(3), O.ZTMPS(random_mps_data([2, 2, 2], rng)))


def test_apply_paired_identity():                       # test_apply.jl:277-300
    rng = np.random.default_rng(5)
    psi = O.ZTMPS(random_mps_data([2, 3, 2, 3, 2], rng), amplitude=2.5)
    out = O.apply(O.PairedSiteMPO.identity(3), psi)
    assert isinstance(out, O.ZTMPS) and out.amplitude == 2.5
    assert np.abs(dense_mps(out.data) - dense_mps(psi.data)).max() < 1e-12


def test_mpo_mpo_composition():                          # test_apply.jl:302-455
    rng = np.random.default_rng(9)
    W1 = O.SingleSiteMPO(random_mpo_data([2, 3, 2], rng))
    W2 = O.SingleSiteMPO(random_mpo_data([3, 2, 2], rng))
    W12 = O.apply(W1, W2)
    assert W12.bond_dims == [6, 6, 4]
    assert np.abs(dense_mpo(W12.data) - dense_mpo(W1.data) @ dense_mpo(W2.data)).max() < 1e-10
    psi = O.SignalMPS(random_mps_data([2, 4, 2], rng))
    seq = O.apply(W2, O.apply(W1, psi))
    one = O.apply(W12, psi)
    assert np.abs(dense_mps(seq.data) - dense_mps(one.data)).max() < 1e-10
    # unequal lengths: window embed
    Ws = O.SingleSiteMPO(random_mpo_data([2], rng), sites=[2, 3])
    Wl = O.SingleSiteMPO(random_mpo_data([2, 2, 2], rng), sites=[1, 2, 3, 4])
    emb = O.SingleSiteMPO([np.eye(2).reshape(1, 2, 2, 1)] + list(Ws.data) + [np.eye(2).reshape(1, 2, 2, 1)])
    got = O.apply(Ws, Wl)
    assert np.abs(dense_mpo(got.data) - dense_mpo(emb.data) @ dense_mpo(Wl.data)).max() < 1e-9
    got2 = O.apply(Wl, Ws)
    assert np.abs(dense_mpo(got2.data) - dense_mpo(Wl.data) @ dense_mpo(emb.data)).max() < 1e-9
    with pytest.raises(ValueError, match="No matching sites"):
        O.apply(O.SingleSiteMPO.identity(2, sites=[7, 8]), Wl)


# ---------------------------------------------------------------- coefficient (test_mps.jl:404-445)
def test_coefficient_front_ends_and_errors():
    data = []
    for b in (1, 0, 1):
        A = np.zeros((1, 2, 1)); A[0, b, 0] = 1.0; data.append(A)
    psi = O.SignalMPS(data, amplitude=3.0)
    for cfg in ([1, 0, 1], (1, 0, 1), "101", "[1,0,1]", "1 0 1", 0b101):
        assert abs(O.coefficient(psi, cfg) - 3.0) < 1e-12
    assert abs(O.coefficient(psi, "100")) < 1e-12
    with pytest.raises(ValueError, match="expected 3 entries"):
        O.coefficient(psi, [1, 0])
    with pytest.raises(ValueError, match="outside"):
        O.coefficient(psi, [1, 0, 2])
    with pytest.raises(ValueError, match="more than 3 bits"):
        O.coefficient(psi, 8)
    with pytest.raises(ValueError, match="non-negative"):
        O.coefficient(psi, -1)
    with pytest.raises(ValueError, match="only '0' or '1'"):
        O.coefficient(psi, "1a1")


def test_coefficient_matches_signal_and_vector_orders():  # test_signal_converters.jl:146-191
    x = np.arange(1.0, 9.0)
    psi = O.signal_mps(x)
    assert abs(psi.amplitude - np.linalg.norm(x)) < 1e-12
    assert abs(O.coefficient(psi, 5) - x[5]) < 1e-12      # bits 1,0,1 -> x[6] (1-based)
    assert np.abs(O.coefficient_batch(psi, all_bits(3)) - x).max() < 1e-12
    assert np.abs(O.mps_to_vector(psi) - x).max() < 1e-12
    rev = np.array([O.bitrev(i, 3) for i in range(8)])
    assert np.abs(O.mps_to_vector(psi, reverse=True) - x[rev]).max() < 1e-12


def test_lazy_coefficient_equals_materialised():
    rng = np.random.default_rng(21)
    L = 8
    psi = O.SignalMPS(random_mps_data(saturated_profile(L, 4), rng), amplitude=1.7)
    W = O.SingleSiteMPO(random_mpo_data(saturated_profile(L, 6, base=4), rng))
    bits = rng.integers(0, 2, size=(64, L))
    a = O.coefficient_batch(O.apply(W, psi), bits)
    b = O.lazy_coefficient_batch(W, psi, bits)
    assert np.abs(a - b).max() < 1e-13 * max(1.0, np.abs(a).max())


# ---------------------------------------------------------------- norm / canonicalize / compress
def test_norm_vs_dense():                                # test_mps.jl:268-328
    rng = np.random.default_rng(2)
    for dt in (np.float64, np.complex128):
        d = random_mps_data([2, 4, 4, 2], rng, dt, normalize=False)
        assert abs(O.norm(O.SignalMPS(d)) - np.linalg.norm(dense_mps(d))) < 1e-10


@pytest.mark.parametrize("direction", ["left", "right"])
def test_canonicalize_preserves_state(direction):        # test_mps.jl:156-180
    rng = np.random.default_rng(4)
    d = random_mps_data([2, 4, 4, 2], rng, np.complex128)
    psi = O.SignalMPS([t.copy() for t in d])
    O.canonicalize(psi, direction)
    assert np.abs(dense_mps(psi.data) - dense_mps(d)).max() < 1e-10
    if direction == "left":                              # sites 2..N right-orthogonal
        for A in psi.data[1:]:
            M = A.reshape(A.shape[0], -1)
            assert np.abs(M @ M.conj().T - np.eye(M.shape[0])).max() < 1e-10
    else:
        for A in psi.data[:-1]:
            M = A.reshape(-1, A.shape[2])
            assert np.abs(M.conj().T @ M - np.eye(M.shape[1])).max() < 1e-10
    with pytest.raises(ArithmeticError):
        O.canonicalize(psi, direction, center=9)
    with pytest.raises(ValueError):
        O.canonicalize(psi, "up")


def test_compress_postconditions():                      # test_mps.jl:331-369
    rng = np.random.default_rng(6)
    psi = O.SignalMPS(random_mps_data([2, 4, 2], rng, normalize=False), amplitude=1.0)
    before = dense_mps(psi.data)
    O.compress(psi, maxdim=2, tol=1e-8, sweeps=2)
    assert max(psi.bond_dims) <= 2
    assert abs(O.norm(psi) - 1.0) < 1e-8
    # lossless when maxdim is not binding: state (x amplitude) unchanged
    psi2 = O.SignalMPS(random_mps_data([2, 4, 2], rng, normalize=False))
    ref = dense_mps(psi2.data)
    O.compress(psi2, tol=1e-12)
    assert np.abs(dense_mps(psi2.data) * psi2.amplitude - ref).max() < 1e-10
    zt = O.ZTMPS(random_mps_data([2, 4, 4, 4, 2], rng, normalize=False))
    O.compress(zt, maxdim=2, tol=1e-8, sweeps=2)
    assert max(zt.bonds_copy + zt.bonds_main) <= 2 and abs(O.norm(zt) - 1.0) < 1e-8
    with pytest.raises(ArithmeticError):
        O.compress(O.SignalMPS([np.ones((1, 2, 1))]))
    del before


def test_compress_after_apply_recovers_low_rank():
    x = np.sin(2 * np.pi * np.arange(256) / 256 * 3.0)
    psi = O.signal_mps(x, cutoff=1e-14)
    out = O.apply(O.build_qft_mpo(8), psi)
    full = O.mps_to_vector(out)
    O.compress(out, tol=1e-5)             # MPO-cutoff noise (~1e-7 amplitude) is truncated away
    assert max(out.bond_dims) <= 2        # two spectral lines
    assert np.abs(O.mps_to_vector(out) - full).max() < 1e-5


# ---------------------------------------------------------------- encode (test_signal_converters.jl)
def test_signal_mps_svd_rsvd_reconstruct_and_agree():
    rng = np.random.default_rng(8)
    x = rng.standard_normal(64)
    for method, kw, tol in (("svd", {}, 1e-12), ("rsvd", dict(k=16, p=8, q=2), 1e-8)):
        psi = O.signal_mps(x, method=method, **kw)
        assert np.abs(O.mps_to_vector(psi) - x).max() < tol
    xc = x + 1j * rng.standard_normal(64)
    assert np.abs(O.mps_to_vector(O.signal_mps(xc)) - xc).max() < 1e-12
    with pytest.raises(ValueError, match="unknown method"):
        O.signal_mps(x, method="qr")
    with pytest.warns(UserWarning):
        psi = O.signal_mps(np.arange(1.0, 7.0))          # N=6 -> n = round(log2 6) = 3, zero-filled
    assert np.abs(O.mps_to_vector(psi)[:6] - np.arange(1.0, 7.0)).max() < 1e-12


def test_signal_ztmps_structure():
    rng = np.random.default_rng(10)
    x = rng.standard_normal(16)
    zt = O.signal_ztmps(x, cutoff=1e-14)
    n = 4
    for j in range(16):
        b = O.int_to_bits(j, n)
        bits = [v for pair in zip(b, b) for v in pair]
        assert abs(O.coefficient(zt, bits) - x[j]) < 1e-12
    assert abs(O.coefficient(zt, [0, 1] + [0, 0] * 3)) < 1e-13     # main != copy -> 0


def test_rsvd_low_rank_fixture():                        # test_rsvd.jl:5-16, 27-62
    rng = np.random.default_rng(12)
    m = 100
    U0, _ = np.linalg.qr(rng.standard_normal((m, 10)))
    V0, _ = np.linalg.qr(rng.standard_normal((m, 10)))
    s0 = np.exp(-np.arange(1, 11) / 2.0)
    A = (U0 * s0) @ V0.T
    U, S, Vh = O.rsvd(A, k=15, p=5, q=2)
    assert np.linalg.norm(A - (U * S) @ Vh) / np.linalg.norm(A) < 1e-10
    assert np.all(np.diff(S) <= 0) and np.all(S >= 0)
    assert np.abs(U.T @ U - np.eye(len(S))).max() < 1e-10
    assert np.abs(Vh @ Vh.T - np.eye(len(S))).max() < 1e-10
    assert len(O.rsvd(A, k=15, p=5, maxdim=4)[1]) == 4
    assert len(O.rsvd(A, k=15, p=5, cutoff=1e-4)[1]) < 10
    U2, S2, _ = O.rsvd(A, k=15, p=5, q=2)
    assert np.array_equal(S, S2)                         # seed determinism
    with pytest.raises(ValueError, match="empty"):
        O.rsvd(np.zeros((0, 4)))


def test_truncation_rule():
    s = np.array([1.0, 1e-3, 1e-8, 0.0])
    assert O.truncation_rank(s, cutoff=None) == 4
    assert O.truncation_rank(s, cutoff=0.0) == 3          # exact zeros are dropped
    assert O.truncation_rank(s, cutoff=1e-15) == 2        # 1e-16 <= 1e-15 * sum
    assert O.truncation_rank(s, cutoff=1e-15, maxdim=1) == 1
    assert O.truncation_rank(s, cutoff=1.0, mindim=2) == 2
    assert O.truncation_rank(np.zeros(3), cutoff=1e-15) == 1
