"""Host-side MPO builders of the product package (qilaplace.jl_amd/builders.py) against the oracle's
dense operators, the closed forms, and the reference's committed bond-dimension series."""
import numpy as np
import pytest

import oracle as O
from helpers import dense_mpo


@pytest.fixture(scope="module")
def qil():
    import qilaplace_jl_amd as q
    return q


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5])
def test_qft_tensors_are_the_bit_reversed_dft(qil, n):
    M = dense_mpo(qil.qft_mpo_tensors(n))
    assert np.abs(M - O.qn_matrix(n).T).max() < 1e-10


@pytest.mark.parametrize("n", [1, 2, 3, 4])
@pytest.mark.parametrize("wr", [0.0, 0.75, 5.0])
def test_dt_zt_tensors_equal_oracle_operators(qil, n, wr):
    for mine, ref in ((qil.dt_mpo_tensors(n, wr), O.build_dt_mpo(n, wr).data),
                      (qil.zt_mpo_tensors(n, wr), O.build_zt_mpo(n, wr).data)):
        assert [t.shape[3] for t in mine] == [t.shape[3] for t in ref]
        assert np.abs(dense_mpo(mine) - dense_mpo(ref)).max() < 2e-7


def test_bond_series_match_reference_artifact(qil, pins):
    want = pins["mpo_maxbond_n2_30"]
    mb = lambda ts: max(t.shape[3] for t in ts)
    for i, n in enumerate(range(2, 9)):
        assert mb(qil.qft_mpo_tensors(n, cutoff=1e-15, maxdim=None)) == want["qft"][i]
        assert mb(qil.dt_mpo_tensors(n, 2 * np.pi, cutoff=1e-15, maxdim=None)) == want["dt"][i]
        assert mb(qil.zt_mpo_tensors(n, 2 * np.pi, cutoff=1e-15, maxdim=None)) == want["zt"][i]
    assert mb(qil.qft_mpo_tensors(14, cutoff=1e-15)) == want["qft"][12]
    assert mb(qil.dt_mpo_tensors(10, 2 * np.pi, cutoff=1e-15)) == want["dt"][8]


def test_tutorial_bond_pins(qil, pins):
    assert [t.shape[3] for t in qil.qft_mpo_tensors(4, cutoff=1e-14, maxdim=100)][:-1] == pins["dft_tutorial"]["qft_mpo_bonds_n4"]
    assert [t.shape[3] for t in qil.zt_mpo_tensors(2, 2 * np.pi, cutoff=1e-14, maxdim=64)][:-1] == pins["zt_tutorial"]["zt_mpo_chain_bonds"]
    many = qil.dt_mpo_tensors_many(3, [0.5, 1.2], workers=2)
    assert np.abs(dense_mpo(many[1]) - dense_mpo(O.build_dt_mpo(3, 1.2).data)).max() < 1e-12


def test_builder_argument_errors(qil):
    with pytest.raises(ValueError):
        qil.qft_mpo_tensors(0)
    with pytest.raises(ValueError):
        qil.dt_mpo_tensors(0, 1.0)
    with pytest.raises(ValueError):
        qil.zt_mpo_tensors(0, 1.0)


def test_zt_qft_half_is_built_from_cached_prefixes():
    """The damping-independent half of build_zt_mpo for n is the one for n - 1 zipped with one more block
    (zt_transformer.jl:78-98): every intermediate chain is cached, a chain continued from a cached prefix equals the one
    built from scratch bit for bit, and the cached prefixes are not modified by the continuation."""
    from qilaplace_jl_amd import builders as B
    B._ZT_Q_CACHE.clear()
    scratch = [t.copy() for t in B.zt_qft_chain_tensors(7)]
    B._ZT_Q_CACHE.clear()
    five = B.zt_qft_chain_tensors(5)
    keep = [t.copy() for t in five]
    cont = B.zt_qft_chain_tensors(7)
    assert len(cont) == 14 and all(np.array_equal(a, b) for a, b in zip(scratch, cont))
    assert all(np.array_equal(a, b) for a, b in zip(five, keep))
    assert (6, 1e-14, 1000) in B._ZT_Q_CACHE and len(B.zt_qft_chain_tensors(6)) == 12
