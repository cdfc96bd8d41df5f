"""libqilcpu.so (oracle/cpu): the C++/OpenMP CPU baseline behind the same C ABI -- apply and coefficient against the
numpy oracle on the same seeded inputs.  CPU only; the library is baseline infrastructure, never part of the product."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle as O
from helpers import random_mps_data, random_mpo_data, saturated_profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "oracle", "cpu", "lib", "libqilcpu.so")


@pytest.fixture(scope="module")
def cpu():
    if not os.path.exists(LIB):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle", "cpu")], check=True)
    from oracle.cpu_backend import CpuBackend
    return CpuBackend(LIB)


@pytest.mark.parametrize("wdt,adt", [(np.float64, np.float64), (np.complex128, np.float64), (np.float64, np.complex128),
                                     (np.complex128, np.complex128)])
def test_cpu_apply_matches_oracle(cpu, wdt, adt):
    rng = np.random.default_rng(77)
    L = 7
    a = random_mps_data(saturated_profile(L, 6), rng, adt)
    w = random_mpo_data(saturated_profile(L, 5, base=4), rng, wdt)
    ref = O.apply(O.SingleSiteMPO(w), O.SignalMPS(a, amplitude=1.3))
    for threads in (1, 0):
        cpu.set_threads(threads)
        got = cpu.apply(w, a)
        for g, r in zip(got, ref.data):
            assert g.shape == r.shape and np.abs(g - r).max() <= 1e-14 * max(1.0, np.abs(r).max())
    bits = rng.integers(0, 2, size=(32, L)).astype(np.uint8)
    c = cpu.apply_coefficients(w, a, bits, amplitude=1.3)
    assert np.abs(c - O.coefficient_batch(ref, bits)).max() < 1e-12 * np.abs(c).max()


def test_cpu_apply_errors(cpu):
    rng = np.random.default_rng(1)
    a = random_mps_data([2, 2], rng)
    w = random_mpo_data([2], rng)
    with pytest.raises(ValueError, match="same number of sites"):
        cpu.apply(w, a)
