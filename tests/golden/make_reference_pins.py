"""Generates tests/golden/reference_pins.json.

Every value below is DATA the reference itself publishes: printed outputs of its
executed tutorials (docs/src/tutorials/*.md) and integer series stored in its
committed benchmark artifact scripts/benchmark/results/mpo_bond_dim.jld2.  No
reference source text is copied.  The jld2 series are re-extracted when
/root/reference is present (this container only); otherwise the committed JSON
is authoritative.

Run:  python tests/golden/make_reference_pins.py
"""
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def scan_int_series(path):
    """Dict{Int,Int} series are stored as runs of (Int64 key, Int64 value) pairs in
    hash order; find every run whose keys are distinct values in 2..30."""
    b = open(path, "rb").read()
    out, i, L = [], 0, len(b)
    while i + 16 <= L:
        j, vals = i, []
        while j + 16 <= L:
            k, v = struct.unpack_from("<qq", b, j)
            if not (2 <= k <= 30) or not (1 <= v <= 2000):
                break
            vals.append((k, v))
            j += 16
        if len(vals) >= 20 and len({k for k, _ in vals}) == len(vals):
            out.append([v for _, v in sorted(vals)])
            i = j
            continue
        i += 1
    return out


pins = {
    # scripts/benchmark/results/mpo_bond_dim.jld2: max MPO bond for n = 2..30,
    # cutoff 1e-15, omega_r = 2 pi (scripts/benchmark/mpo_bond_dim.jl:21-24)
    "mpo_maxbond_n2_30": {
        "qft": [2, 2, 4, 4, 7, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8, 8],
        "zt": [8, 8, 37, 39, 78, 85, 89, 92, 92, 92, 91, 91, 91, 91, 91, 89, 89, 91, 91, 91, 91, 89, 89, 89, 89, 89, 91, 91, 89],
        "dt": [4, 4, 10, 11, 16, 17, 17, 18, 18, 18, 18, 18, 18, 18, 18, 17, 17, 18, 18, 18, 18, 17, 17, 17, 17, 17, 18, 18, 17],
    },
    # docs/src/tutorials/signal.md:24-74 -- generate_signal(4; kind=:sin, dt=1/16,
    # freq=[2pi,6pi], phase=[0.2,-0.4]); signal_mps(:svd, cutoff=1e-14) bonds
    "signal_tutorial": {
        "x": [-0.1907490115135893, 1.2605272231070337, 1.7601409795355494, 0.9887917666210433,
              0.059005583838356745, 0.11718557170551103, 0.9284594163678446, 1.1914820619096866,
              0.19074901151358986, -1.260527223107033, -1.7601409795355503, -0.9887917666210435,
              -0.059005583838356856, -0.11718557170551058, -0.9284594163678468, -1.1914820619096873],
        "bonds": [1, 2, 2],
    },
    # docs/src/tutorials/dft.md:71-75, 117-121, 139-143, 196
    "dft_tutorial": {
        "signal_bonds": [1, 2, 2],          # signal_mps(sin(2 pi j/16)), default cutoff
        "qft_mpo_bonds_n4": [2, 4, 2],      # build_qft_mpo(n=4, cutoff=1e-14)
        "applied_bonds": [2, 8, 4],         # W*psi: products, no truncation in apply
        "fft_error_l2": 3.4588662520960263e-15,
    },
    # docs/src/tutorials/dt.md:63-113, 232-241, 276, 319-326
    "dt_tutorial": {
        "n": 3, "dt": 0.3, "wr": 1.2, "a": 0.8,
        "ztmps_bonds_copy": [2, 2, 2], "ztmps_bonds_main": [1, 1],
        "applied_chain_bonds": [4, 4, 8, 4, 2],
        "L_s0": 1.199865702432454,
        "L_rounded5": [1.19987, 0.88794, 0.70943, 0.59949, 0.52726, 0.47721, 0.44101, 0.41393],
    },
    # docs/src/tutorials/zt.md:36-52, 106-113, 185-190, 223-227, 286-305
    "zt_tutorial": {
        "n": 2, "a": 0.7, "w0_over_pi": 1.0 / 3.0, "wr_over_pi": 2.0,
        "x_rounded4": [1.0, 0.35, -0.245, -0.343],
        "amp_match_j2": -0.24499999999999986,
        "zt_mpo_chain_bonds": [2, 8, 2],
        "chi_rounded4_re": [[0.1905, 0.3112, 0.187, 0.3112], [0.2648, 0.2526, 0.2299, 0.2526],
                            [0.2537, 0.2501, 0.2461, 0.2501], [0.2508, 0.25, 0.2492, 0.25]],
        "chi_rounded4_im": [[0.0, -0.1732, 0.0, 0.1732], [0.0, -0.019, 0.0, 0.019],
                            [0.0, -0.0038, 0.0, 0.0038], [0.0, -0.0008, 0.0, 0.0008]],
        "max_rel_err_published": 2.762e-15,
    },
    # docs/src/tutorials/zt.md:318-392 -- n=20 complex two-pole signal x_j = a^j cos(w0 j),
    # a = 1.00015 exp(0.002i), w0 = 0.0061; signal_ztmps(:rsvd, k=50, p=5, q=2, cutoff=1e-12, maxdim=128)
    "zt_tutorial_big": {
        "n": 20, "a_abs": 1.00015, "a_arg": 0.002, "w0": 0.0061,
        "k": 50, "p": 5, "q": 2, "cutoff": 1e-12, "maxdim": 128,
        "bonds_main": [1, 1, 1] + [2] * 16,
        "bonds_copy": [1, 1, 1, 2, 3] + [4] * 14 + [2],
        # executed output of the three |chi(k, l)| scans, docs/src/tutorials/zt.md:452-456, 519-523, 557-561
        "mpo_cutoff": 1e-12, "mpo_maxdim": 128,
        "coarse": {"wr": "2pi", "step": 4096, "peak_k": 0, "peak_l": 0, "pole_error": 4.102e-03},
        "fine": {"wr": 0.5, "peak_k": 0, "peak_l": 1047889, "z_re": 0.999992, "z_im": 0.004117, "pole_error": 1.509e-04},
        "superfine": {"wr": 0.5, "half": 24, "peak_k": 320, "peak_l": 1047872, "z_re": 0.999839, "z_im": 0.004218,
                      "pole_error": 1.185e-04},
    },
}

if os.path.isdir(REF):
    series = scan_int_series(os.path.join(REF, "scripts/benchmark/results/mpo_bond_dim.jld2"))
    want = pins["mpo_maxbond_n2_30"]
    for name in ("qft", "zt", "dt"):
        assert want[name] in series, f"{name} series not found in the jld2 artifact"

with open(os.path.join(HERE, "reference_pins.json"), "w") as f:
    json.dump(pins, f, indent=1)
print("wrote reference_pins.json")
