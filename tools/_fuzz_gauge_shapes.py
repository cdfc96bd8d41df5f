"""Fuzz of the gauge / truncation sweeps over site shapes around every route boundary of the device SVD (17, 96/97, 640
columns; tall, wide and SQUARE sites; both sweep directions; f64 and c64; certificate on and off; full-rank and planted
rank-deficient bonds).  Checks, gauge-invariantly and without the oracle (so bonds of 1000+ stay cheap): canonicalize!(1e-12)
of a full-rank chain keeps the state to sqrt(cutoff) (sampled coefficients against the untouched chain's, which never go
through an SVD), leaves isometries behind, gives the same bonds and coefficients with the certificate on and off, and the certificate on/off runs agree on the bond dimensions of a chain with
planted deficient bonds and of compress!(maxdim)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import qilaplace_jl_amd as qil


def chain(bonds, rng, dtype):
    dims = [1] + list(bonds) + [1]
    out = []
    for i in range(len(dims) - 1):
        t = rng.standard_normal((dims[i], 2, dims[i + 1]))
        if dtype == np.complex128:
            t = t + 1j * rng.standard_normal(t.shape)
        out.append((t / np.sqrt(dims[i])).astype(dtype))
    return out


def ramp(peak_profile):
    """2,4,8,... up to the first entry, the given profile, then halving back down to 2."""
    up, b = [], 2
    while b < peak_profile[0]:
        up.append(b); b *= 2
    down, b = [], peak_profile[-1]
    while b > 2:
        b = (b + 1) // 2; down.append(b)
    return up + list(peak_profile) + down


PROFILES = [
    [16, 17, 17, 16], [24, 48, 96, 97, 96, 48], [60, 97, 120, 97], [128, 256, 256, 128],
    [320, 640, 640, 320], [320, 639, 641, 400], [512, 1024, 512], [400, 800, 700, 350],
    [350, 700, 1400, 700], [640, 1280, 640], [330, 660, 660, 660, 330],
]

fails = 0
t0 = time.time()
for pi, prof in enumerate(PROFILES):
    bonds = ramp(prof)
    L = len(bonds) + 1
    for dtype in (np.float64, np.complex128):
        if dtype == np.complex128 and max(bonds) > 1100:
            continue
        rng = np.random.default_rng(100 * pi + (dtype == np.complex128))
        a = chain(bonds, rng, dtype)
        bits = rng.integers(0, 2, size=(48, L))
        want = qil.coefficient_batch(qil.SignalMPS([t.copy() for t in a]), bits)
        # planted deficiency at the peak bond (a left -> right sweep, direction "right", meets it and must cut the bond to
        # the planted rank; the opposite sweep factors the neighbour first and legitimately keeps it): project the right bond of the site before the peak onto rank r
        ip = int(np.argmax(bonds))
        cr = a[ip].shape[2]
        rdef = max(2, (3 * cr) // 5)
        P = rng.standard_normal((cr, rdef)) @ rng.standard_normal((rdef, cr)) / cr
        adef = [t.copy() for t in a]
        adef[ip] = np.einsum("asb,bc->asc", a[ip], P).astype(dtype)
        res = {}
        for cert in ("1", "0"):
            os.environ["QIL_SVD_CERT"] = cert
            for direction in ("left", "right"):
                psi = qil.SignalMPS([t.copy() for t in a])
                qil.canonicalize(psi, direction, cutoff=1e-12)
                got = qil.coefficient_batch(psi, bits)
                err = np.abs(got - want).max() / np.abs(want).max()
                iso = 0.0
                host = psi.to_host()
                for t in (host[1:] if direction == "left" else host[:-1]):
                    m = t.reshape(t.shape[0], -1) if direction == "left" else t.reshape(-1, t.shape[2]).conj().T
                    iso = max(iso, np.abs(m @ m.conj().T - np.eye(m.shape[0])).max())
                # a random chain's Schmidt tails can dip under the cutoff: bonds may shrink and the state moves by ~sqrt(cutoff)
                ok = all(x <= y for x, y in zip(psi.bond_dims, bonds)) and err < 2e-5 and iso < 1e-11
                if not ok:
                    fails += 1
                    print("FAIL canon", prof, dtype.__name__, "cert", cert, direction, psi.bond_dims == bonds, err, iso, flush=True)
                res[(cert, direction, "full")] = (psi.bond_dims, got)
                phi = qil.SignalMPS([t.copy() for t in adef])
                qil.canonicalize(phi, direction, cutoff=1e-12)
                res[(cert, direction, "def")] = (phi.bond_dims, qil.coefficient_batch(phi, bits))
            phi = qil.SignalMPS([t.copy() for t in a])
            qil.compress(phi, maxdim=max(8, max(bonds) // 3), tol=1e-9)
            res[(cert, "cmp")] = (phi.bond_dims, qil.coefficient_batch(phi, bits))
        wdef = qil.coefficient_batch(qil.SignalMPS([t.copy() for t in adef]), bits)
        for direction in ("left", "right"):
            b1, c1 = res[("1", direction, "def")]
            b0, c0 = res[("0", direction, "def")]
            e1 = np.abs(c1 - wdef).max() / np.abs(wdef).max()
            e0 = np.abs(c0 - wdef).max() / np.abs(wdef).max()
            if b1 != b0 or e1 > 2e-5 or e0 > 2e-5 or (direction == "right" and b1[ip] > rdef) or np.abs(c1 - c0).max() > 1e-9 * np.abs(wdef).max():
                fails += 1
                print("FAIL deficient", prof, dtype.__name__, direction, b1 == b0, b1[ip], rdef, e1, e0, flush=True)
            b1, c1 = res[("1", direction, "full")]
            b0, c0 = res[("0", direction, "full")]
            if b1 != b0 or np.abs(c1 - c0).max() > 1e-9 * np.abs(want).max():
                fails += 1
                print("FAIL cert on/off", prof, dtype.__name__, direction, b1, b0, np.abs(c1 - c0).max() / np.abs(want).max(), flush=True)
        b1, c1 = res[("1", "cmp")]
        b0, c0 = res[("0", "cmp")]
        if b1 != b0 or np.abs(c1 - c0).max() > 1e-9 * np.abs(want).max():
            fails += 1
            print("FAIL compress", prof, dtype.__name__, b1 == b0, np.abs(c1 - c0).max() / np.abs(want).max(), flush=True)
        print("done", prof, dtype.__name__, round(time.time() - t0, 1), "s", flush=True)
print({"profiles": len(PROFILES), "failures": fails})
