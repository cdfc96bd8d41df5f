"""numpy model of the one-sided Jacobi iteration of the one-factor SVD (round-robin pairing, the library's block pairing with
blocks of 8 / 16, its tolerances): sweeps needed on the columns of R against the columns of R^T, for flat (Gaussian) and
graded spectra.  CPU only:  python tools/_jacobi_orientation_numpy.py   (MEASUREMENTS.md 3.4, "Which factor to rotate")
flat: 10 sweeps in every variant; graded over 2 / 6 / 12 decades: R 14 / 24-26 / 38-40 sweeps, R^T 11-12."""
import numpy as np, sys
rng = np.random.default_rng(1)
def rr_rounds(n):
    # classical round-robin: n even, n-1 rounds of n/2 disjoint pairs
    idx = list(range(n))
    out = []
    for r in range(n - 1):
        pairs = [(idx[i], idx[n - 1 - i]) for i in range(n // 2)]
        out.append(pairs)
        idx = [idx[0]] + [idx[-1]] + idx[1:-1]
    return out
def block_rounds(n, bb):
    # my scheme: blocks of bb columns; block round-robin; round 0 of each sweep = all pairs inside each block pair (AP),
    # later rounds = cross pairs of the paired blocks, bb inner rounds each
    nb = n // bb
    rounds = []
    brr = rr_rounds(nb)
    for r, bpairs in enumerate(brr):
        if r == 0:
            # all pairs within the 2bb columns of each block pair: round-robin over 2bb columns
            inner = rr_rounds(2 * bb)
            for ir in inner:
                pairs = []
                for (P, Q) in bpairs:
                    cols = list(range(P * bb, P * bb + bb)) + list(range(Q * bb, Q * bb + bb))
                    pairs += [(cols[a], cols[b]) for a, b in ir]
                rounds.append(pairs)
        else:
            for s in range(bb):
                pairs = []
                for (P, Q) in bpairs:
                    pairs += [(P * bb + a, Q * bb + (a + s) % bb) for a in range(bb)]
                rounds.append(pairs)
    return rounds
def sweeps_needed(X, rounds, tol=7e-15, quad=1e-8, maxs=40):
    X = X.copy()
    for s in range(maxs):
        big = 0
        for pairs in rounds:
            p = np.array([a for a, b in pairs]); q = np.array([b for a, b in pairs])
            x, y = X[:, p], X[:, q]
            al, be, g = (x * x).sum(0), (y * y).sum(0), (x * y).sum(0)
            rel = np.abs(g) / np.sqrt(al * be)
            big = max(big, rel.max())
            rot = rel > tol
            d = be - al
            t = np.sign(d) * 2 * g / (np.abs(d) + np.sqrt(d * d + 4 * g * g)); t[d == 0] = np.sign(g[d == 0])
            c = 1 / np.sqrt(1 + t * t); sn = c * t
            c = np.where(rot, c, 1.0); sn = np.where(rot, sn, 0.0)
            X[:, p], X[:, q] = c * x - sn * y, sn * x + c * y
        if big < quad: return s + 1
    return maxs
n = 256
for trial in range(2):
    A = rng.standard_normal((2 * n, n))
    R = np.linalg.qr(A, mode="r")
    for name, M in (("R", R), ("R^T", R.T.copy())):
        print(trial, name, "round-robin:", sweeps_needed(M, rr_rounds(n)), " blocks of 8:", sweeps_needed(M, block_rounds(n, 8)), " blocks of 16:", sweeps_needed(M, block_rounds(n, 16)), flush=True)
print("--- graded spectra")
for decades in (2, 6, 12):
    U = np.linalg.qr(rng.standard_normal((2 * n, n)))[0]; V = np.linalg.qr(rng.standard_normal((n, n)))[0]
    s = np.logspace(0, -decades, n)
    A = (U * s) @ V.T
    R = np.linalg.qr(A, mode="r")
    for name, M in (("R", R), ("R^T", R.T.copy())):
        print(decades, name, "round-robin:", sweeps_needed(M, rr_rounds(n), quad=1e-9), " blocks of 8:", sweeps_needed(M, block_rounds(n, 8), quad=1e-9), flush=True)
