import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import qilaplace_jl_amd as qil
import oracle as O
from helpers import dense_mps

def schmidt(vec, n2, cut):
    M = vec.reshape(2 ** cut, -1)
    return np.linalg.svd(M, compute_uv=False)

for n, maxdim, tol in [(8, 12, 1e-8), (10, 16, 1e-6), (10, 24, 1e-10), (10, 16, 1e-4), (10, 8, 1e-3)]:
    x = O.generate_signal(n, kind="sin_decay", freq=[1.0, 2.5], decay_rate=[0.08, 0.03])
    psi = qil.signal_ztmps(x, cutoff=1e-12)
    W = qil.build_zt_mpo(psi, 2 * np.pi)
    ref = O.apply(O.SingleSiteMPO(W.to_host()), O.SignalMPS(psi.to_host(), amplitude=psi.amplitude))
    exact = ref.amplitude * dense_mps(ref.data)
    O.compress(ref, maxdim=maxdim, tol=tol)
    want = ref.amplitude * dense_mps(ref.data)
    fused = qil.apply_compress(W, psi, maxdim=maxdim, tol=tol)
    got = fused.amplitude * dense_mps(fused.to_host())
    slow = W * psi
    qil.compress(slow, maxdim=maxdim, tol=tol)
    nrm = np.linalg.norm(exact)
    print(f"n={n} maxdim={maxdim} tol={tol:g}: e_trunc {np.linalg.norm(want-exact)/nrm:.2e} e_fused {np.linalg.norm(got-exact)/nrm:.2e} "
          f"e_slow {np.linalg.norm(slow.amplitude*dense_mps(slow.to_host())-exact)/nrm:.2e}")
    print("  oracle", ref.bond_dims); print("  fused ", fused.bond_dims); print("  slow  ", slow.bond_dims)
    bd_o, bd_f = ref.bond_dims, fused.bond_dims
    for b, (p, q) in enumerate(zip(bd_o, bd_f)):
        if p != q:
            s = schmidt(exact, 2 * n, b + 1)
            s /= np.linalg.norm(s)
            cutoff = tol * tol / (2 * n - 1)
            print(f"  bond {b}: oracle {p} fused {q}; exact schmidt^2 around: {np.array2string(s[min(p,q)-2:max(p,q)+2]**2, precision=2)} cutoff {cutoff:.1e}")
