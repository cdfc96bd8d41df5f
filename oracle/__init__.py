"""CPU oracle for the QILaplace MPO x MPS hot path -- TEST INFRASTRUCTURE ONLY.

This package is a plain numpy restatement of the reference algorithm
(SUTD-MDQS/QILaplace.jl v0.1.1, Julia on ITensors.jl 0.9).  It exists to CHECK
the HIP path; it is never the thing measured or shipped.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it.  The product package (``qilaplace.jl_amd``) must never import it.

Parity status: PINNED against the reference's own closed-form oracles and
executed tutorial outputs (see ``tests/test_oracle_*.py`` and
``tests/golden/``):
  * QFT      vs bit-reversed DFT / numpy FFT   (test/test_qft_transformer.jl:6-34, 331-464)
  * DT       vs analytical_dt                  (test/test_dt_transformer.jl:71-92, 211-238)
  * zT       vs analytical_zt                  (test/test_zt_transformer.jl:20-39, 68-109)
  * apply    vs dense contraction              (test/test_apply.jl:50-92, 174-217, 302-455)
  * tutorial outputs (bond dimensions, Laplace values, 4x4 chi table)
    docs/src/tutorials/{signal,dft,dt,zt}.md
UNPINNED (ITensors.jl is an un-vendored dependency, Project.toml:9,18, no
Manifest): the exact SVD/QR gauge and sign conventions, ``factorize``'s choice
of decomposition and Julia's RNG streams.  No reference test depends on them
beyond gauge-invariant outputs, and neither do ours.

The reference cannot be built or run here (no ``julia`` binary, no ITensors
source, no network), so there is no ``oracle/_ref``.

Array conventions (index order, NOT memory order):
  MPS site  A[alpha, s, beta]            shape (chi_l, 2, chi_r)
  MPO site  W[a, s_in, s_out, b]         shape (D_l, 2, 2, D_r)
            s_in  = the reference's primed leg  s'  (contracted with the MPS)
            s_out = the reference's unprimed leg s  (survives)   apply.jl:98-101
Edge tensors carry explicit dim-1 bonds.
"""

from .containers import SignalMPS, ZTMPS, SingleSiteMPO, PairedSiteMPO  # noqa: F401
from .linalg import svd_trunc, truncation_rank, rsvd, qr_positive  # noqa: F401
from .gates import (  # noqa: F401
    gate_I, gate_H, gate_P, gate_Pi, gate_dampedH, gate_R,
    control_Hphase_mpo, control_damping_mpo, control_damping_copy_mpo,
    control_Hphase_ztmps_mpo,
)
from .apply import apply, apply_mpo_mpo, apply_site  # noqa: F401
from .mps import (  # noqa: F401
    coefficient, coefficient_batch, parse_config, mps_to_vector, norm,
    canonicalize, compress, lazy_coefficient_batch,
)
from .builders import build_qft_mpo, build_dt_mpo, build_zt_mpo  # noqa: F401
from .encode import signal_mps, signal_ztmps, array_to_tensor  # noqa: F401
from .signals import generate_signal  # noqa: F401
from .analytic import (  # noqa: F401
    dft_unitary, qn_matrix, analytical_dt, analytical_zt, bitrev, int_to_bits,
)
