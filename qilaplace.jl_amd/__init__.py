"""qilaplace.jl_amd -- MI355X-native MPO x MPS apply / coefficient / compress / encode path
of QILaplace.jl, behind the reference's operator names.

The directory name contains a dot, so import it through the root-level shim:

    import qilaplace_jl_amd as qil
    psi = qil.signal_mps(x)                 # encode on the GPU
    out = W * psi                           # fused single-pass site contraction (HIP)
    qil.coefficient(out, "0101...")

Everything numeric runs in lib/libqilhip.so (HIP, gfx950); importing this package fails
loudly if that library has not been built.
"""
from ._lib import QilError, QilDomainError, LIB_PATH, last_error  # noqa: F401
from .containers import (Context, default_context, set_default_context, device_count, host_cpu_budget,  # noqa: F401
                         SignalMPS, ZTMPS, SingleSiteMPO, PairedSiteMPO)
from .ops import (apply, apply_compress, apply_compress_batch, mpo_compress, compress_batch, mpo_compress_batch, mps_block, coefficient, coefficient_batch, apply_coefficient_batch, apply_coefficient_sweep,  # noqa: F401
                  marginal_batch, coefficient_grid, laplace_values,
                  mps_to_vector, norm, canonicalize, compress, signal_mps, signal_ztmps, signal_mps_batch,
                  signal_ztmps_batch, rsvd,
                  svd_trunc, gemm, gemm_device_time, qr_positive)
from .builders import (build_qft_mpo, build_dt_mpo, build_zt_mpo, qft_mpo_tensors,  # noqa: F401
                       dt_mpo_tensors, zt_mpo_tensors, dt_mpo_tensors_many, build_dt_mpo_batch,
                       build_zt_mpo_batch, zt_qft_chain_tensors, qft_mpo_device, zt_qft_chain_device)
from .interchange import save, load  # noqa: F401
from .sweep import shard_items, sweep, damping_sweep, gather_results, damping_sample_bits, Comm, unshuffle  # noqa: F401

__all__ = [
    "Context", "default_context", "set_default_context", "device_count", "host_cpu_budget",
    "SignalMPS", "ZTMPS", "SingleSiteMPO", "PairedSiteMPO",
    "apply", "apply_compress", "apply_compress_batch", "coefficient", "coefficient_batch", "apply_coefficient_batch", "apply_coefficient_sweep", "marginal_batch", "coefficient_grid", "laplace_values", "mps_to_vector", "norm",
    "canonicalize", "compress", "signal_mps", "signal_ztmps", "signal_mps_batch", "signal_ztmps_batch", "rsvd", "svd_trunc", "gemm",
    "build_qft_mpo", "build_dt_mpo", "build_zt_mpo", "qft_mpo_tensors", "dt_mpo_tensors", "zt_mpo_tensors",
    "dt_mpo_tensors_many", "build_dt_mpo_batch", "build_zt_mpo_batch", "zt_qft_chain_tensors", "qft_mpo_device", "zt_qft_chain_device", "mpo_compress", "compress_batch", "mpo_compress_batch", "mps_block",
    "save", "load",
    "shard_items", "sweep", "damping_sweep", "gather_results", "damping_sample_bits", "Comm", "unshuffle",
    "QilError", "QilDomainError",
]
