"""ctypes binding of libqilhip.so (the C ABI declared in include/qilaplace_hip.h).

There is NO CPU fallback: if the HIP library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QILHIP_LIB") or os.path.join(_HERE, "lib", "libqilhip.so")   # same variable as the Julia shim

QIL_F64, QIL_C64 = 0, 1
QIL_METHOD_SVD, QIL_METHOD_RSVD = 0, 1
QIL_DIR_RIGHT, QIL_DIR_LEFT = 0, 1
QIL_MAXDIM_NONE = 2 ** 63 - 1

(QIL_OK, QIL_EINVAL_LENGTH, QIL_EINVAL_SITES, QIL_EINVAL_CONFIG, QIL_EDOMAIN, QIL_ENOMEM, QIL_EHIP,
 QIL_EINVAL_ARG, QIL_EEMPTY) = range(9)


class QilError(RuntimeError):
    """A HIP/runtime failure inside libqilhip (QIL_EHIP / QIL_ENOMEM)."""


class QilDomainError(ArithmeticError):
    """Counterpart of Julia's DomainError (compress! on N < 2, bad canonical centre)."""


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 (same SONAME as the system
    one) and load it by path; if the system copy is already in the process (pulled in by libqilhip.so), torch then
    brings a SECOND runtime and finds no GPUs.  The other order is fine (our NEEDED entry matches torch's copy by
    SONAME), so when torch is installed but not yet imported its copy is loaded first -- without importing torch.
    QIL_SYSTEM_HIP=1 skips this."""
    import sys
    if "torch" in sys.modules or os.environ.get("QIL_SYSTEM_HIP") == "1":
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.origin:
            return
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except (ImportError, OSError, ValueError):
        pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C qilaplace.jl_amd/csrc` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`). "
            "The HIP path is the product; there is no CPU fallback.")
    _share_hip_runtime_with_torch()
    return C.CDLL(LIB_PATH)


lib = _load()

_vp, _i64, _int, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
_pi64, _pint, _pdbl, _pvp = C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_void_p)
_pu8, _u64 = C.POINTER(C.c_uint8), C.c_uint64

# name -> argtypes; every function returns int (qil_status) unless listed in _RET
PROTOTYPES = {
    "qil_device_count": [_pint],
    "qil_context_create": [_int, _vp, _pvp],
    "qil_context_destroy": [_vp],
    "qil_context_synchronize": [_vp],
    "qil_context_trim": [_vp],
    "qil_context_mem_info": [_vp, _pi64, _pi64, _pi64, _pi64],
    "qil_context_fail_alloc_after": [_vp, _i64],
    "qil_context_unowned_bytes": [_vp, _pi64],
    "qil_host_cpu_budget": [_pint],
    "qil_timer_start": [_vp],
    "qil_timer_stop": [_vp, _pdbl],
    "qil_profile_enable": [_vp, _int],
    "qil_profile_read": [_vp, _pi64, _pdbl, _int],
    "qil_mps_create": [_vp, _i64, _int, _int, _pi64, _pi64, _pvp, _dbl, _pvp],
    "qil_mps_alloc": [_vp, _i64, _int, _int, _pi64, _pi64, _dbl, _pvp],
    "qil_mps_destroy": [_vp],
    "qil_mps_clone": [_vp, _pvp],
    "qil_mps_nsites": [_vp, _pi64],
    "qil_mps_dtype": [_vp, _pint],
    "qil_mps_is_paired": [_vp, _pint],
    "qil_mps_bond_dims": [_vp, _pi64],
    "qil_mps_site_ids": [_vp, _pi64],
    "qil_mps_amplitude": [_vp, _pdbl],
    "qil_mps_set_amplitude": [_vp, _dbl],
    "qil_mps_site_nbytes": [_vp, _i64, _pi64],
    "qil_mps_download_site": [_vp, _i64, _vp],
    "qil_mps_upload_site": [_vp, _i64, _vp],
    "qil_mps_site_device_ptr": [_vp, _i64, _pvp],
    "qil_mps_fill_random": [_vp, _u64],
    "qil_mpo_create": [_vp, _i64, _int, _int, _pi64, _pi64, _pvp, _pvp],
    "qil_mpo_alloc": [_vp, _i64, _int, _int, _pi64, _pi64, _pvp],
    "qil_mpo_destroy": [_vp],
    "qil_mpo_nsites": [_vp, _pi64],
    "qil_mpo_dtype": [_vp, _pint],
    "qil_mpo_is_paired": [_vp, _pint],
    "qil_mpo_bond_dims": [_vp, _pi64],
    "qil_mpo_site_ids": [_vp, _pi64],
    "qil_mpo_site_nbytes": [_vp, _i64, _pi64],
    "qil_mpo_download_site": [_vp, _i64, _vp],
    "qil_mpo_site_device_ptr": [_vp, _i64, _pvp],
    "qil_mpo_fill_random": [_vp, _u64],
    "qil_apply": [_vp, _vp, _pvp],
    "qil_apply_into": [_vp, _vp, _vp],
    "qil_apply_mpo_mpo": [_vp, _vp, _pvp],
    "qil_coefficient_batch": [_vp, _i64, _pu8, _pdbl],
    "qil_coefficient_marginal_batch": [_vp, _i64, _pu8, _pdbl],
    "qil_apply_coefficient_batch": [_vp, _vp, _i64, _pu8, _pdbl],
    "qil_apply_coefficient_sweep": [_pvp, _i64, _vp, _i64, _pu8, _pdbl],
    "qil_mps_to_vector": [_vp, _int, _vp],
    "qil_mps_block": [_vp, _pu8, _int, _vp],
    "qil_norm": [_vp, _pdbl],
    "qil_canonicalize": [_vp, _int, _i64, _dbl, _i64],
    "qil_compress": [_vp, _i64, _dbl, _int],
    "qil_mpo_compress": [_vp, _int, _dbl, _i64],
    "qil_compress_batch": [_pvp, _i64, _i64, _dbl, _int],
    "qil_mpo_compress_batch": [_pvp, _i64, _int, _dbl, _i64],
    "qil_apply_compress": [_vp, _vp, _i64, _dbl, _int, _i64, _pvp],
    "qil_apply_compress_batch": [_pvp, _pvp, _i64, _i64, _dbl, _int, _i64, _pvp],
    "qil_signal_mps": [_vp, _vp, _i64, _int, _int, _dbl, _i64, _i64, _i64, _int, _u64, _i64, _pvp],
    "qil_signal_ztmps": [_vp, _vp, _i64, _int, _int, _dbl, _i64, _i64, _i64, _int, _u64, _i64, _pvp],
    "qil_signal_mps_batch": [_vp, _pvp, _i64, _i64, _int, _int, _dbl, _i64, _i64, _i64, _int, _u64, _i64, _pvp],
    "qil_signal_ztmps_batch": [_vp, _pvp, _i64, _i64, _int, _int, _dbl, _i64, _i64, _i64, _int, _u64, _i64, _pvp],
    "qil_rsvd": [_vp, _vp, _i64, _i64, _int, _i64, _i64, _int, _u64, _dbl, _i64, _i64, _pi64, _vp, _pdbl, _vp],
    "qil_build_dt_mpo_batch": [_vp, _i64, _i64, _pdbl, _dbl, _i64, _pi64, _pvp],
    "qil_build_zt_mpo_batch": [_vp, _i64, _i64, _pdbl, _dbl, _i64, _pi64, _pvp],
    "qil_build_qft_mpo": [_vp, _i64, _dbl, _i64, _pi64, _pvp, _pint],
    "qil_build_zt_qft_chain": [_vp, _i64, _dbl, _i64, _pi64, _pvp, _pint],
    "qil_gemm": [_vp, _int, _int, _int, _i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64],
    "qil_qr_positive": [_vp, _int, _i64, _i64, _vp, _vp, _vp],
    "qil_gemm_device_time": [_vp, _int, _int, _int, _i64, _i64, _i64, _int, _pdbl],
    "qil_hbm_store_peak": [_vp, _i64, _int, _pdbl, _pint],
    "qil_svd_trunc": [_vp, _vp, _i64, _i64, _int, _dbl, _i64, _i64, _pi64, _vp, _pdbl, _vp],
    "qil_comm_unique_id": [_vp],
    "qil_comm_create": [_vp, _int, _int, _vp, _pvp],
    "qil_comm_destroy": [_vp],
    "qil_comm_info": [_vp, _pint, _pint],
    "qil_gather_coefficients": [_vp, _i64, _i64, _pdbl, _pdbl],
    "qil_gather_coefficients_device": [_vp, _i64, _i64, _vp, _vp],
    "qil_apply_coefficient_sweep_gather": [_vp, _pvp, _i64, _vp, _i64, _pu8, _i64, _pdbl],
    "qil_sweep_unshuffle_device": [_vp, _int, _i64, _i64, _vp, _vp],
    "qil_sweep_unshuffle": [_int, _i64, _i64, _pdbl, _pdbl],
}
_RET = {"qil_last_error": C.c_char_p, "qil_version": C.c_char_p}

for _name, _args in PROTOTYPES.items():
    _f = getattr(lib, _name)
    _f.argtypes = _args
    _f.restype = C.c_int
for _name, _rt in _RET.items():
    _f = getattr(lib, _name)
    _f.argtypes = []
    _f.restype = _rt


def last_error() -> str:
    return lib.qil_last_error().decode("utf-8", "replace")


def check(status: int):
    """Re-raise a qil_status the way the reference raises (error convention)."""
    if status == QIL_OK:
        return
    msg = last_error()
    if status in (QIL_EINVAL_LENGTH, QIL_EINVAL_SITES, QIL_EINVAL_CONFIG, QIL_EINVAL_ARG, QIL_EEMPTY):
        raise ValueError(msg)                      # Julia: ArgumentError / error()
    if status == QIL_EDOMAIN:
        raise QilDomainError(msg)                  # Julia: DomainError
    if status == QIL_ENOMEM:
        raise MemoryError(msg)
    raise QilError(f"[status {status}] {msg}")
