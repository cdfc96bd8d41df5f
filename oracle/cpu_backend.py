"""ctypes loader of oracle/cpu/lib/libqilcpu.so, the C++/OpenMP CPU implementation of the apply path behind the same C
ABI as the HIP library (CPU baseline (1)).  TEST / BASELINE INFRASTRUCTURE: used by bench.py's
`cpu_baseline` leg and tests/test_cpu_backend.py only."""
from __future__ import annotations

import ctypes as C
import time

import numpy as np

_vp, _i64, _int, _dbl = C.c_void_p, C.c_int64, C.c_int, C.c_double
_pi64, _pvp = C.POINTER(C.c_int64), C.POINTER(C.c_void_p)


class CpuBackend:
    def __init__(self, path):
        self.lib = lib = C.CDLL(path)
        lib.qil_last_error.restype = C.c_char_p
        lib.qil_version.restype = C.c_char_p
        lib.qil_context_create.argtypes = [_int, _vp, _pvp]
        lib.qilcpu_set_threads.argtypes = [_vp, _int]
        lib.qil_mps_create.argtypes = [_vp, _i64, _int, _int, _pi64, _pi64, _pvp, _dbl, _pvp]
        lib.qil_mpo_create.argtypes = [_vp, _i64, _int, _int, _pi64, _pi64, _pvp, _pvp]
        lib.qil_apply.argtypes = [_vp, _vp, _pvp]
        lib.qilcpu_apply_site.argtypes = [_vp, _vp, _i64, _vp]
        lib.qilcpu_first_touch.argtypes = [_vp, _vp, _i64]
        lib.qil_mps_destroy.argtypes = [_vp]
        lib.qil_mpo_destroy.argtypes = [_vp]
        lib.qil_mps_bond_dims.argtypes = [_vp, _pi64]
        lib.qil_mps_download_site.argtypes = [_vp, _i64, _vp]
        lib.qil_coefficient_batch.argtypes = [_vp, _i64, C.POINTER(C.c_uint8), C.POINTER(C.c_double)]
        h = _vp()
        self._check(lib.qil_context_create(0, None, C.byref(h)))
        self.ctx = h
        self.threads = self.set_threads(0)

    def _check(self, st):
        if st != 0:
            raise ValueError(self.lib.qil_last_error().decode())

    def set_threads(self, n):
        self.threads = int(self.lib.qilcpu_set_threads(self.ctx, int(n)))
        return self.threads

    def _chain(self, tensors, mpo, amplitude=1.0):
        ts = [np.asfortranarray(t) for t in tensors]
        dt = np.result_type(*[t.dtype for t in ts])
        ts = [np.asfortranarray(t, dtype=dt) for t in ts]
        n = len(ts)
        bonds = (C.c_int64 * max(n - 1, 1))(*[t.shape[-1] for t in ts[:-1]])
        ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in ts])
        h = _vp()
        code = 1 if dt == np.complex128 else 0
        if mpo:
            self._check(self.lib.qil_mpo_create(self.ctx, n, code, 0, bonds, None, ptrs, C.byref(h)))
        else:
            self._check(self.lib.qil_mps_create(self.ctx, n, code, 0, bonds, None, ptrs, float(amplitude), C.byref(h)))
        return h, ts, dt

    def apply(self, w, a):
        """apply(W, psi) -> list of site tensors (numpy, canonical layout)."""
        hw, tw, dw = self._chain(w, True)
        ha, ta, da = self._chain(a, False)
        out = _vp()
        try:
            self._check(self.lib.qil_apply(hw, ha, C.byref(out)))
            odt = np.result_type(dw, da)
            res = []
            for i in range(len(ta)):
                shape = (tw[i].shape[0] * ta[i].shape[0], 2, tw[i].shape[3] * ta[i].shape[2])
                buf = np.empty(shape, dtype=odt, order="F")
                self.lib.qil_mps_download_site(out, i, buf.ctypes.data)
                res.append(buf)
            return res
        finally:
            if out:
                self.lib.qil_mps_destroy(out)
            self.lib.qil_mpo_destroy(hw)
            self.lib.qil_mps_destroy(ha)

    def apply_coefficients(self, w, a, bits, amplitude=1.0):
        hw, tw, dw = self._chain(w, True)
        ha, ta, da = self._chain(a, False, amplitude)
        out = _vp()
        try:
            self._check(self.lib.qil_apply(hw, ha, C.byref(out)))
            b = np.ascontiguousarray(bits, dtype=np.uint8)
            res = np.zeros(b.shape[0], dtype=np.complex128)
            self._check(self.lib.qil_coefficient_batch(out, b.shape[0], b.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                       res.ctypes.data_as(C.POINTER(C.c_double))))
            return res
        finally:
            if out:
                self.lib.qil_mps_destroy(out)
            self.lib.qil_mpo_destroy(hw)
            self.lib.qil_mps_destroy(ha)

    def time_apply(self, w, a, threads, reps=1):
        """Seconds for one apply over ALL sites, each site written into one reusable buffer of the largest site's
        size (so the 80 GB result of the metric configuration never has to exist at once).  threads: an int, or a
        list of team sizes to try on the largest site first (the fastest is used: `nproc` may exceed the cores a
        container really gets).  The buffer is first-touched by the team outside the timed region."""
        hw, tw, dw = self._chain(w, True)
        ha, ta, da = self._chain(a, False)
        odt = np.result_type(dw, da)
        esz = 16 if odt == np.complex128 else 8
        sizes = [tw[i].shape[0] * ta[i].shape[0] * 2 * tw[i].shape[3] * ta[i].shape[2] * esz for i in range(len(ta))]
        buf = np.empty(max(sizes) // 8, dtype=np.float64)
        big = int(np.argmax(sizes))
        try:
            cands = list(threads) if isinstance(threads, (list, tuple)) else [threads]
            best = None
            for th in cands:
                used = self.set_threads(th)
                self.lib.qilcpu_first_touch(self.ctx, buf.ctypes.data, buf.nbytes)
                self._check(self.lib.qilcpu_apply_site(hw, ha, big, buf.ctypes.data))      # dry run
                if len(cands) > 1:
                    t0 = time.perf_counter()
                    self._check(self.lib.qilcpu_apply_site(hw, ha, big, buf.ctypes.data))
                    dt = time.perf_counter() - t0
                    if best is None or dt < best[0]:
                        best = (dt, th)
            if best is not None:
                used = self.set_threads(best[1])
                self.lib.qilcpu_first_touch(self.ctx, buf.ctypes.data, buf.nbytes)
                self._check(self.lib.qilcpu_apply_site(hw, ha, big, buf.ctypes.data))
            t0 = time.perf_counter()
            for _ in range(reps):
                for i in range(len(ta)):
                    self._check(self.lib.qilcpu_apply_site(hw, ha, i, buf.ctypes.data))
            return (time.perf_counter() - t0) / reps, used, sum(sizes)
        finally:
            self.lib.qil_mpo_destroy(hw)
            self.lib.qil_mps_destroy(ha)
