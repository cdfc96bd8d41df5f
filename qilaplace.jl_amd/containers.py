"""Device-resident mirrors of the reference's containers (host side of the drop-in boundary).

  SignalMPS      src/mps.jl:70-79       ZTMPS          src/mps.jl:98-117
  SingleSiteMPO  src/mpo.jl:26-44       PairedSiteMPO  src/mpo.jl:57-74

Tensors live in HBM behind opaque libqilhip handles; numpy arrays cross the boundary only
in the constructors and the explicit download helpers.  Index order of the numpy views:
MPS site A[alpha, s, beta], MPO site W[a, s_in, s_out, b] (s_in = the reference's primed leg).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

_default_ctx = None


class Context:
    """One device + one HIP stream + a caching device pool (qil_context)."""

    def __init__(self, device: int = 0, stream: int | None = None):
        h = C.c_void_p()
        L.check(L.lib.qil_context_create(int(device), C.c_void_p(stream) if stream else None, C.byref(h)))
        self.handle = h
        self.device = device

    def synchronize(self):
        L.check(L.lib.qil_context_synchronize(self.handle))

    def trim(self):
        L.check(L.lib.qil_context_trim(self.handle))

    def mem_info(self):
        v = [C.c_int64() for _ in range(4)]
        L.check(L.lib.qil_context_mem_info(self.handle, *[C.byref(x) for x in v]))
        return dict(zip(("pool_in_use", "pool_cached", "device_free", "device_total"), (x.value for x in v)))

    def unowned_bytes(self) -> int:
        """Testing aid: pool bytes in use that no MPS/MPO handle owns (0 between calls)."""
        v = C.c_int64()
        L.check(L.lib.qil_context_unowned_bytes(self.handle, C.byref(v)))
        return v.value

    def hbm_store_peak(self, nbytes=8 << 30, reps=5):
        """Diagnostic (qil_hbm_store_peak): this GPU's store-only HBM ceiling in GB/s and the writer that reached it."""
        g, k = C.c_double(), C.c_int()
        L.check(L.lib.qil_hbm_store_peak(self.handle, int(nbytes), int(reps), C.byref(g), C.byref(k)))
        return g.value, ("hipMemsetAsync", "span256KiB_plain", "span256KiB_nt", "grid_stride_plain")[k.value]

    def fail_alloc_after(self, n):
        """Testing aid: make the n-th pool allocation from now fail (None / negative: off)."""
        L.check(L.lib.qil_context_fail_alloc_after(self.handle, -1 if n is None else int(n)))

    def timer_start(self):
        L.check(L.lib.qil_timer_start(self.handle))

    def timer_stop(self) -> float:
        ms = C.c_double()
        L.check(L.lib.qil_timer_stop(self.handle, C.byref(ms)))
        return ms.value

    def profile_enable(self, on=True):
        L.check(L.lib.qil_profile_enable(self.handle, 1 if on else 0))

    def profile_read(self, reset=True):
        n, ms = C.c_int64(), C.c_double()
        L.check(L.lib.qil_profile_read(self.handle, C.byref(n), C.byref(ms), 1 if reset else 0))
        return n.value, ms.value

    def close(self):
        if self.handle:
            L.lib.qil_context_destroy(self.handle)
            self.handle = None


def default_context() -> Context:
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def set_default_context(ctx: Context):
    global _default_ctx
    _default_ctx = ctx


def device_count() -> int:
    n = C.c_int()
    L.check(L.lib.qil_device_count(C.byref(n)))
    return n.value


def host_cpu_budget() -> int:
    """CPUs the batch runners may keep busy: min(cgroup quota, affinity) / LOCAL_WORLD_SIZE (QIL_CPU_BUDGET overrides)."""
    n = C.c_int()
    L.check(L.lib.qil_host_cpu_budget(C.byref(n)))
    return n.value


def _dtype_code(arrs):
    return L.QIL_C64 if any(np.iscomplexobj(a) for a in arrs) else L.QIL_F64


def _np_dtype(code):
    return np.complex128 if code == L.QIL_C64 else np.float64


def _i64arr(vals):
    return (C.c_int64 * max(len(vals), 1))(*[int(v) for v in vals])


class _Chain:
    """Shared plumbing for MPS/MPO handles."""
    _pfx = "qil_mps"
    _rank = 1

    def __init__(self, handle, ctx):
        self.handle = handle
        self.ctx = ctx

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                getattr(L.lib, self._pfx + "_destroy")(self.handle)
                self.handle = None
        except Exception:
            pass

    def _fn(self, name):
        return getattr(L.lib, f"{self._pfx}_{name}")

    def __len__(self):
        n = C.c_int64()
        L.check(self._fn("nsites")(self.handle, C.byref(n)))
        return n.value

    nsite = __len__

    @property
    def dtype(self):
        d = C.c_int()
        L.check(self._fn("dtype")(self.handle, C.byref(d)))
        return _np_dtype(d.value)

    @property
    def paired(self) -> bool:
        p = C.c_int()
        L.check(self._fn("is_paired")(self.handle, C.byref(p)))
        return bool(p.value)

    @property
    def bond_dims(self):
        n = len(self)
        b = (C.c_int64 * max(n - 1, 1))()
        L.check(self._fn("bond_dims")(self.handle, b))
        return [int(b[i]) for i in range(n - 1)]

    @property
    def site_ids(self):
        n = len(self)
        s = (C.c_int64 * n)()
        L.check(self._fn("site_ids")(self.handle, s))
        return [int(v) for v in s]

    def site_shape(self, i):
        d = [1] + self.bond_dims + [1]
        return (d[i], 2, d[i + 1]) if self._rank == 1 else (d[i], 2, 2, d[i + 1])

    def site(self, i) -> np.ndarray:
        """Download site tensor i (0-based) as a numpy array in index order."""
        shp = self.site_shape(i)
        out = np.empty(shp, dtype=self.dtype, order="F")
        L.check(self._fn("download_site")(self.handle, int(i), out.ctypes.data_as(C.c_void_p)))
        return out

    def to_host(self):
        return [self.site(i) for i in range(len(self))]

    def site_device_ptr(self, i) -> int:
        p = C.c_void_p()
        L.check(self._fn("site_device_ptr")(self.handle, int(i), C.byref(p)))
        return p.value

    def fill_random(self, seed: int):
        L.check(self._fn("fill_random")(self.handle, C.c_uint64(seed)))
        return self


def _create(pfx, ctx, data, paired, site_ids, amplitude=None):
    data = list(data)
    n = len(data)
    code = _dtype_code(data)
    npdt = _np_dtype(code)
    host = [np.asfortranarray(np.asarray(t, dtype=npdt)) for t in data]
    rank = host[0].ndim - 2
    for i, t in enumerate(host):
        if t.ndim != rank + 2 or any(d != 2 for d in t.shape[1:-1]):
            raise ValueError(f"site {i+1}: bad tensor shape {t.shape}")
    if host[0].shape[0] != 1 or host[-1].shape[-1] != 1:
        raise ValueError("edge bonds must have dimension 1")
    for i in range(n - 1):
        if host[i].shape[-1] != host[i + 1].shape[0]:
            raise ValueError(f"bond {i+1}: dims disagree ({host[i].shape[-1]} vs {host[i+1].shape[0]})")
    bonds = _i64arr([t.shape[-1] for t in host[:-1]])
    ids = _i64arr(site_ids) if site_ids is not None else None
    ptrs = (C.c_void_p * n)(*[t.ctypes.data for t in host])
    h = C.c_void_p()
    if pfx == "qil_mps":
        L.check(L.lib.qil_mps_create(ctx.handle, n, code, int(paired), bonds, ids, ptrs,
                                     float(amplitude), C.byref(h)))
    else:
        L.check(L.lib.qil_mpo_create(ctx.handle, n, code, int(paired), bonds, ids, ptrs, C.byref(h)))
    return h


class SignalMPS(_Chain):
    """n-site MPS on the device; ``amplitude`` = ||x||_2 of the encoded signal (src/mps.jl:70-79)."""
    _pfx, _rank = "qil_mps", 1

    def __init__(self, data=None, sites=None, amplitude=1.0, ctx=None, _handle=None):
        ctx = ctx or default_context()
        if _handle is None:
            _handle = _create("qil_mps", ctx, data, self._paired(), sites, amplitude)
        super().__init__(_handle, ctx)

    @staticmethod
    def _paired():
        return False

    @classmethod
    def alloc(cls, bond_dims, dtype=np.float64, sites=None, amplitude=1.0, ctx=None):
        """Uninitialised device tensors with the given internal bond dims."""
        ctx = ctx or default_context()
        n = len(bond_dims) + 1
        code = L.QIL_C64 if np.dtype(dtype) == np.complex128 else L.QIL_F64
        h = C.c_void_p()
        L.check(L.lib.qil_mps_alloc(ctx.handle, n, code, int(cls._paired()), _i64arr(bond_dims),
                                    _i64arr(sites) if sites is not None else None, float(amplitude),
                                    C.byref(h)))
        return cls(ctx=ctx, _handle=h)

    @property
    def amplitude(self) -> float:
        a = C.c_double()
        L.check(L.lib.qil_mps_amplitude(self.handle, C.byref(a)))
        return a.value

    @amplitude.setter
    def amplitude(self, v):
        L.check(L.lib.qil_mps_set_amplitude(self.handle, float(v)))

    def copy(self):
        h = C.c_void_p()
        L.check(L.lib.qil_mps_clone(self.handle, C.byref(h)))
        return type(self)(ctx=self.ctx, _handle=h)

    def __getitem__(self, bits):
        """psi[b1, b2, ...] == coefficient(psi, (b1, b2, ...))  (src/mps.jl:692-693)."""
        from .ops import coefficient
        return coefficient(self, list(bits) if isinstance(bits, tuple) else [bits])


class ZTMPS(SignalMPS):
    """Paired-register MPS (src/mps.jl:98-117), held as its interleaved 2n-tensor chain
    main_1, copy_1, main_2, ... (src/mps.jl:421-444)."""

    @staticmethod
    def _paired():
        return True

    def __len__(self):
        return super().__len__() // 2

    @property
    def ntensors(self):
        return super().__len__()

    @property
    def bond_dims(self):
        n = self.ntensors
        b = (C.c_int64 * max(n - 1, 1))()
        L.check(L.lib.qil_mps_bond_dims(self.handle, b))
        return [int(b[i]) for i in range(n - 1)]

    @property
    def site_ids(self):
        n = self.ntensors
        s = (C.c_int64 * n)()
        L.check(L.lib.qil_mps_site_ids(self.handle, s))
        return [int(v) for v in s]

    @property
    def bonds_copy(self):
        return self.bond_dims[0::2]

    @property
    def bonds_main(self):
        return self.bond_dims[1::2]

    def to_host(self):
        return [self.site(i) for i in range(self.ntensors)]


class SingleSiteMPO(_Chain):
    """n-site MPO on the device (src/mpo.jl:26-44)."""
    _pfx, _rank = "qil_mpo", 2

    def __init__(self, data=None, sites=None, ctx=None, _handle=None):
        ctx = ctx or default_context()
        if _handle is None:
            _handle = _create("qil_mpo", ctx, data, self._paired(), sites)
        super().__init__(_handle, ctx)

    @staticmethod
    def _paired():
        return False

    @classmethod
    def alloc(cls, bond_dims, dtype=np.complex128, sites=None, ctx=None):
        ctx = ctx or default_context()
        n = len(bond_dims) + 1
        code = L.QIL_C64 if np.dtype(dtype) == np.complex128 else L.QIL_F64
        h = C.c_void_p()
        L.check(L.lib.qil_mpo_alloc(ctx.handle, n, code, int(cls._paired()), _i64arr(bond_dims),
                                    _i64arr(sites) if sites is not None else None, C.byref(h)))
        return cls(ctx=ctx, _handle=h)

    @classmethod
    def identity(cls, n, sites=None, ctx=None):
        """SingleSiteMPO(n) / PairedSiteMPO(n) identity constructors (src/mpo.jl:77-147)."""
        eye = np.eye(2).reshape(1, 2, 2, 1)
        m = 2 * n if cls._paired() else n
        return cls([eye] * m, sites=sites, ctx=ctx)


class PairedSiteMPO(SingleSiteMPO):
    """2n-tensor MPO alternating main/copy sites (src/mpo.jl:57-74)."""

    @staticmethod
    def _paired():
        return True

    def __len__(self):
        return super().__len__() // 2

    @property
    def ntensors(self):
        return SingleSiteMPO.__len__(self)

    @property
    def bond_dims(self):
        n = self.ntensors
        b = (C.c_int64 * max(n - 1, 1))()
        L.check(L.lib.qil_mpo_bond_dims(self.handle, b))
        return [int(b[i]) for i in range(n - 1)]

    @property
    def site_ids(self):
        n = self.ntensors
        s = (C.c_int64 * n)()
        L.check(L.lib.qil_mpo_site_ids(self.handle, s))
        return [int(v) for v in s]

    def to_host(self):
        return [self.site(i) for i in range(self.ntensors)]
