"""Host-side operator API mirroring the reference's method names (drop-in boundary):

  apply / `*`        src/linalg/apply.jl:75-122, 124-199, 201-230, 233-236
  coefficient        src/mps.jl:669-693 (parsers :616-645)
  mps_to_vector      src/mps.jl:716-743
  norm               src/mps.jl:754-771
  canonicalize       src/mps.jl:787-847, 866-901   (Julia: canonicalize!)
  compress           src/mps.jl:913-999            (Julia: compress!)
  signal_mps         src/signals/SignalConverters.jl:228-233
  signal_ztmps       src/signals/SignalConverters.jl:247-283
  rsvd               src/linalg/rsvd.jl:38-121

All arithmetic happens in libqilhip.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
import re
import warnings

import numpy as np

from . import _lib as L
from .containers import (SignalMPS, ZTMPS, SingleSiteMPO, PairedSiteMPO, default_context, _np_dtype)


def _wrap_like(psi, handle):
    return type(psi)(ctx=psi.ctx, _handle=handle)


# ---------------------------------------------------------------- apply
def apply(W, psi, out=None, **kwargs):
    """apply(W, psi; kwargs...) -> psi_out.  ``cutoff``/``maxdim`` kwargs are accepted and
    ignored, exactly like the reference (apply.jl:75): apply never truncates."""
    if isinstance(W, SingleSiteMPO) and isinstance(psi, SingleSiteMPO):
        if W.paired != psi.paired:
            raise TypeError("apply: cannot mix SingleSiteMPO and PairedSiteMPO")
        h = C.c_void_p()
        L.check(L.lib.qil_apply_mpo_mpo(W.handle, psi.handle, C.byref(h)))
        return type(W)(ctx=W.ctx, _handle=h)
    if not (isinstance(W, SingleSiteMPO) and isinstance(psi, SignalMPS)):
        raise TypeError("apply: unsupported operand types")
    if W.paired != psi.paired:
        raise TypeError("apply: PairedSiteMPO acts on ZTMPS, SingleSiteMPO on SignalMPS")
    if out is not None:
        L.check(L.lib.qil_apply_into(W.handle, psi.handle, out.handle))
        return out
    h = C.c_void_p()
    L.check(L.lib.qil_apply(W.handle, psi.handle, C.byref(h)))
    return _wrap_like(psi, h)


def _mul(self, other):
    return apply(self, other)


SingleSiteMPO.__mul__ = _mul          # W * psi, W1 * W2  (apply.jl:233-236)


# ---------------------------------------------------------------- coefficient
def _parse_config(spec, n):
    """Front-ends of `coefficient` (src/mps.jl:616-645, 680-693)."""
    if isinstance(spec, str):
        s = spec.strip().strip("[](){}").strip()
        if not s:
            raise ValueError("coefficient: configuration string is empty")
        if re.search(r"[,\s]", s):
            toks = [t for t in re.split(r"[,\s]+", s) if t]
            if not toks:
                raise ValueError("coefficient: configuration string did not contain any entries")
            return [int(t) for t in toks]
        if any(c not in "01" for c in s):
            raise ValueError("coefficient: bit strings may contain only '0' or '1'")
        return [1 if c == "1" else 0 for c in s]
    if isinstance(spec, (int, np.integer)) and not isinstance(spec, bool):
        v = int(spec)
        if v < 0:
            raise ValueError("coefficient: integer configuration must be non-negative")
        if v >> n:
            raise ValueError(f"coefficient: integer {v} requires more than {n} bits")
        return [(v >> (n - 1 - i)) & 1 for i in range(n)]
    return [int(b) for b in spec]


def _ntensors(psi):
    return psi.ntensors if isinstance(psi, (ZTMPS, PairedSiteMPO)) else len(psi)


def _bits_array(psi, bits, max_bit=1):
    n = _ntensors(psi)
    b = np.asarray(bits)
    if b.ndim != 2 or b.shape[1] != n:
        got = b.shape[1] if b.ndim == 2 else b.shape
        raise ValueError(f"coefficient: expected {n} entries, got {got}")
    if b.size and (b.min() < 0 or b.max() > max_bit):
        bad = int(b[(b < 0) | (b > max_bit)][0])
        raise ValueError(f"coefficient: bit value {bad} outside [0,{max_bit}]")
    return np.ascontiguousarray(b, dtype=np.uint8)


def coefficient_batch(psi, bits):
    """Vectorised `coefficient`: bits is (nb, n_tensors) of {0,1}; returns (nb,) values
    (complex for complex MPS, real otherwise), each amplitude * prod_i A_i[:, bit_i, :]."""
    b = _bits_array(psi, bits)
    nb = b.shape[0]
    out = np.zeros(nb, dtype=np.complex128)
    L.check(L.lib.qil_coefficient_batch(psi.handle, nb, b.ctypes.data_as(C.POINTER(C.c_uint8)),
                                        out.ctypes.data_as(C.POINTER(C.c_double))))
    return out if psi.dtype == np.complex128 else out.real.copy()


def marginal_batch(psi, bits):
    """Like coefficient_batch, but a bit value of 2 SUMS that site's physical index (marginal).  One
    chain then replaces 2^m coefficient calls when m sites are summed."""
    b = _bits_array(psi, bits, max_bit=2)
    nb = b.shape[0]
    out = np.zeros(nb, dtype=np.complex128)
    L.check(L.lib.qil_coefficient_marginal_batch(psi.handle, nb, b.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                 out.ctypes.data_as(C.POINTER(C.c_double))))
    return out if psi.dtype == np.complex128 else out.real.copy()


def _lsb_bits(vals, n):
    v = np.asarray(vals, dtype=np.int64)
    return ((v[:, None] >> np.arange(n)[None, :]) & 1).astype(np.uint8)


def _msb_bits(vals, n):
    v = np.asarray(vals, dtype=np.int64)
    return ((v[:, None] >> np.arange(n - 1, -1, -1)[None, :]) & 1).astype(np.uint8)


def coefficient_grid(psi, ks, ls, chunk=1 << 16):
    """chi[k, l] = coefficient(psi, interleave(lsb(k), lsb(l))) for a transformed ZTMPS (the (k, l) scans
    of docs/src/tutorials/zt.jl:152-157, 283-309) -- all pairs in batched launches."""
    n = len(psi)
    ks, ls = np.asarray(ks, dtype=np.int64), np.asarray(ls, dtype=np.int64)
    rk, rl = _bit_block_range(ks), _bit_block_range(ls)
    if rk is not None and rl is not None and rk[0] + rk[1] <= n and rl[0] + rl[1] <= n and rk[1] + rl[1] <= 30:
        # a 2^a x 2^b grid of aligned power-of-two strides (the tutorials' full and coarse scans) = every
        # configuration of a bit block of each register with the other bits fixed at 0: one dense block read-out
        # instead of 2^(a+b) chains
        (sk, a), (sl, b) = rk, rl
        spec = np.zeros(2 * n, dtype=np.uint8)
        spec[2 * sk:2 * (sk + a):2] = FREE
        spec[2 * sl + 1:2 * (sl + b) + 1:2] = FREE
        t = np.asarray(mps_block(psi, spec, reverse=True)).reshape([2] * (a + b))   # axis 0 = LAST free site
        free_sites = sorted([2 * (sk + i) for i in range(a)] + [2 * (sl + i) + 1 for i in range(b)])
        axis_of = {site: (a + b - 1 - pos) for pos, site in enumerate(free_sites)}
        order = ([axis_of[2 * (sk + i)] for i in range(a - 1, -1, -1)] +
                 [axis_of[2 * (sl + i) + 1] for i in range(b - 1, -1, -1)])
        return np.ascontiguousarray(t.transpose(order)).reshape(2 ** a, 2 ** b).astype(np.complex128)
    kb, lb = _lsb_bits(ks, n), _lsb_bits(ls, n)
    out = np.empty((len(ks), len(ls)), dtype=np.complex128)
    rows = max(1, chunk // max(len(ls), 1))
    for r0 in range(0, len(ks), rows):
        kk = kb[r0:r0 + rows]
        bits = np.empty((len(kk), len(ls), 2 * n), dtype=np.uint8)
        bits[:, :, 0::2] = kk[:, None, :]
        bits[:, :, 1::2] = lb[None, :, :]
        out[r0:r0 + rows] = np.asarray(coefficient_batch(psi, bits.reshape(-1, 2 * n))).reshape(len(kk), len(ls))
    return out


def laplace_values(psi_out, ks, dt):
    """L(s_k) = dt sqrt(N) sum_j coefficient(psi_out, interleave(lsb(k), msb(j))) for a damping-transformed
    ZTMPS (docs/src/tutorials/dt.jl:172-197).  The sum over the copy register is a marginal: one chain per
    k instead of N coefficient calls."""
    n = len(psi_out)
    a = _full_low_range(ks)
    if a is not None and a <= min(n, 30):
        # all of k = 0 .. 2^a - 1: main bits free (lsb first), copy register summed -- one dense contraction
        spec = np.full(2 * n, SUM, dtype=np.uint8)
        spec[0::2] = FIX0
        spec[0:2 * a:2] = FREE
        return dt * np.sqrt(2.0 ** n) * mps_block(psi_out, spec, reverse=True)
    kb = _lsb_bits(ks, n)
    bits = np.full((len(kb), 2 * n), 2, dtype=np.uint8)
    bits[:, 0::2] = kb
    return dt * np.sqrt(2.0 ** n) * marginal_batch(psi_out, bits)


def coefficient(psi, config):
    """coefficient(psi, config): config is a list/tuple of bits, a bit string ("101" or
    "[1,0,1]") or a non-negative integer read as an n-bit big-endian pattern."""
    n = _ntensors(psi)
    bits = _parse_config(config, n)
    if len(bits) != n:
        raise ValueError(f"coefficient: expected {n} entries, got {len(bits)}")
    return coefficient_batch(psi, [bits])[0]


def apply_coefficient_batch(W, psi, bits):
    """<bits| W psi> without materialising W*psi (lazy path; same numbers as
    coefficient_batch(apply(W, psi), bits))."""
    b = _bits_array(psi, bits)
    nb = b.shape[0]
    out = np.zeros(nb, dtype=np.complex128)
    L.check(L.lib.qil_apply_coefficient_batch(W.handle, psi.handle, nb,
                                              b.ctypes.data_as(C.POINTER(C.c_uint8)),
                                              out.ctypes.data_as(C.POINTER(C.c_double))))
    if psi.dtype == np.complex128 or W.dtype == np.complex128:
        return out
    return out.real.copy()


def apply_coefficient_sweep(Ws, psi, bits, comm=None, n_items=None):
    """For every operator of `Ws` (e.g. the DT MPOs of a damping sweep): materialise W * psi with the apply kernel
    and read it out at the same configurations -- the loop `out = W * psi; coefficient(out, bits)` of the
    reference's sweeps (docs/src/tutorials/dt.jl:150-197) with one upload, one download and one synchronisation
    for the whole batch.  Returns a (len(Ws), nb) complex array.

    With `comm` (a `sweep.Comm`) the call is COLLECTIVE: `Ws` is this rank's round-robin share of `n_items` operators, the
    samples stay in HBM, one all-gather exchanges them (qil_apply_coefficient_sweep_gather) and every rank gets the
    (n_items, nb) table in item order."""
    Ws = list(Ws)
    b = _bits_array(psi, bits)
    nb = b.shape[0]
    if comm is not None:
        if n_items is None:
            raise ValueError("apply_coefficient_sweep: n_items is required with a communicator")
        out = np.zeros((int(n_items), nb), dtype=np.complex128)
        hs = (C.c_void_p * max(len(Ws), 1))(*[W.handle for W in Ws])
        L.check(L.lib.qil_apply_coefficient_sweep_gather(comm.handle, hs, len(Ws), psi.handle, nb, b.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                         int(n_items), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out
    out = np.zeros((len(Ws), nb), dtype=np.complex128)
    if not Ws or nb == 0:
        return out
    hs = (C.c_void_p * len(Ws))(*[W.handle for W in Ws])
    L.check(L.lib.qil_apply_coefficient_sweep(hs, len(Ws), psi.handle, nb, b.ctypes.data_as(C.POINTER(C.c_uint8)),
                                              out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


# ---------------------------------------------------------------- dense read-out, norm
def mps_to_vector(psi, reverse=False):
    n = _ntensors(psi)
    out = np.empty(2 ** n, dtype=psi.dtype)
    L.check(L.lib.qil_mps_to_vector(psi.handle, 1 if reverse else 0, out.ctypes.data_as(C.c_void_p)))
    return out


FIX0, FIX1, SUM, FREE = 0, 1, 2, 3


def mps_block(psi, spec, reverse=False):
    """All 2^F coefficients of the configurations that agree with `spec` on its fixed sites, as one dense
    contraction: spec[i] = 0 / 1 fixes site i's bit, 2 (SUM) sums the site, 3 (FREE) leaves it free.  The result is
    indexed by the free sites in chain order, the first one the most significant bit (reverse=False, like
    mps_to_vector) or the least (reverse=True)."""
    n = _ntensors(psi)
    sp = np.ascontiguousarray(np.asarray(spec), dtype=np.uint8)
    if sp.shape != (n,):
        raise ValueError(f"coefficient: expected {n} entries, got {sp.shape}")
    if sp.size and sp.max() > 3:
        raise ValueError(f"coefficient: spec value {int(sp.max())} outside [0,3]")
    out = np.empty(2 ** int((sp == FREE).sum()), dtype=psi.dtype)
    L.check(L.lib.qil_mps_block(psi.handle, sp.ctypes.data_as(C.POINTER(C.c_uint8)), 1 if reverse else 0,
                                out.ctypes.data_as(C.c_void_p)))
    return out


def _bit_block_range(v):
    """(s, a) if v == arange(2^a) << s -- every pattern of bits s .. s+a-1, all other bits zero -- else None."""
    v = np.asarray(v, dtype=np.int64)
    a = int(round(np.log2(len(v)))) if len(v) else -1
    if a < 0 or len(v) != 2 ** a:
        return None
    if a == 0:
        return (0, 0) if v[0] == 0 else None
    step = int(v[1] - v[0])
    if step <= 0 or step & (step - 1) or not np.array_equal(v, step * np.arange(2 ** a, dtype=np.int64)):
        return None
    return step.bit_length() - 1, a


def _full_low_range(v):
    """log2(len) if v == arange(2^a), else None."""
    r = _bit_block_range(v)
    return r[1] if r is not None and (r[0] == 0 or r[1] == 0) else None


def norm(psi) -> float:
    v = C.c_double()
    L.check(L.lib.qil_norm(psi.handle, C.byref(v)))
    return v.value


# ---------------------------------------------------------------- truncation
def _maxdim(m):
    return L.QIL_MAXDIM_NONE if m is None else int(m)


def canonicalize(psi, direction, center=None, cutoff=1e-12, maxdim=None):
    """canonicalize!(psi, direction; center, cutoff, maxdim) -- in place, returns psi."""
    if direction not in ("right", "left"):
        raise ValueError("Direction must be :right or :left")
    L.check(L.lib.qil_canonicalize(psi.handle, L.QIL_DIR_RIGHT if direction == "right" else L.QIL_DIR_LEFT,
                                   0 if center is None else int(center), float(cutoff), _maxdim(maxdim)))
    return psi


def compress(psi, maxdim=None, tol=1e-12, sweeps=1):
    """compress!(psi; maxdim, tol, sweeps) -- in place, returns psi."""
    L.check(L.lib.qil_compress(psi.handle, _maxdim(maxdim), float(tol), int(sweeps)))
    return psi


def mpo_compress(W, direction="down", cutoff=1e-14, maxdim=None):
    """zip_to_compress_mpo(W, direction; cutoff, maxdim) over the whole MPO (dt_transformer.jl:167-288) -- in
    place, returns W.  "down": exact gauge sweep left -> right, truncating sweep right -> left; "up": mirror."""
    if direction not in ("down", "up"):
        raise ValueError(f"zip_to_compress_mpo: unknown direction '{direction}'")
    L.check(L.lib.qil_mpo_compress(W.handle, 0 if direction == "down" else 1, float(cutoff), _maxdim(maxdim)))
    return W


def apply_compress_batch(Ws, psis, maxdim=None, tol=1e-12, sweeps=1, zip_maxdim=None):
    """apply_compress(W, psi) for every (W, psi) pair -- the (signal, damping value) items of a sweep -- concurrently on the
    context's streams.  `Ws` / `psis` may each be a single operand (used for every item) or a sequence; returns the list
    of results."""
    if hasattr(Ws, "handle"):
        Ws = [Ws] * (len(psis) if not hasattr(psis, "handle") else 1)
    if hasattr(psis, "handle"):
        psis = [psis] * len(Ws)
    Ws, psis = list(Ws), list(psis)
    if len(Ws) != len(psis):
        raise ValueError(f"apply_compress_batch: {len(Ws)} operators for {len(psis)} states")
    for W, psi in zip(Ws, psis):
        if W.paired != psi.paired:
            raise TypeError("apply: PairedSiteMPO acts on ZTMPS, SingleSiteMPO on SignalMPS")
    nb = len(Ws)
    _, wa = _handle_array(Ws)
    _, pa = _handle_array(psis)
    outs = (C.c_void_p * max(nb, 1))()
    L.check(L.lib.qil_apply_compress_batch(wa, pa, nb, _maxdim(maxdim), float(tol), int(sweeps),
                                           0 if zip_maxdim is None else int(zip_maxdim), outs))
    return [_wrap_like(psi, C.c_void_p(h)) for psi, h in zip(psis, outs[:nb])]


def _handle_array(items):
    items = list(items)
    arr = (C.c_void_p * max(len(items), 1))(*[it.handle.value if isinstance(it.handle, C.c_void_p) else it.handle
                                              for it in items])
    return items, arr


def compress_batch(psis, maxdim=None, tol=1e-12, sweeps=1):
    """compress!(psi; maxdim, tol, sweeps) for every MPS of `psis` (independent chains of one context, e.g. the signals
    of a sweep) -- in place, concurrently on the context's worker streams; returns the list."""
    items, arr = _handle_array(psis)
    L.check(L.lib.qil_compress_batch(arr, len(items), _maxdim(maxdim), float(tol), int(sweeps)))
    return items


def mpo_compress_batch(Ws, direction="down", cutoff=1e-14, maxdim=None):
    """zip_to_compress_mpo(W, direction; cutoff, maxdim) for every MPO of `Ws` (one per damping value of a sweep) -- in
    place, concurrently on the context's worker streams; returns the list."""
    if direction not in ("down", "up"):
        raise ValueError(f"zip_to_compress_mpo: unknown direction '{direction}'")
    items, arr = _handle_array(Ws)
    L.check(L.lib.qil_mpo_compress_batch(arr, len(items), 0 if direction == "down" else 1, float(cutoff), _maxdim(maxdim)))
    return items


def apply_compress(W, psi, maxdim=None, tol=1e-12, sweeps=1, zip_maxdim=None):
    """compress(apply(W, psi), maxdim, tol, sweeps) fused: a zip-up sweep that never writes the (D chi)^2
    product tensors, then the exact-gauge compress.  (The reference's `apply` ignores cutoff/maxdim, and so
    does `apply` here; this is the explicit truncating variant.)  As for every zip-up, the intermediate
    truncations are near-optimal for decaying spectra and can lose more than `compress(apply(W, psi))` -- the
    exact route -- on flat-spectrum (random) operands; `zip_maxdim` buys head-room."""
    if W.paired != psi.paired:
        raise TypeError("apply: PairedSiteMPO acts on ZTMPS, SingleSiteMPO on SignalMPS")
    h = C.c_void_p()
    L.check(L.lib.qil_apply_compress(W.handle, psi.handle, _maxdim(maxdim), float(tol), int(sweeps),
                                     0 if zip_maxdim is None else int(zip_maxdim), C.byref(h)))
    return _wrap_like(psi, h)


# ---------------------------------------------------------------- encode
def _encode(fn, cls, x, method, cutoff, maxdim, k, p, q, random_seed, mindim, ctx):
    if method not in ("svd", "rsvd"):
        raise ValueError(f"tensor_to_mps: unknown method {method}. Use :svd or :rsvd.")
    ctx = ctx or default_context()
    cai = getattr(x, "__cuda_array_interface__", None)
    if cai is not None:
        # samples already in HBM (a torch / cupy-style device array): hand the device pointer over, no PCIe
        # trip.  Must be 1-D, contiguous, float64 or complex128; the producer's stream is drained first.
        if len(cai["shape"]) != 1 or cai.get("strides") not in (None, (np.dtype(cai["typestr"]).itemsize,)):
            raise ValueError("signal_mps: device signal must be a contiguous 1-D array")
        dt = np.dtype(cai["typestr"])
        if dt not in (np.dtype(np.float64), np.dtype(np.complex128)):
            raise ValueError(f"signal_mps: device signal must be float64 or complex128, got {dt}")
        code = L.QIL_C64 if dt == np.dtype(np.complex128) else L.QIL_F64
        N = int(cai["shape"][0])
        ptr = C.c_void_p(int(cai["data"][0]))
        sync = getattr(x, "device", None)
        if sync is not None and "torch" in type(x).__module__:
            import torch
            torch.cuda.current_stream(x.device).synchronize()
        keep = x
    else:
        x = np.asarray(x)
        code = L.QIL_C64 if np.iscomplexobj(x) else L.QIL_F64
        keep = np.ascontiguousarray(x, dtype=_np_dtype(code))
        N = len(keep)
        ptr = keep.ctypes.data_as(C.c_void_p)
    n = max(1, int(round(np.log2(max(N, 1)))))
    if N < 2 ** n:
        warnings.warn(f"_array_to_tensor: input length {N} is not a power of 2; zero-filling to {2**n}")
    h = C.c_void_p()
    L.check(fn(ctx.handle, ptr, N, code,
               L.QIL_METHOD_SVD if method == "svd" else L.QIL_METHOD_RSVD, float(cutoff), _maxdim(maxdim),
               int(k), int(p), int(q), C.c_uint64(random_seed), int(mindim), C.byref(h)))
    return cls(ctx=ctx, _handle=h)


def _encode_batch(fn, cls, xs, method, cutoff, maxdim, k, p, q, random_seed, mindim, ctx):
    if method not in ("svd", "rsvd"):
        raise ValueError(f"tensor_to_mps: unknown method {method}. Use :svd or :rsvd.")
    ctx = ctx or default_context()
    xs = [np.asarray(x) for x in xs]
    if not xs:
        return []
    code = L.QIL_C64 if any(np.iscomplexobj(x) for x in xs) else L.QIL_F64
    keep = [np.ascontiguousarray(x, dtype=_np_dtype(code)) for x in xs]
    N = len(keep[0])
    if any(x.ndim != 1 or len(x) != N for x in keep):
        raise ValueError("signal batch: all signals must be 1-D and of one length")
    n = max(1, int(round(np.log2(max(N, 1)))))
    if N < 2 ** n:
        warnings.warn(f"_array_to_tensor: input length {N} is not a power of 2; zero-filling to {2**n}")
    ptrs = (C.c_void_p * len(keep))(*[x.ctypes.data for x in keep])
    outs = (C.c_void_p * len(keep))()
    L.check(fn(ctx.handle, ptrs, len(keep), N, code,
               L.QIL_METHOD_SVD if method == "svd" else L.QIL_METHOD_RSVD, float(cutoff), _maxdim(maxdim),
               int(k), int(p), int(q), C.c_uint64(random_seed), int(mindim), outs))
    return [cls(ctx=ctx, _handle=C.c_void_p(h)) for h in outs]


def signal_mps_batch(xs, method="svd", cutoff=1e-15, maxdim=None, k=20, p=10, q=0, random_seed=1234, mindim=1, ctx=None):
    """signal_mps for several signals of one length (the signal kinds of a benchmark sweep), encoded concurrently on the
    context's streams; item j is exactly signal_mps(xs[j]; ...)."""
    return _encode_batch(L.lib.qil_signal_mps_batch, SignalMPS, xs, method, cutoff, maxdim, k, p, q, random_seed, mindim, ctx)


def signal_ztmps_batch(xs, cutoff=1e-10, maxdim=None, method="svd", k=20, p=10, q=0, random_seed=1234, mindim=1, ctx=None):
    """signal_ztmps for several signals of one length, encoded concurrently; item j is exactly signal_ztmps(xs[j]; ...)."""
    return _encode_batch(L.lib.qil_signal_ztmps_batch, ZTMPS, xs, method, cutoff, maxdim, k, p, q, random_seed, mindim, ctx)


def signal_mps(x, method="svd", cutoff=1e-15, maxdim=None, k=20, p=10, q=0, random_seed=1234, mindim=1,
               ctx=None):
    """signal_mps(x; method=:svd, cutoff, maxdim, k, p, q, random_seed, mindim)."""
    return _encode(L.lib.qil_signal_mps, SignalMPS, x, method, cutoff, maxdim, k, p, q, random_seed, mindim, ctx)


def signal_ztmps(x, cutoff=1e-10, maxdim=None, method="svd", k=20, p=10, q=0, random_seed=1234, mindim=1,
                 ctx=None):
    """signal_ztmps(x; cutoff=1e-10, maxdim, kwargs...)."""
    return _encode(L.lib.qil_signal_ztmps, ZTMPS, x, method, cutoff, maxdim, k, p, q, random_seed, mindim, ctx)


def _factor_out(m, n, r0, code):
    dt = _np_dtype(code)
    return (np.empty((m, r0), dtype=dt, order="F"), np.empty(r0, dtype=np.float64),
            np.empty(r0 * n, dtype=dt))


def rsvd(A, k=20, p=10, q=0, random_seed=1234, cutoff=1e-15, maxdim=None, mindim=1, ctx=None):
    """rsvd(A, Linds...; k, p, q, random_seed, cutoff, maxdim=k, mindim) on the matricised
    operand A (m x n).  Returns (U, S, Vh) with A ~= U diag(S) Vh."""
    ctx = ctx or default_context()
    A = np.asarray(A)
    if A.ndim != 2 or A.shape[0] == 0 or A.shape[1] == 0:
        raise ValueError("In `rsvd`, left or right index set is empty.")
    code = L.QIL_C64 if np.iscomplexobj(A) else L.QIL_F64
    Af = np.asfortranarray(A, dtype=_np_dtype(code))
    m, n = Af.shape
    r0 = min(k + p, m, n)
    U, S, Vh = _factor_out(m, n, r0, code)
    r = C.c_int64()
    L.check(L.lib.qil_rsvd(ctx.handle, Af.ctypes.data_as(C.c_void_p), m, n, code, int(k), int(p), int(q),
                           C.c_uint64(random_seed), float(cutoff), int(k if maxdim is None else maxdim),
                           int(mindim), C.byref(r), U.ctypes.data_as(C.c_void_p),
                           S.ctypes.data_as(C.POINTER(C.c_double)), Vh.ctypes.data_as(C.c_void_p)))
    r = r.value
    return U[:, :r].copy(), S[:r].copy(), Vh[: r * n].reshape((r, n), order="F").copy()


def svd_trunc(A, cutoff=None, maxdim=None, mindim=1, ctx=None):
    """Truncated svd with the ITensors rule (keep largest; drop tail while the discarded squared
    weight <= cutoff * total; cap maxdim; floor mindim)."""
    ctx = ctx or default_context()
    A = np.asarray(A)
    code = L.QIL_C64 if np.iscomplexobj(A) else L.QIL_F64
    Af = np.asfortranarray(A, dtype=_np_dtype(code))
    m, n = Af.shape
    r0 = min(m, n)
    U, S, Vh = _factor_out(m, n, r0, code)
    r = C.c_int64()
    L.check(L.lib.qil_svd_trunc(ctx.handle, Af.ctypes.data_as(C.c_void_p), m, n, code,
                                -1.0 if cutoff is None else float(cutoff), _maxdim(maxdim), int(mindim),
                                C.byref(r), U.ctypes.data_as(C.c_void_p),
                                S.ctypes.data_as(C.POINTER(C.c_double)), Vh.ctypes.data_as(C.c_void_p)))
    r = r.value
    return U[:, :r].copy(), S[:r].copy(), Vh[: r * n].reshape((r, n), order="F").copy()


_OPS = {"N": 0, "T": 1, "H": 2, "C": 3}


def gemm(A, B, opA="N", opB="N", ctx=None):
    """op(A) @ op(B) on the GPU's f64 matrix cores (utility / test hook); op in N, T, H, C(onj)."""
    ctx = ctx or default_context()
    A, B = np.asarray(A), np.asarray(B)
    code = L.QIL_C64 if (np.iscomplexobj(A) or np.iscomplexobj(B)) else L.QIL_F64
    dt = _np_dtype(code)
    Af, Bf = np.asfortranarray(A, dtype=dt), np.asfortranarray(B, dtype=dt)
    m, k = (Af.shape if opA in "NC" else Af.shape[::-1])
    k2, n = (Bf.shape if opB in "NC" else Bf.shape[::-1])
    if k != k2:
        raise ValueError(f"gemm: inner dimensions disagree ({k} vs {k2})")
    Cm = np.empty((m, n), dtype=dt, order="F")
    L.check(L.lib.qil_gemm(ctx.handle, code, _OPS[opA], _OPS[opB], m, n, k, Af.ctypes.data_as(C.c_void_p),
                           Af.shape[0], Bf.ctypes.data_as(C.c_void_p), Bf.shape[0],
                           Cm.ctypes.data_as(C.c_void_p), m))
    return Cm


def gemm_device_time(m, n, k, dtype=np.float64, opA="N", opB="N", reps=10, ctx=None):
    """ms per call of the device-resident MFMA GEMM (diagnostic)."""
    ctx = ctx or default_context()
    code = L.QIL_C64 if np.dtype(dtype) == np.complex128 else L.QIL_F64
    ms = C.c_double()
    L.check(L.lib.qil_gemm_device_time(ctx.handle, code, _OPS[opA], _OPS[opB], int(m), int(n), int(k),
                                       int(reps), C.byref(ms)))
    return ms.value


def qr_positive(A, ctx=None):
    """Thin QR with non-negative diagonal of R (the device Gram-Schmidt QR; utility / test hook)."""
    ctx = ctx or default_context()
    A = np.asarray(A)
    code = L.QIL_C64 if np.iscomplexobj(A) else L.QIL_F64
    Af = np.asfortranarray(A, dtype=_np_dtype(code))
    m, n = Af.shape
    Q = np.empty((m, n), dtype=Af.dtype, order="F")
    R = np.empty((n, n), dtype=Af.dtype, order="F")
    L.check(L.lib.qil_qr_positive(ctx.handle, code, m, n, Af.ctypes.data_as(C.c_void_p),
                                  Q.ctypes.data_as(C.c_void_p), R.ctypes.data_as(C.c_void_p)))
    return Q, R
