"""Producers of the transform MPOs (the operand ``W`` of the hot path): device builders by default, host chains beside them.

Reference: src/transforms/qft_transformer.jl:121-165, dt_transformer.jl:312-412, zt_transformer.jl:41-112, with the gate
blocks of src/circuits/{qft,dt,zt}_gates.jl -- CPU chains of tiny latency-bound QR / SVD steps (D <= 8 / 18 / 92).
``build_qft_mpo`` / ``build_dt_mpo(_batch)`` / ``build_zt_mpo(_batch)`` build in HBM for EVERY n (the persistent
kernels of csrc/qil_build_persist.hip and qil_build_chain.hip, one launch per chain / per damping sweep; r06: the whole
build_zt_mpo behind one C verb, qil_build_zt_mpo_batch; only 2 x 2 gate entries come from the host) and return device handles
(SingleSiteMPO / PairedSiteMPO).  The numpy chains of this file (``*_tensors``, ``device=False``) remain as the independent
restatement the device builders are tested against; they return the site tensors W[a, s_in, s_out, b].

Implementation notes: every sweep is written once, for the left-to-right direction; the right-to-left
("up") variants of the reference run the same routine on the mirrored chain.  Truncation is the
ITensors rule (keep largest; drop the tail while the discarded squared weight <= cutoff * total; cap
maxdim).
"""
from __future__ import annotations

import numpy as np

from .containers import SingleSiteMPO, PairedSiteMPO

_I = np.eye(2)
_H = np.array([[1.0, 1.0], [1.0, -1.0]]) / np.sqrt(2.0)


def _P(theta):                                       # qft_gates.jl:24-30
    return np.diag([1.0, np.exp(-1j * theta)])


def _R(w):                                           # dt_gates.jl:19-25
    return np.diag([1.0, np.exp(-w)])


def _Hd(w):                                          # dt_gates.jl:11-17
    return np.array([[1.0, 1.0], [1.0, np.exp(-w / 2.0)]]) / np.sqrt(2.0)


def _proj(i):
    M = np.zeros((2, 2))
    M[i, i] = 1.0
    return M


def _site(entries, Dl, Dr, dtype):
    """entries: {(a, b): 2x2 gate[s_in, s_out]}."""
    W = np.zeros((Dl, 2, 2, Dr), dtype=dtype)
    for (a, b), g in entries.items():
        W[a, :, :, b] += g
    return W


# ------------------------------------------------------------------ gate blocks
def _qft_block(k):
    """control_Hphase_mpo (qft_gates.jl:43-97): H then project the output of site 1; P(2pi/2^l) on site l."""
    c = np.complex128
    if k == 1:
        return [_site({(0, 0): _H}, 1, 1, c)]
    out = [_site({(0, 0): _H @ _proj(0), (0, 1): _H @ _proj(1)}, 1, 2, c)]
    for l in range(2, k):
        out.append(_site({(0, 0): _I, (1, 1): _P(2 * np.pi / 2.0 ** l)}, 2, 2, c))
    out.append(_site({(0, 0): _I, (1, 0): _P(2 * np.pi / 2.0 ** k)}, 2, 1, c))
    return out


def _dt_main_block(k, w):
    """control_damping_mpo (dt_gates.jl:30-130): control = input bit of main_k."""
    f = np.float64
    if k == 1:
        return [_site({(0, 0): _Hd(w)}, 1, 1, f), _site({(0, 0): _I}, 1, 1, f)]
    out = []
    for l in range(1, k):
        Rf = _R(w * 2.0 ** (l - k - 1))
        out.append(_site({(0, 0): _I, (0 if l == 1 else 1, 1): Rf}, 1 if l == 1 else 2, 2, f))
        out.append(_site({(0, 0): _I, (1, 1): _I}, 2, 2, f))
    out.append(_site({(0, 0): _proj(0) @ _Hd(w), (1, 1): _proj(1) @ _Hd(w)}, 2, 2, f))
    out.append(_site({(0, 0): _I, (1, 0): _I}, 2, 1, f))
    return out


def _dt_copy_block(n, k, w):
    """control_damping_copy_mpo (dt_gates.jl:133-229): control = projector on copy_k; pairs k..n."""
    f = np.float64
    L = n - k + 1
    if L == 1:
        return [_site({(0, 0): _I}, 1, 1, f), _site({(0, 0): _I}, 1, 1, f)]
    out = [_site({(0, 0): _I}, 1, 2, f), _site({(0, 0): _proj(0), (0, 1): _proj(1)}, 2, 2, f)]
    for j in range(2, L + 1):
        out.append(_site({(0, 0): _I, (1, 1): _R(w * 2.0 ** (j - 2))}, 2, 2, f))
        last = j == L
        out.append(_site({(0, 0): _I, (1, 0 if last else 1): _I}, 2, 1 if last else 2, f))
    return out


def _zt_block(k):
    """control_Hphase_ztmps_mpo (zt_gates.jl:12-114): control = input bit of copy_k (project, then H)."""
    c = np.complex128
    if k == 1:
        return [_site({(0, 0): _I}, 1, 1, c), _site({(0, 0): _H}, 1, 1, c)]
    out = [_site({(0, 0): _I, (0, 1): _I}, 1, 2, c),
           _site({(0, 0): _I, (1, 1): _P(2 * np.pi / 2.0 ** k)}, 2, 2, c)]
    for j in range(2, k):
        out.append(_site({(0, 0): _I, (1, 1): _I}, 2, 2, c))
        out.append(_site({(0, 0): _I, (1, 1): _P(2 * np.pi / 2.0 ** (k - j + 1))}, 2, 2, c))
    out.append(_site({(0, 0): _I, (1, 1): _I}, 2, 2, c))
    out.append(_site({(0, 0): _proj(0) @ _H, (1, 0): _proj(1) @ _H}, 2, 1, c))
    return out


# ------------------------------------------------------------------ chain algebra
def _mirror(chain):
    """Reverse the site order and swap the two bond axes of every tensor."""
    return [np.ascontiguousarray(t.transpose(3, 1, 2, 0)) for t in reversed(chain)]


def _keep(S, cutoff, maxdim):
    P = S * S
    n = len(P)
    if n <= 1 or P[0] <= 0:
        return 1
    err = 0.0
    if maxdim is not None:
        while n > maxdim:
            err += P[n - 1]
            n -= 1
    if cutoff is not None:
        tot = P.sum() or 1.0
        while n > 1 and err + P[n - 1] <= cutoff * tot:
            err += P[n - 1]
            n -= 1
    return max(n, 1)


def _svd(M, cutoff, maxdim):
    try:
        U, S, Vh = np.linalg.svd(M, full_matrices=False)
    except np.linalg.LinAlgError:                     # pragma: no cover
        import scipy.linalg
        U, S, Vh = scipy.linalg.svd(M, full_matrices=False, lapack_driver="gesvd")
    r = _keep(S, cutoff, maxdim)
    return U[:, :r], S[:r], Vh[:r]


def _compose(first, second):
    """Site tensor of (second o first): first's output leg feeds second's input leg; the bond of
    `first` is the fast index of the fused bonds (apply.jl:163-171)."""
    t = np.einsum("aimb,cmod->caiodb", first, second)
    return t.reshape(first.shape[0] * second.shape[0], 2, 2, first.shape[3] * second.shape[3])


def _zip_lr(M, B):
    """Left-aligned zip of block B (acting after M) into chain M with a QR remainder carried to the
    right (zip_to_combine_mpos "down", dt_transformer.jl:38-95).  len(B) <= len(M)."""
    out = list(M)
    T = np.ones((1, 1, 1), dtype=np.result_type(M[0], B[0]))            # (new, bondM, bondB)
    for k, Bk in enumerate(B):
        # core[r,i,o,b,d] = sum_{a,c,m} T[r,a,c] M[a,i,m,b] B[c,m,o,d] as two GEMM-shaped contractions (the three-operand
        # einsum ran numpy's unoptimised loops: a third of the whole QFT-half build)
        X = np.tensordot(T, M[k], axes=([1], [0]))                        # (r, c, i, m, b)
        core = np.tensordot(X, Bk, axes=([1, 3], [0, 1])).transpose(0, 1, 3, 2, 4)   # (r, i, b, o, d) -> (r, i, o, b, d)
        r, b1, b2 = core.shape[0], core.shape[3], core.shape[4]
        Q, Rm = np.linalg.qr(core.reshape(r * 4, b1 * b2))
        out[k] = Q.reshape(r, 2, 2, Q.shape[1])
        T = Rm.reshape(Q.shape[1], b1, b2)
    T = T[:, :, 0]                                                        # B has ended: its bond is 1
    if len(M) > len(B):
        out[len(B)] = np.tensordot(T, out[len(B)], axes=([1], [0]))
    else:
        out[-1] = np.tensordot(out[-1], T, axes=([3], [0]))
    return out


def _compress_lr(M, cutoff, maxdim):
    """QR gauge sweep left -> right, then truncating two-site SVD sweep right -> left
    (zip_to_compress_mpo "down", dt_transformer.jl:185-230)."""
    out = list(M)
    L = len(out)
    for i in range(L - 1):
        a, _, _, b = out[i].shape
        Q, Rm = np.linalg.qr(out[i].reshape(a * 4, b))
        out[i] = Q.reshape(a, 2, 2, Q.shape[1])
        out[i + 1] = np.tensordot(Rm, out[i + 1], axes=([1], [0]))
    for i in range(L - 1, 0, -1):
        a0, b1 = out[i - 1].shape[0], out[i].shape[3]
        core = np.tensordot(out[i - 1], out[i], axes=([3], [0])).reshape(a0 * 4, 4 * b1)
        U, S, Vh = _svd(core, cutoff, maxdim)
        out[i] = Vh.reshape(len(S), 2, 2, b1)
        out[i - 1] = (U * S).reshape(a0, 2, 2, len(S))
    return out


def _zip_rl(M, B):
    return _mirror(_zip_lr(_mirror(M), _mirror(B)))                      # "up", dt_transformer.jl:97-153


def _compress_rl(M, cutoff, maxdim):
    return _mirror(_compress_lr(_mirror(M), cutoff, maxdim))             # "up", dt_transformer.jl:233-276


def _pad_pair(M, dtype):
    eye = np.eye(2, dtype=dtype).reshape(1, 2, 2, 1)
    return list(M) + [eye.copy(), eye.copy()]


# ------------------------------------------------------------------ builders (host tensors)
def _single_thread_blas(fn):
    """The factorizations are tiny (D <= 8 / 18 / 92): a many-threaded BLAS only adds fork-join
    overhead (measured 3x on a 128-core host), so builds pin it to one thread."""
    import functools

    @functools.wraps(fn)
    def wrapped(*a, **kw):
        try:
            from threadpoolctl import threadpool_limits
        except ImportError:                           # pragma: no cover
            return fn(*a, **kw)
        with threadpool_limits(limits=1):
            return fn(*a, **kw)
    return wrapped


@_single_thread_blas
def qft_mpo_tensors(n, cutoff=1e-14, maxdim=1000):
    """build_qft_mpo (qft_transformer.jl:121-160): n-1 rounds of zip-up (QR, no truncation) on the
    trailing sites followed by a truncating zip-down SVD sweep."""
    if n < 1:
        raise ValueError(f"build_qft_mpo: Number of qubits 'n' must be at least 1. Found n={n}")
    M = _qft_block(n)
    for it in range(1, n):
        B = _qft_block(n - it)
        # zip-up == right-aligned zip with QR-type factorisation (no cutoff => QR), remainder to the left
        M = _zip_rl(M, B)
        for k in range(it - 1, n - 1):                                   # zip-down (:69-101)
            a, _, _, b = M[k].shape
            U, S, Vh = _svd(M[k].reshape(a * 4, b), cutoff, maxdim)
            M[k] = U.reshape(a, 2, 2, len(S))
            M[k + 1] = np.tensordot(S[:, None] * Vh, M[k + 1], axes=([1], [0]))
    return M


@_single_thread_blas
def dt_mpo_tensors(n, wr, cutoff=1e-14, maxdim=1000):
    """build_dt_mpo (dt_transformer.jl:312-407): part 1 main-control blocks k = 1..n zipped "down",
    part 2 copy-control blocks k = 1..n-1 zipped "up", whole-chain compression after each."""
    if n < 1:
        raise ValueError(f"build_dt_mpo: n must be >= 1. Found n={n}")
    M = _dt_main_block(1, wr)
    for k in range(2, n + 1):
        M = _compress_lr(_zip_lr(_pad_pair(M, np.float64), _dt_main_block(k, wr)), cutoff, maxdim)
    for k in range(1, n):
        M = _compress_rl(_zip_rl(M, _dt_copy_block(n, k, wr)), cutoff, maxdim)
    return M


@_single_thread_blas
def zt_mpo_tensors(n, wr, cutoff=1e-14, maxdim=1000):
    """build_zt_mpo (zt_transformer.jl:41-106): DT first, then the paired QFT chain, fused by one
    MPO x MPO product and one compression."""
    if n < 1:
        raise ValueError(f"build_zt_mpo: n must be >= 1. Found n={n}")
    Wdt = dt_mpo_tensors(n, wr, cutoff, maxdim)
    Q = _zt_block(1)
    for k in range(2, n + 1):
        Q = _compress_lr(_zip_lr(_pad_pair(Q, np.complex128), _zt_block(k)), cutoff, maxdim)
    W = [_compose(a, b) for a, b in zip(Wdt, Q)]
    return W if n == 1 else _compress_lr(W, cutoff, maxdim)


# ------------------------------------------------------------------ device handles (reference signatures)
def _n_of(x):
    return len(x) if hasattr(x, "handle") else int(x)


def build_qft_mpo(n_or_psi, sites=None, cutoff=1e-14, maxdim=1000, ctx=None, device=None):
    """build_qft_mpo(n, sites; cutoff, maxdim) / build_qft_mpo(psi::SignalMPS; ...).  Default (`device` None / True): the whole
    chain in one launch of the persistent complex builder (`qft_mpo_device` -> qil_build_qft_mpo; r04: n = 24 in 11 ms against
    13 ms for the host chain + upload, n = 8 0.8 against 1.6 ms); `device=False`: the host chain (`qft_mpo_tensors`, numpy)."""
    psi = n_or_psi if hasattr(n_or_psi, "handle") else None
    n = _n_of(n_or_psi)
    if sites is not None and len(sites) != n:
        raise ValueError(f"build_qft_mpo: Number of sites must be equal to n. Found length(sites)={len(sites)}, n={n}")
    if psi is not None and sites is None:
        sites, ctx = psi.site_ids, ctx or psi.ctx
    if device is None or device:
        return qft_mpo_device(n, sites, cutoff, maxdim, ctx)
    return SingleSiteMPO(qft_mpo_tensors(n, cutoff, maxdim), sites=sites, ctx=ctx)


def build_dt_mpo(n_or_psi, wr, cutoff=1e-14, maxdim=1000, ctx=None, device=None):
    """build_dt_mpo(n, wr, ...; cutoff, maxdim) / build_dt_mpo(psi::ZTMPS, wr; ...).  Default (`device` None / True), for every n:
    on the GPU (qil_build_dt_mpo_batch with one value); `device=False`: the host chain (`dt_mpo_tensors`, numpy) -- the
    independent restatement the device builder is tested against."""
    psi = n_or_psi if hasattr(n_or_psi, "handle") else None
    n = _n_of(n_or_psi)
    if device is None or device:
        return build_dt_mpo_batch(n_or_psi, [wr], cutoff, maxdim, ctx)[0]
    sites = psi.site_ids if psi is not None else None
    return PairedSiteMPO(dt_mpo_tensors(n, wr, cutoff, maxdim), sites=sites,
                         ctx=ctx or (psi.ctx if psi is not None else None))


def build_zt_mpo(n_or_psi, wr, cutoff=1e-14, maxdim=1000, ctx=None, device=None):
    """build_zt_mpo(n, wr, ...; cutoff, maxdim) / build_zt_mpo(psi::ZTMPS, wr; ...) (zt_transformer.jl:41-112).  Default
    (`device` None / True), for every n: ONE C verb, every step on the GPU (qil_build_zt_mpo_batch with one value: DT half and
    paired QFT chain concurrently on two streams, MPO x MPO product, compression); `device=False`: everything with the host
    chain (`zt_mpo_tensors`, numpy) -- the independent restatement."""
    psi = n_or_psi if hasattr(n_or_psi, "handle") else None
    n = _n_of(n_or_psi)
    if device is None or device:
        return build_zt_mpo_batch(n_or_psi, [wr], cutoff, maxdim, ctx)[0]
    sites = psi.site_ids if psi is not None else None
    return PairedSiteMPO(zt_mpo_tensors(n, wr, cutoff, maxdim), sites=sites,
                         ctx=ctx or (psi.ctx if psi is not None else None))


def _dt_worker(args):
    n, w, cutoff, maxdim = args
    return dt_mpo_tensors(n, w, cutoff, maxdim)


def dt_mpo_tensors_many(n, wrs, cutoff=1e-14, maxdim=1000, workers=8):
    """Independent builds for a sweep of damping values, spread over host PROCESSES (spawned, so no
    GPU state is inherited), one BLAS thread each -- the reference builds them one after another
    (1.6 s each at n=24)."""
    wrs = list(wrs)
    jobs = [(n, float(w), cutoff, maxdim) for w in wrs]
    if workers <= 1 or len(jobs) <= 1:
        return [_dt_worker(j) for j in jobs]
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    with ProcessPoolExecutor(max_workers=min(workers, len(jobs)), mp_context=mp.get_context("spawn")) as ex:
        return list(ex.map(_dt_worker, jobs))


def build_dt_mpo_batch(n_or_psi, wrs, cutoff=1e-14, maxdim=1000, ctx=None):
    """All damping values of a sweep built TOGETHER on the GPU (qil_build_dt_mpo_batch): ONE kernel launch, one
    workgroup per damping value running that value's whole chain in LDS.  Returns a list of PairedSiteMPO handles,
    each with the bond dimensions of a single build_dt_mpo call, labelled with psi's site ids when psi is given."""
    import ctypes as C
    from . import _lib as L
    from .containers import default_context
    psi = n_or_psi if hasattr(n_or_psi, "handle") else None
    ctx = ctx or (psi.ctx if psi is not None else default_context())
    n = _n_of(n_or_psi)
    w = np.ascontiguousarray(np.asarray(list(wrs), dtype=np.float64))
    outs = (C.c_void_p * len(w))()
    ids = None
    if psi is not None:                      # build_dt_mpo(psi::ZTMPS, ...) builds on psi's own sites (:409-412)
        ids = (C.c_int64 * (2 * n))(*[int(i) for i in psi.site_ids])
    L.check(L.lib.qil_build_dt_mpo_batch(ctx.handle, int(n), len(w), w.ctypes.data_as(C.POINTER(C.c_double)),
                                         float(cutoff), -1 if maxdim is None else int(maxdim), ids, outs))
    return [PairedSiteMPO(ctx=ctx, _handle=C.c_void_p(h)) for h in outs]


_ZT_Q_CACHE = {}


@_single_thread_blas
def zt_qft_chain_tensors(n, cutoff=1e-14, maxdim=1000):
    """The damping-independent half of build_zt_mpo (zt_transformer.jl:78-98): the paired-register QFT chain
    of control_Hphase_ztmps_mpo blocks.  Depends on n only, so sweeps over the damping build it once."""
    key = (int(n), float(cutoff), maxdim)
    if key not in _ZT_Q_CACHE:
        # the chain for n is the chain for n - 1 zipped with one more block: start from the longest cached prefix and keep
        # every intermediate (a later call with a smaller or slightly larger n costs nothing or a few steps)
        k0 = max((m for (m, c, d) in _ZT_Q_CACHE if c == key[1] and d == maxdim and m < n), default=1)
        Q = _ZT_Q_CACHE[(k0, key[1], maxdim)] if k0 > 1 else _zt_block(1)
        for k in range(k0 + 1, n + 1):
            Q = _compress_lr(_zip_lr(_pad_pair(Q, np.complex128), _zt_block(k)), cutoff, maxdim)
            _ZT_Q_CACHE[(k, key[1], maxdim)] = Q
        _ZT_Q_CACHE[key] = Q
    return _ZT_Q_CACHE[key]


# ------------------------------------------------------------------ QFT chains assembled ON THE DEVICE
def _persistent_chain(fn, cls, n, ids, cutoff, maxdim, ctx):
    """One launch of the persistent complex chain builder (csrc/qil_build_chain.hip); None when a bond exceeded its in-LDS
    capacity (the caller takes the generic device route)."""
    import ctypes as C
    from . import _lib as L
    out, fb = C.c_void_p(), C.c_int(0)
    arr = (C.c_int64 * len(ids))(*ids)
    L.check(fn(ctx.handle, int(n), float(cutoff), -1 if maxdim is None else int(maxdim), arr, C.byref(out), C.byref(fb)))
    if fb.value:
        return None
    return cls(ctx=ctx, _handle=C.c_void_p(out.value))


def qft_mpo_device(n, sites=None, cutoff=1e-14, maxdim=1000, ctx=None, persistent=True):
    """build_qft_mpo (qft_transformer.jl:121-165) with every factorisation on the GPU: round `it` multiplies the chain by the
    block control_Hphase_mpo(n - it) on its trailing sites -- the window product of apply(W1, W2) (apply.jl:124-199 ->
    qil_apply_mpo_mpo), exact, the bonds multiply -- and re-truncates it with zip_to_compress_mpo "up" (exact QR gauge
    sweep right -> left, truncating SVD sweep left -> right, dt_transformer.jl:233-276 -> qil_mpo_compress): the
    reference's zip-up (:13-66, QR, nothing dropped) followed by its zip-down (:69-101, SVD at `cutoff`) is the same
    exact-product-then-truncate step carried out site by site.  Only the 2 x 2 gate blocks are made on the host."""
    from .containers import default_context
    from .ops import apply, mpo_compress
    if n < 1:
        raise ValueError(f"build_qft_mpo: Number of qubits 'n' must be at least 1. Found n={n}")
    ctx = ctx or default_context()
    ids = [int(i) for i in sites] if sites is not None else list(range(1, n + 1))
    if persistent:
        from . import _lib as L
        W = _persistent_chain(L.lib.qil_build_qft_mpo, SingleSiteMPO, n, ids, cutoff, maxdim, ctx)
        if W is not None:
            return W
    M = SingleSiteMPO(_qft_block(n), sites=ids, ctx=ctx)
    for it in range(1, n):
        B = SingleSiteMPO(_qft_block(n - it), sites=ids[it:], ctx=ctx)
        M = apply(M, B)                                   # M first, then the block on the trailing n - it sites
        mpo_compress(M, "up", cutoff, maxdim)
    return M


def zt_qft_chain_device(n, sites=None, cutoff=1e-14, maxdim=1000, ctx=None, persistent=True):
    """The paired-register QFT half of build_zt_mpo (zt_transformer.jl:78-99) on the GPU: for k = 2..n the chain (2k - 2
    sites) is multiplied by control_Hphase_ztmps_mpo(k) (2k sites; the two new sites see the identity: the window product
    pads the shorter operand exactly as the reference's identity extension does) and compressed "down"
    (zip_to_combine_mpos + zip_to_compress_mpo, dt_transformer.jl:38-95, 185-230)."""
    from .containers import default_context
    from .ops import apply, mpo_compress
    if n < 1:
        raise ValueError(f"build_zt_mpo: n must be >= 1. Found n={n}")
    ctx = ctx or default_context()
    ids = [int(i) for i in sites] if sites is not None else list(range(1, 2 * n + 1))
    if persistent:
        from . import _lib as L
        W = _persistent_chain(L.lib.qil_build_zt_qft_chain, PairedSiteMPO, n, ids, cutoff, maxdim, ctx)
        if W is not None:
            return W
    Q = PairedSiteMPO(_zt_block(1), sites=ids[:2], ctx=ctx)
    for k in range(2, n + 1):
        B = PairedSiteMPO(_zt_block(k), sites=ids[:2 * k], ctx=ctx)
        Q = apply(Q, B)
        mpo_compress(Q, "down", cutoff, maxdim)
    return Q


def build_zt_mpo_batch(n_or_psi, wrs, cutoff=1e-14, maxdim=1000, ctx=None, workers=None, qft="device"):
    """z-transform MPOs for a sweep of damping values (zt_transformer.jl:41-112 per value).  Default `qft="device"`: the C verb
    qil_build_zt_mpo_batch -- the DT halves (one launch, one workgroup per value) and the paired QFT chain (one launch, built
    once: it does not depend on the damping) run concurrently on two streams of the context, then per value the MPO x MPO
    product (:103) and its compression (:104; the per-value chains run as one batch).  No host linear algebra anywhere.
    `qft="parts"`: the same steps composed from their own C entries (qil_build_dt_mpo_batch, qil_build_zt_qft_chain,
    qil_apply_mpo_mpo, qil_mpo_compress_batch) one after another -- the cross-check of the verb; `qft="host"`: the QFT chain from
    the numpy restatement (`zt_qft_chain_tensors`, cached per n) on a host thread next to the device DT build -- the r02-r05
    default, kept as a comparison route.  `workers` is ignored (callers of the earliest host-thread route)."""
    import ctypes as C
    from . import _lib as L
    from .containers import default_context
    psi = n_or_psi if hasattr(n_or_psi, "handle") else None
    n = _n_of(n_or_psi)
    if n < 1:
        raise ValueError(f"build_zt_mpo: n must be >= 1. Found n={n}")
    if qft not in ("host", "device", "parts"):
        raise ValueError(f"build_zt_mpo_batch: qft must be 'device', 'parts' or 'host', got {qft!r}")
    if qft != "device":
        return _build_zt_mpo_batch_parts(n_or_psi, wrs, cutoff, maxdim, ctx, qft)
    ctx = ctx or (psi.ctx if psi is not None else default_context())
    w = np.ascontiguousarray(np.asarray(list(wrs), dtype=np.float64))
    outs = (C.c_void_p * len(w))()
    ids = None
    if psi is not None:                      # build_zt_mpo(psi::ZTMPS, ...) builds on psi's own sites (:107-111)
        ids = (C.c_int64 * (2 * n))(*[int(i) for i in psi.site_ids])
    L.check(L.lib.qil_build_zt_mpo_batch(ctx.handle, int(n), len(w), w.ctypes.data_as(C.POINTER(C.c_double)),
                                         float(cutoff), -1 if maxdim is None else int(maxdim), ids, outs))
    return [PairedSiteMPO(ctx=ctx, _handle=C.c_void_p(h)) for h in outs]


def _build_zt_mpo_batch_parts(n_or_psi, wrs, cutoff, maxdim, ctx, qft):
    """The steps of qil_build_zt_mpo_batch from their own entry points (see build_zt_mpo_batch)."""
    from .ops import apply, mpo_compress_batch
    import threading
    n = _n_of(n_or_psi)
    # the host-side QFT half (numpy, GIL released inside LAPACK) is built while the GPU builds the DT halves
    box = {}

    def _host_half():
        try:
            box["Q"] = zt_qft_chain_tensors(n, cutoff, maxdim)
        except Exception as e:                      # noqa: BLE001  (re-raised on the calling thread)
            box["err"] = e

    th = threading.Thread(target=_host_half)
    if qft == "host":
        th.start()
    try:
        dts = build_dt_mpo_batch(n_or_psi, wrs, cutoff, maxdim, ctx)
    finally:
        if qft == "host":
            th.join()
    if "err" in box:
        raise box["err"]
    home = dts[0].ctx
    ids = dts[0].site_ids
    Q = (zt_qft_chain_device(n, ids, cutoff, maxdim, home) if qft == "parts"
         else PairedSiteMPO(box["Q"], sites=ids, ctx=home))
    prods = [apply(W_dt, Q) for W_dt in dts]
    if n == 1:
        return prods
    return mpo_compress_batch(prods, "down", cutoff, maxdim)
