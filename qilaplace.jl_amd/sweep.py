"""Embarrassingly-parallel sweeps over independent work items (signals, damping values sigma).

The reference loops serially over kinds / omega_r (scripts/benchmark/zt_full_runtime.jl:151-221;
docs/src/tutorials/zt.jl:300-348 builds one MPO per omega_r).  Each (signal, sigma) pipeline
encode -> build -> apply -> sample is self-contained, so items are dealt round-robin to the ranks
(one process per GPU) with NO data-path collective; the only communication is one all_gather of
the per-item coefficient batches at the end (KB-scale, latency-bound; RCCL over xGMI on GPUs,
gloo in the CPU tests).
"""
from __future__ import annotations

import ctypes as C
import os
import time

import numpy as np

from . import _lib as L

COMM_ID_BYTES = 128


class Comm:
    """RCCL communicator of the C ABI (qil_comm_*): the sweep's gather without torch.distributed -- what a Julia host calls.

    ``Comm(ctx, rank, world, uid)`` is collective (every rank, same 128-byte ``uid`` from ``Comm.unique_id()`` on rank 0).
    ``Comm.from_env(ctx)`` reads RANK / WORLD_SIZE and passes the id through a file: rank 0 writes
    ``$QIL_COMM_FILE`` (default ``<tmpdir>/qil_comm_<uid>_<MASTER_PORT>_<QIL_COMM_TAG or launcher pid>.id``, mode 0600) atomically --
    the id plus its own pid and start time -- and the others wait for a record whose writer is alive (a stale file of a crashed
    job is never accepted).  One node, one shared /tmp and /proc: the scope of SURVEY.md 8(e)."""

    def __init__(self, ctx, rank: int, world: int, uid: bytes):
        if len(uid) != COMM_ID_BYTES:
            raise ValueError(f"Comm: unique id must be {COMM_ID_BYTES} bytes, got {len(uid)}")
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        h = C.c_void_p()
        buf = C.create_string_buffer(uid, COMM_ID_BYTES)
        L.check(L.lib.qil_comm_create(ctx.handle, self.rank, self.world, C.cast(buf, C.c_void_p), C.byref(h)))
        self.handle = h

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(COMM_ID_BYTES)
        L.check(L.lib.qil_comm_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    @staticmethod
    def _proc_start_ticks(pid: int):
        """Start time of a live process in clock ticks since boot (/proc/<pid>/stat field 22), None if it is gone / no procfs."""
        try:
            with open(f"/proc/{int(pid)}/stat", "rb") as f:
                return int(f.read().rsplit(b")", 1)[1].split()[19])
        except (OSError, ValueError, IndexError):
            return None

    @classmethod
    def _rendezvous_path(cls):
        import tempfile
        return os.environ.get("QIL_COMM_FILE") or os.path.join(
            tempfile.gettempdir(), f"qil_comm_{os.getuid()}_{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('QIL_COMM_TAG', os.getppid())}.id")

    @classmethod
    def _read_rendezvous(cls, path):
        """The id rank 0 published at `path`, or None while there is nothing TRUSTWORTHY there.  The record is the 128-byte id
        followed by the writer's pid and process start time; a record is accepted only if it is ours (st_uid), complete, and its
        writer is still alive with that start time -- a file left behind by a crashed earlier job under the same key (same uid,
        MASTER_PORT and launcher pid: ADVICE r05) names a dead process and is ignored until rank 0 replaces it."""
        import struct
        try:
            st = os.stat(path)
            if st.st_uid != os.getuid() or st.st_size != COMM_ID_BYTES + 16:
                return None
            with open(path, "rb") as f:
                rec = f.read()
        except OSError:
            return None
        if len(rec) != COMM_ID_BYTES + 16:
            return None
        pid, ticks = struct.unpack("<qq", rec[COMM_ID_BYTES:])
        alive = cls._proc_start_ticks(pid)
        if os.path.isdir("/proc/self") and alive != ticks:
            return None
        return rec[:COMM_ID_BYTES]

    @classmethod
    def from_env(cls, ctx, timeout_s: float = 120.0):
        """Every rank calls this (RANK / WORLD_SIZE from the launcher).  Rank 0 creates the id and publishes it in a file keyed by
        (uid, MASTER_PORT, QIL_COMM_TAG or the launcher's pid); two jobs that run at the same time under one launcher shell with
        the same MASTER_PORT need distinct QIL_COMM_TAG values (bench.py's spawner sets one per job)."""
        import struct
        rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        if world == 1:
            return cls(ctx, 0, 1, cls.unique_id())
        path = cls._rendezvous_path()
        if rank == 0:
            uid = cls.unique_id()
            me = os.getpid()
            rec = uid + struct.pack("<qq", me, cls._proc_start_ticks(me) or 0)
            tmp = f"{path}.{me}.tmp"
            fd = os.open(tmp, os.O_CREAT | os.O_EXCL | os.O_WRONLY, 0o600)      # never through somebody else's file or link
            with os.fdopen(fd, "wb") as f:
                f.write(rec)
            os.replace(tmp, path)                                              # atomically over whatever was there
        else:
            t0 = time.monotonic()
            while True:
                uid = cls._read_rendezvous(path)
                if uid is not None:
                    break
                if time.monotonic() - t0 > timeout_s:
                    raise TimeoutError(f"Comm.from_env: rank 0 never published a live id at {path}")
                time.sleep(0.01)
        comm = cls(ctx, rank, world, uid)              # collective: every rank has read the file when this returns
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
        return comm

    def gather_coefficients(self, local: dict, n_items: int, width: int):
        """{item index -> complex vector (width,)} of THIS rank's items -> (n_items, width) in item order on every rank:
        one ncclAllGather (qil_gather_coefficients)."""
        mine = shard_items(n_items, self.world, self.rank)
        loc = np.zeros((max(len(mine), 1), width), dtype=np.complex128)
        for slot, i in enumerate(mine):
            loc[slot] = local[i]
        out = np.zeros((n_items, width), dtype=np.complex128)
        L.check(L.lib.qil_gather_coefficients(self.handle, int(n_items), int(width),
                                              loc.ctypes.data_as(C.POINTER(C.c_double)), out.ctypes.data_as(C.POINTER(C.c_double))))
        return out

    def gather_coefficients_device(self, local_dev_ptr: int, n_items: int, width: int, out_dev_ptr: int):
        """The same gather on DEVICE buffers of this communicator's context (raw pointers, e.g. `site_device_ptr` or any
        __cuda_array_interface__ producer): stream-ordered, nothing crosses PCIe (qil_gather_coefficients_device)."""
        L.check(L.lib.qil_gather_coefficients_device(self.handle, int(n_items), int(width), C.c_void_p(int(local_dev_ptr)),
                                                     C.c_void_p(int(out_dev_ptr))))

    def close(self):
        if getattr(self, "handle", None):
            L.lib.qil_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:                              # noqa: BLE001  (interpreter shutdown)
            pass


def unshuffle(world: int, n_items: int, width: int, gathered):
    """The gather's layout rule (qil_sweep_unshuffle, host only): world blocks of ceil(n_items / world) x width complex values
    in rank order -> (n_items, width) in item order."""
    g = np.ascontiguousarray(gathered, dtype=np.complex128)
    out = np.zeros((n_items, width), dtype=np.complex128)
    L.check(L.lib.qil_sweep_unshuffle(int(world), int(n_items), int(width), g.ctypes.data_as(C.POINTER(C.c_double)),
                                      out.ctypes.data_as(C.POINTER(C.c_double))))
    return out


def shard_items(n_items: int, world: int, rank: int):
    """Static round-robin partition: rank r owns items r, r + world, r + 2 world, ..."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside [0, {world})")
    return list(range(rank, n_items, world))


def _world_rank(dist):
    if isinstance(dist, Comm):
        return dist.world, dist.rank
    world = dist.get_world_size() if dist is not None and dist.is_initialized() else 1
    return world, (dist.get_rank() if world > 1 else 0)


def gather_results(local: dict, n_items: int, width: int, dist=None, device=None, always_gather=False):
    """`dist`: a torch.distributed module with an initialised process group, or a `Comm` (RCCL through the C ABI), or None.
    All ranks contribute {item index -> complex vector of length `width`}; returns the
    (n_items, width) complex array in item order on every rank.  `always_gather`: run the collective even in a world of
    one rank (exercises the RCCL path on a 1-GPU box: bench.py with QIL_BENCH_FORCE_DIST=1)."""
    if isinstance(dist, Comm):                     # the C ABI's RCCL gather (no torch): qil_gather_coefficients
        if dist.world == 1 and not always_gather:
            dist = None
        else:
            return dist.gather_coefficients(local, n_items, width)
    if dist is None or not dist.is_initialized() or (dist.get_world_size() == 1 and not always_gather):
        out = np.zeros((n_items, width), dtype=np.complex128)
        for i, v in local.items():
            out[i] = v
        return out
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    per = (n_items + world - 1) // world
    buf = np.zeros((per, width, 2), dtype=np.float64)
    for slot, i in enumerate(shard_items(n_items, world, rank)):
        v = np.asarray(local[i], dtype=np.complex128)
        buf[slot, :, 0], buf[slot, :, 1] = v.real, v.imag
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    parts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(parts, t)                      # the one collective of the sweep
    out = np.zeros((n_items, width), dtype=np.complex128)
    for r, p in enumerate(parts):
        p = p.cpu().numpy()
        for slot, i in enumerate(shard_items(n_items, world, r)):
            out[i] = p[slot, :, 0] + 1j * p[slot, :, 1]
    return out


def sweep(items, work_fn, width: int, dist=None, device=None, always_gather=False):
    """Run ``work_fn(item) -> complex vector (width,)`` on this rank's share of ``items`` and
    gather all results in item order."""
    world, rank = _world_rank(dist)
    local = {i: work_fn(items[i]) for i in shard_items(len(items), world, rank)}
    return gather_results(local, len(items), width, dist if (world > 1 or always_gather) else None, device, always_gather)


def damping_sample_bits(n: int, nsamp: int, seed: int = 7, kmax: int = 64):
    """(nsamp, 2n) paired-register configurations for checking a damping sweep where its values are NOT negligible.

    The DT output is x_j exp(-sigma k j / N) / sqrt(N) (test/test_dt_transformer.jl:60-92), so uniformly random (k, j)
    at n = 24 give k j / N ~ 1e6 and every reference value underflows to 0.0 -- a check against zeros.  The reference's
    own tests and tutorial look at small k (test/test_dt_transformer.jl:211-238, docs/src/tutorials/dt.jl:150-197).
    Strata: 1/8 the k = 0 row (undamped signal), 3/8 k in {1, 2, 3} with uniform j, 1/2 k uniform in [0, kmax) with
    log-uniform j (j < 2^m, m uniform in 10..n).  Returns (bits, k, j); main bits are LSB first on the even positions,
    copy bits MSB first on the odd ones (mps.jl:421-444)."""
    r = np.random.default_rng(seed)
    N = 1 << n
    q = nsamp // 8
    rest = nsamp - 4 * q
    kmax = min(kmax, N)
    kk = np.concatenate([np.zeros(q, np.int64), r.integers(1, min(4, N), 3 * q), r.integers(0, kmax, rest)]).astype(np.int64)
    m = r.integers(min(10, n), n + 1, rest)
    jj = np.concatenate([r.integers(0, N, 4 * q), (r.random(rest) * 2.0 ** m).astype(np.int64)]).astype(np.int64)
    bits = np.zeros((nsamp, 2 * n), dtype=np.uint8)
    bits[:, 0::2] = (kk[:, None] >> np.arange(n)[None, :]) & 1
    bits[:, 1::2] = (jj[:, None] >> np.arange(n - 1, -1, -1)[None, :]) & 1
    return bits, kk, jj


def damping_sweep(psi, sigmas, bits, build_mpo=None, dist=None, device=None, cutoff=1e-14, maxdim=1000, always_gather=False):
    """BASELINE.json configs[3]: one paired-register signal x many damping values.

    psi        device ZTMPS (replicated on every rank; it is MBs)
    sigmas     sequence of omega_r values, dealt round-robin to the ranks
    bits       (nb, 2n) sampled configurations
    build_mpo  None (default): this rank's share of the DT MPOs is built in ONE launch of the persistent device
               builder (`build_dt_mpo_batch`, one workgroup per damping value) and applied / sampled by
               `apply_coefficient_sweep` (one synchronisation for the whole share).  A callable
               sigma -> list of numpy site tensors W[a, s_in, s_out, b] takes the per-value host route instead
               (any other operator family).
    Returns (len(sigmas), nb) coefficients of W(sigma) * psi, in sigma order, on every rank."""
    from .containers import PairedSiteMPO
    from .ops import apply, coefficient_batch, apply_coefficient_sweep
    bits = np.asarray(bits)
    sigmas = list(sigmas)
    if build_mpo is not None:
        def work(sig):
            W = PairedSiteMPO(build_mpo(sig), sites=psi.site_ids, ctx=psi.ctx)
            return coefficient_batch(apply(W, psi), bits)

        return sweep(sigmas, work, bits.shape[0], dist, device, always_gather)
    from .builders import build_dt_mpo_batch
    world, rank = _world_rank(dist)
    mine = shard_items(len(sigmas), world, rank)
    local = {}
    if isinstance(dist, Comm) and (world > 1 or always_gather):
        # the C ABI's route: this rank's samples never leave HBM before the one all-gather (qil_apply_coefficient_sweep_gather)
        Ws = build_dt_mpo_batch(psi, [sigmas[i] for i in mine], cutoff, maxdim, psi.ctx) if mine else []
        return apply_coefficient_sweep(Ws, psi, bits, comm=dist, n_items=len(sigmas))
    if mine:
        Ws = build_dt_mpo_batch(psi, [sigmas[i] for i in mine], cutoff, maxdim, psi.ctx)
        res = apply_coefficient_sweep(Ws, psi, bits)
        local = {i: res[k] for k, i in enumerate(mine)}
    return gather_results(local, len(sigmas), bits.shape[0], dist if (world > 1 or always_gather) else None, device, always_gather)
