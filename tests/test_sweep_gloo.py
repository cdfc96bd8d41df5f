"""The N>1 path on CPU: world_size-2 gloo run of the sweep driver (sharding + the one gather).
The per-item work is the oracle here (no GPU in this container); on the GPU box the same driver
runs the HIP path (tests/test_gpu_parity.py::test_damping_sweep_single_rank)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _work_item(sig):
    import oracle as O
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import interleave
    from oracle.analytic import int_to_bits
    n = 3
    x = np.exp(-0.8 * 0.3 * np.arange(2 ** n))
    out = O.apply(O.build_dt_mpo(n, sig), O.signal_ztmps(x, cutoff=1e-14))
    bits = np.array([interleave(int_to_bits(k, n, "lsb"), int_to_bits(j, n)) for k in range(2 ** n) for j in (0, 3)])
    return O.coefficient_batch(out, bits).astype(np.complex128)


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sweep = importlib.import_module("qilaplace_jl_amd.sweep")
    sig = [0.25 * (i + 1) for i in range(5)]                 # ragged: 5 items over 2 ranks
    res = sweep.sweep(sig, _work_item, 16, dist)
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_items():
    sweep = importlib.import_module("qilaplace_jl_amd.sweep")
    assert sweep.shard_items(64, 8, 3) == list(range(3, 64, 8))
    assert sweep.shard_items(5, 2, 0) == [0, 2, 4] and sweep.shard_items(5, 2, 1) == [1, 3]
    assert sweep.shard_items(1, 4, 2) == []
    got = sorted(i for r in range(8) for i in sweep.shard_items(64, 8, r))
    assert got == list(range(64))
    with pytest.raises(ValueError):
        sweep.shard_items(4, 2, 2)


def test_sweep_world2_gloo_matches_serial():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    serial = np.stack([_work_item(0.25 * (i + 1)) for i in range(5)])
    for r in range(2):
        assert np.abs(results[r] - serial).max() < 1e-14


def test_cabi_gather_layout_equals_the_gloo_gather():
    """qil_sweep_unshuffle -- the layout rule of the C ABI's RCCL gather (qil_gather_coefficients: per-rank blocks of
    ceil(n_items / world) x width, rank order -> item order) -- gives what the torch.distributed path of sweep.py gives:
    world 2 against the gloo run's construction (ragged share), and worlds 1 / 3 / 8 against the definition."""
    sweep = importlib.import_module("qilaplace_jl_amd.sweep")
    rng = np.random.default_rng(8)
    for world, n_items, width in ((2, 5, 16), (1, 4, 3), (3, 10, 7), (8, 64, 1024), (8, 3, 2), (4, 0, 5)):
        items = rng.standard_normal((n_items, width)) + 1j * rng.standard_normal((n_items, width))
        per = (n_items + world - 1) // world
        blocks = np.zeros((world, per, width), dtype=np.complex128)          # what every rank would contribute
        for r in range(world):
            for slot, i in enumerate(sweep.shard_items(n_items, world, r)):
                blocks[r, slot] = items[i]
        out = sweep.unshuffle(world, n_items, width, blocks)
        assert np.array_equal(out, items)
        # ... and the torch path's own reassembly of the same blocks (gather_results' loop), restated
        ref = np.zeros_like(items)
        for r in range(world):
            for slot, i in enumerate(sweep.shard_items(n_items, world, r)):
                ref[i] = blocks[r, slot]
        assert np.array_equal(out, ref)
    with pytest.raises(ValueError):
        sweep.unshuffle(0, 4, 4, np.zeros((4, 4), dtype=np.complex128))


def test_comm_requires_a_full_id():
    sweep = importlib.import_module("qilaplace_jl_amd.sweep")
    with pytest.raises(ValueError, match="128 bytes"):
        sweep.Comm(None, 0, 1, b"short")
