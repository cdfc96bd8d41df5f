"""Comm.from_env's file rendezvous with 2-8 real processes and NO GPU: every rank must end
up with rank 0's id -- also when a stale record of a crashed earlier job sits at the same path (same uid / MASTER_PORT / launcher
pid), in the r05 format (bare 128 bytes) or in the r06 format naming a dead writer.  `Comm.__init__` (ncclCommInitRank) and
`Comm.unique_id` (ncclGetUniqueId) are stubbed: the test is about the host protocol only."""
import os
import struct
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, time
sys.path.insert(0, %(root)r)
import importlib
sweep = importlib.import_module("qilaplace_jl_amd.sweep")
seen = {}
def fake_init(self, ctx, rank, world, uid):
    self.ctx, self.rank, self.world, self.handle = ctx, rank, world, None
    seen["uid"] = uid
    # the real constructor is collective: nobody returns before everybody has arrived (rank 0 unlinks the file afterwards)
    d = os.environ["QIL_TEST_BARRIER_DIR"]
    open(os.path.join(d, "arrived.%%d" %% rank), "w").close()
    t0 = time.time()
    while len([f for f in os.listdir(d) if f.startswith("arrived.")]) < world:
        assert time.time() - t0 < 60, "barrier timed out"
        time.sleep(0.005)
sweep.Comm.__init__ = fake_init
sweep.Comm.unique_id = staticmethod(lambda: os.urandom(sweep.COMM_ID_BYTES))
time.sleep(float(os.environ.get("QIL_TEST_DELAY", "0")))
c = sweep.Comm.from_env(None, timeout_s=30.0)
print("UID", c.rank, seen["uid"].hex(), flush=True)
"""


def _run_world(world, tmp_path, stale=None, rank0_delay=0.0):
    bdir = tmp_path / f"barrier_{world}_{stale}"
    bdir.mkdir()
    path = str(tmp_path / f"rv_{world}_{stale}.id")
    if stale == "r05":                       # bare id of a crashed r05 job
        with open(path, "wb") as f:
            f.write(b"\x07" * 128)
    elif stale == "dead_writer":             # r06 record whose writer is gone (a pid that cannot exist)
        with open(path, "wb") as f:
            f.write(b"\x09" * 128 + struct.pack("<qq", 2 ** 22 + 12345, 42))
    elif stale == "reused_pid":              # ... or whose pid now belongs to another live process (this one, other start time)
        with open(path, "wb") as f:
            f.write(b"\x0b" * 128 + struct.pack("<qq", os.getpid(), 1))
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), QIL_COMM_FILE=path, QIL_TEST_BARRIER_DIR=str(bdir),
                   QIL_TEST_DELAY=str(rank0_delay if r == 0 else 0.0))
        procs.append(subprocess.Popen([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    uids = {}
    for r, p in enumerate(procs):
        out, err = p.communicate(timeout=120)
        assert p.returncode == 0, (r, err[-2000:])
        line = [l for l in out.splitlines() if l.startswith("UID")][0].split()
        uids[int(line[1])] = line[2]
    assert len(uids) == world and len(set(uids.values())) == 1, uids
    assert uids[0] not in ("07" * 128, "09" * 128, "0b" * 128)
    assert not os.path.exists(path)                      # rank 0 removed the record after the collective constructor
    return uids[0]


@pytest.mark.parametrize("world", [2, 8])
def test_rendezvous_all_ranks_agree(world, tmp_path):
    _run_world(world, tmp_path)


@pytest.mark.parametrize("stale", ["r05", "dead_writer", "reused_pid"])
def test_rendezvous_ignores_stale_record(stale, tmp_path):
    """Rank 0 is held back for 0.5 s, so every other rank meets the stale record first; it must wait for the live one."""
    _run_world(3, tmp_path, stale=stale, rank0_delay=0.5)


def test_rendezvous_key_and_record_format(monkeypatch):
    sys.path.insert(0, ROOT)
    import importlib
    sweep = importlib.import_module("qilaplace_jl_amd.sweep")
    monkeypatch.delenv("QIL_COMM_FILE", raising=False)
    monkeypatch.setenv("MASTER_PORT", "29511")
    monkeypatch.setenv("QIL_COMM_TAG", "jobA")
    a = sweep.Comm._rendezvous_path()
    monkeypatch.setenv("QIL_COMM_TAG", "jobB")
    assert a != sweep.Comm._rendezvous_path() and "29511" in a and str(os.getuid()) in a
    assert sweep.Comm._proc_start_ticks(os.getpid()) is not None
    assert sweep.Comm._proc_start_ticks(2 ** 22 + 999) is None
