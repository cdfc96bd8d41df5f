"""The lock-step combiner of the batch entry points (qilaplace.jl_amd/csrc/qil_lockstep_core.h: rings, launcher loop, park / wake
of chain threads waiting for a read-back) under ThreadSanitizer on the CPU: the header is HIP-free, tests/lockstep_stress.cpp binds
it to a stub launch function and drives it with up to 64 producer threads -- random progress keys, random kernel classes and LDS
sizes, read-back waits, ring drains, launches that fail in the middle.  The product compiles the same header into libqilhip.so
(qil_context.hip); GPU sanitizers are not available on the pool, so this is where a protocol race would show."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def stress_binary(tmp_path_factory):
    cxx = shutil.which("g++")
    if not cxx:
        pytest.skip("no g++")
    exe = str(tmp_path_factory.mktemp("lockstep") / "lockstep_stress")
    r = subprocess.run([cxx, "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-pthread",
                        "-I", os.path.join(ROOT, "qilaplace.jl_amd", "csrc"), os.path.join(ROOT, "tests", "lockstep_stress.cpp"),
                        "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.parametrize("chains,groups,steps,seed,delay_us", [
    (64, 4, 200, 1, 0),        # the shape of a 64-chain batch: four groups of 16
    (64, 4, 150, 2, 5),        # ... with a "device" that takes 5 us per launch (read-backs arrive while chains are parked)
    (37, 3, 200, 4, 2),        # uneven groups
    (16, 1, 300, 3, 0),        # one group, one launcher
    (8, 4, 300, 5, 0),         # two chains per group
])
def test_combiner_under_thread_sanitizer(stress_binary, chains, groups, steps, seed, delay_us):
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66")
    r = subprocess.run([stress_binary, str(chains), str(groups), str(steps), str(seed), str(delay_us)], capture_output=True,
                       text=True, timeout=300, env=env)
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    assert " 0 errors" in r.stdout, r.stdout


def test_product_uses_the_same_header():
    src = open(os.path.join(ROOT, "qilaplace.jl_amd", "csrc", "qil_context.hip")).read()
    assert '#include "qil_lockstep_core.h"' in src
    for fn in ("qil_ls_run", "qil_ls_park", "qil_ls_begin", "qil_ls_commit", "qil_ls_drain", "qil_ls_set_key"):
        assert fn + "(" in src, fn
