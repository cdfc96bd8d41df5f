"""bench.py's contract pieces that need no GPU: the algorithmic-bytes figure the roofline numerator uses,
the workload table, and the N-rank spawn plumbing (the parent must not touch torch or the library)."""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
bench = importlib.import_module("bench")


def test_algorithmic_bytes_match_survey_figures():
    # cfg3: n=24 paired (48 sites), chi_s=64, chi_c=128, c64 output: 80.06 GB + 45.6 MB of operands
    L, paired, chi, D, _ = bench.WORKLOADS["zt_n24_chi64_D128"]
    cb, db = bench.profiles(L, chi, D)
    assert bench.algorithmic_bytes(cb, db) == 80108583232
    out_only = sum(16 * (a * c) * 2 * (b * d) for a, b, c, d in zip([1] + db, db + [1], [1] + cb, cb + [1]))
    assert abs(out_only - 80.06e9) < 0.01e9
    # cfg2: n=20, chi 32, D 64: 1.512 GB written
    L, paired, chi, D, _ = bench.WORKLOADS["qft_n20_chi32_D64"]
    cb, db = bench.profiles(L, chi, D)
    out_only = sum(16 * (a * c) * 2 * (b * d) for a, b, c, d in zip([1] + db, db + [1], [1] + cb, cb + [1]))
    assert abs(out_only - 1.512e9) < 0.001e9


def test_default_workload_is_the_metric_configuration():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'default="zt_n24_chi64_D128"' in src
    assert "dt_sweep_n24_s64" in bench.WORKLOADS


def test_spawn_parent_never_imports_torch_or_the_library():
    """`python bench.py --gpus 2` without a launcher: the parent starts the ranks BEFORE anything GPU-related is imported
    (a process that has initialised the GPU must not spawn/exec).  Here (no GPU) the children fail; the parent must
    relay a non-zero exit code without ever having imported torch or qilaplace_jl_amd itself."""
    code = (
        "import sys, runpy\n"
        "sys.argv = ['bench.py', '--gpus', '2', '--workload', 'tiny', '--steps', '1', '--warmup', '0']\n"
        "try:\n"
        "    runpy.run_path(%r, run_name='__main__')\n"
        "except SystemExit as e:\n"
        "    rc = e.code\n"
        "assert 'torch' not in sys.modules and 'qilaplace_jl_amd' not in sys.modules, sorted(m for m in sys.modules if 'torch' in m)[:3]\n"
        "print('PARENT_OK', rc)\n" % os.path.join(ROOT, "bench.py"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert "PARENT_OK" in r.stdout, r.stdout + r.stderr
    assert r.stdout.strip().split()[-1] not in ("0", "None")          # children cannot initialise a GPU here
