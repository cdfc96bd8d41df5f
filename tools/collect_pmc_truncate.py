#!/usr/bin/env python3
"""Matrix-core work of the truncate half from rocprofv3 PMC passes (not a flop model): SQ_INSTS_VALU_MFMA_MOPS_F64 (x 512 =
f64 MFMA flops), SQ_VALU_MFMA_BUSY_CYCLES and SQ_INSTS_VALU summed over every dispatch of
  * the exact route of the bench's truncate block, compress!(W_zt psi; maxdim=64, tol=1e-8) on the bond-1008 product
    (tools/_exact_compress_time.py), and
  * compress! chi 256 -> 128 on 24 sites (tools/_compress_one.py 256 f64), and
  * (r04) ONE qil_apply_compress_batch of the 64 (zT operator, signal) pairs of a damping sweep (tools/_apply_compress_batch64.py),
and (r05, the two other rooflines of SURVEY.md 8d) the 64-query coefficient_batch on an 80 GB cfg3-shaped product
(tools/_coeff_cfg3.py) and signal_ztmps(:rsvd, k=128) of 2^30 i.i.d. samples (tools/_prof_encode30.py),
each as the DIFFERENCE of a 3-repetition and a 1-repetition run (the set-up -- encode, MPO build, warm-up -- cancels), per
repetition.  Writes gpurun_out/<round>_pmc_truncate.json (round tag: QIL_ROUND, default r05) keyed to the library's sha256 (copy it to profiles/): bench.py reports
`truncate.roofline` from it only while the sha matches the library it runs.  PMC passes serialise the kernels, so times come
from the un-profiled bench run, never from here.

    python tools/collect_pmc_truncate.py          (on the GPU box, from the repo root; four rocprofv3 processes)
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("QIL_ROUND", "r05")
COUNTERS = ["SQ_INSTS_VALU_MFMA_MOPS_F64", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_INSTS_MFMA"]


def one_pass(tag, script_args):
    out = os.path.join(ROOT, "gpurun_out", "pmc_trunc", tag)
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(out, exist_ok=True)
    cmd = ["rocprofv3", "--pmc"] + COUNTERS + ["-d", out, "--output-format", "csv", "--", sys.executable] + script_args
    for attempt in range(4):          # (the profiler's interception layer crashes on multi-threaded batches about one run in three)
        shutil.rmtree(out, ignore_errors=True)
        os.makedirs(out, exist_ok=True)
        if subprocess.run(cmd, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp", stdout=subprocess.DEVNULL).returncode == 0:
            break
    else:
        raise RuntimeError("rocprofv3 --pmc failed four times: " + " ".join(script_args))
    tot = collections.defaultdict(float)
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = 0
    for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            v = float(r["Counter_Value"])
            tot[r["Counter_Name"]] += v
            m = re.search(r"qil_k[1n]<\s*(?:\(anonymous namespace\)::)?(\w+)", r["Kernel_Name"])
            name = m.group(1) if m else r["Kernel_Name"].split("(")[0][-60:]
            per_kernel[name][r["Counter_Name"]] += v
            disp += r["Counter_Name"] == COUNTERS[0]
    shutil.rmtree(out, ignore_errors=True)
    return dict(tot), disp, per_kernel


def workload(name, script, args):
    t3, d3, k3 = one_pass(name + "_3", [os.path.join(ROOT, "tools", script)] + args(3))
    t1, d1, k1 = one_pass(name + "_1", [os.path.join(ROOT, "tools", script)] + args(1))
    per = {c: (t3.get(c, 0.0) - t1.get(c, 0.0)) / 2.0 for c in COUNTERS}
    kern = {}
    for k in k3:
        m = (k3[k].get(COUNTERS[0], 0.0) - k1.get(k, {}).get(COUNTERS[0], 0.0)) / 2.0 * 512.0
        if m > 0:
            kern[k] = m
    top = dict(sorted(kern.items(), key=lambda kv: -kv[1])[:8])
    return {"mfma_f64_flops": per[COUNTERS[0]] * 512.0, "mfma_busy_cycles": per[COUNTERS[1]], "valu_insts": per[COUNTERS[2]],
            "mfma_insts": per[COUNTERS[3]], "dispatches": (d3 - d1) / 2.0, "mfma_f64_flops_by_kernel_top8": top}


def main():
    sys.path.insert(0, ROOT)
    import qilaplace_jl_amd as qil
    sha = hashlib.sha256(open(qil.LIB_PATH, "rb").read()).hexdigest()[:16]
    rec = {"lib_sha16": sha,
           "_note": "per repetition, (3-repetition run - 1-repetition run) / 2; flops = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512",
           "exact_compress_product_1008": workload("exact", "_exact_compress_time.py", lambda r: [str(r)]),
           "compress_chi256_24_sites": workload("chi256", "_compress_one.py", lambda r: ["256", "f64", str(r)]),
           # the batch as the operating point of the truncate half: 64 (zT operator, signal) pairs of a damping sweep through
           # ONE qil_apply_compress_batch (tools/_apply_compress_batch64.py 64 zt <repetitions>)
           "apply_compress_batch64_zt": workload("batch64", "_apply_compress_batch64.py", lambda r: ["64", "zt", str(r)]),
           # the read-out and the RSVD encode, counted
           "coefficient_batch_64_cfg3": workload("coeff64", "_coeff_cfg3.py", lambda r: [str(r)]),
           "encode_n30_random_k128": workload("encode30", "_prof_encode30.py", lambda r: [str(r)])}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", ROUND + "_pmc_truncate.json"), "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
