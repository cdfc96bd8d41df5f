#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM / rocprofv3 section):
WRITE_SIZE and FETCH_SIZE are collected in SEPARATE passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), both in
KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled.  Writes
profiles/<round>_pmc_traffic.json (QIL_ROUND, default r05) = {workload: bytes per launch, lib_sha16: ...}; bench.py reports `roofline.traffic` from it
only while the sha matches the library it runs.

Run on the GPU box from the repo root (each pass is its own rocprofv3 process; the profiled program is python itself):
    python tools/collect_pmc.py            # spawns the two passes, aggregates, writes the json + the two csv summaries
"""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = os.environ.get("QIL_ROUND", "r05")
WORKLOAD = "zt_n24_chi64_D128"
KERNEL = "site_apply_grouped"


def one_pass(counter, outdir):
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "-d", outdir, "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
           "--no-truncate", "--no-configs", "--workload", WORKLOAD]
    subprocess.run(cmd, check=True, env=env, cwd="/tmp", stdout=subprocess.DEVNULL)
    vals = collections.defaultdict(list)
    rows_out = []
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if KERNEL in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals[counter].append(float(r["Counter_Value"]))
                rows_out.append((r["Kernel_Name"], counter, r["Counter_Value"], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""),
                                 r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("SGPR_Count", "")))
    v = vals[counter]
    return sum(v) / max(len(v), 1), len(v), rows_out


def main():
    out = os.path.join(ROOT, "gpurun_out", "pmc")
    w_kb, nw, wrows = one_pass("WRITE_SIZE", os.path.join(out, "write"))
    f_kb, nf, frows = one_pass("FETCH_SIZE", os.path.join(out, "fetch"))
    sys.path.insert(0, ROOT)
    import qilaplace_jl_amd as qil
    sha = hashlib.sha256(open(qil.LIB_PATH, "rb").read()).hexdigest()[:16]
    traffic = w_kb * 1024 + 2 * f_kb * 1024
    rec = {WORKLOAD: traffic, "lib_sha16": sha,
           "_note": "HBM bytes per site_apply_grouped launch = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (gfx950 FETCH_SIZE "
                    "correction, MI355X_MICROARCH.md HBM section), separate --pmc passes, mean over the profiled launches",
           "_write_size_kb": w_kb, "_fetch_size_kb": f_kb, "_launches": [nw, nf]}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for name, rows in ((ROUND + "_pmc_write_site_apply.csv", wrows), (ROUND + "_pmc_fetch_site_apply.csv", frows)):
        with open(os.path.join(ROOT, "gpurun_out", name), "w", newline="") as fh:
            wr = csv.writer(fh)
            wr.writerow(["Kernel_Name", "Counter_Name", "Counter_Value_KB", "Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                         "VGPR_Count", "SGPR_Count"])
            wr.writerows(rows)
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", ROUND + "_pmc_traffic.json"), "w"), indent=1)
    import shutil
    shutil.rmtree(out, ignore_errors=True)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
