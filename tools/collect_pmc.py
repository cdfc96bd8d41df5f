#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from rocprofv3 PMC passes (MI355X_MICROARCH.md, HBM / rocprofv3 section):
WRITE_SIZE and FETCH_SIZE are collected in SEPARATE passes (TCC has 4 slots: FETCH_SIZE costs 3, WRITE_SIZE 2), both in
KB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so it is doubled.  Writes
profiles/<round>_pmc_traffic.json (QIL_ROUND, default r06) = {workload: bytes per launch / per repetition, lib_sha16: ...}: the headline
apply, the cfg2 apply, the 64-query read-out on the 80 GB product and the n = 30 i.i.d. encode -- every roofline of SURVEY.md 8(d);
bench.py / bench_configs.py report `roofline.traffic` from it only while the sha matches the library they run.

Run on the GPU box from the repo root (each pass is its own rocprofv3 process; the profiled program is python itself):
    python tools/collect_pmc.py            # spawns the passes (12 rocprofv3 processes), aggregates, writes the json + the two csv summaries
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
ROUND = os.environ.get("QIL_ROUND", "r06")
KERNEL = "site_apply_grouped"
# per-LAUNCH traffic of the apply kernel (mean over the profiled launches): the headline workload and cfg2
APPLY_WORKLOADS = ["zt_n24_chi64_D128", "qft_n20_chi32_D64"]
# per-REPETITION traffic of whole operations made of many launches (the other rooflines of SURVEY 8d): every
# dispatch of the process summed, as the difference of a 3-repetition and a 1-repetition run (set-up and warm-up cancel)
OPERATIONS = {"coefficient_batch_64_cfg3": ("_coeff_cfg3.py", lambda r: [str(r)]),
              "encode_n30_random_k128": ("_prof_encode30.py", lambda r: [str(r)])}


def run_pass(counter, outdir, argv):
    """One rocprofv3 --pmc process; returns [(kernel name, counter value, row)] for `counter`."""
    shutil.rmtree(outdir, ignore_errors=True)
    os.makedirs(outdir, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = ["rocprofv3", "--pmc", counter, "-d", outdir, "--output-format", "csv", "--", sys.executable] + argv
    for attempt in range(3):
        if subprocess.run(cmd, env=env, cwd="/tmp", stdout=subprocess.DEVNULL).returncode == 0:
            break
        shutil.rmtree(outdir, ignore_errors=True)
        os.makedirs(outdir, exist_ok=True)
    else:
        raise RuntimeError("rocprofv3 --pmc failed three times: " + " ".join(argv))
    rows = []
    for f in glob.glob(os.path.join(outdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                rows.append((r["Kernel_Name"], float(r["Counter_Value"]), r))
    shutil.rmtree(outdir, ignore_errors=True)
    return rows


def apply_pass(counter, outdir, workload):
    rows = run_pass(counter, outdir, [os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2", "--no-cpu-baseline",
                                      "--no-truncate", "--no-configs", "--workload", workload])
    sel = [(k, v, r) for k, v, r in rows if KERNEL in k]
    rows_out = [(k, counter, r["Counter_Value"], r.get("Grid_Size", ""), r.get("Workgroup_Size", ""), r.get("LDS_Block_Size", ""),
                 r.get("VGPR_Count", ""), r.get("SGPR_Count", "")) for k, v, r in sel]
    return sum(v for _, v, _ in sel) / max(len(sel), 1), len(sel), rows_out


def operation_pass(counter, outdir, script, args):
    """KB per repetition and the top kernels of the difference."""
    tot, by_k = {}, {}
    for reps in (3, 1):
        rows = run_pass(counter, outdir, [os.path.join(ROOT, "tools", script)] + args(reps))
        tot[reps] = sum(v for _, v, _ in rows)
        d = collections.defaultdict(float)
        for k, v, _ in rows:
            m = re.search(r"qil_k[1n]<\s*(?:\(anonymous namespace\)::)?(\w+)", k)
            d[m.group(1) if m else k.split("(")[0][-48:]] += v
        by_k[reps] = d
    per = (tot[3] - tot[1]) / 2.0
    top = {k: (by_k[3][k] - by_k[1].get(k, 0.0)) / 2.0 * 1024 for k in by_k[3]}
    return per, dict(sorted(((k, v) for k, v in top.items() if v > 0), key=lambda kv: -kv[1])[:6])


def main():
    out = os.path.join(ROOT, "gpurun_out", "pmc")
    sys.path.insert(0, ROOT)
    import qilaplace_jl_amd as qil
    sha = hashlib.sha256(open(qil.LIB_PATH, "rb").read()).hexdigest()[:16]
    rec = {"lib_sha16": sha,
           "_note": "HBM bytes = WRITE_SIZE*1024 + 2*FETCH_SIZE*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md HBM section), "
                    "separate --pmc passes.  Apply workloads: per site_apply_grouped launch, mean over the profiled launches.  "
                    "Operations (read-out, encode): per repetition, every dispatch summed, (3-repetition run - 1-repetition run) / 2"}
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    for wl in APPLY_WORKLOADS:
        w_kb, nw, wrows = apply_pass("WRITE_SIZE", os.path.join(out, "write"), wl)
        f_kb, nf, frows = apply_pass("FETCH_SIZE", os.path.join(out, "fetch"), wl)
        rec[wl] = w_kb * 1024 + 2 * f_kb * 1024
        rec["_detail_" + wl] = {"write_size_kb": w_kb, "fetch_size_kb": f_kb, "launches": [nw, nf]}
        if wl == APPLY_WORKLOADS[0]:
            rec["_write_size_kb"], rec["_fetch_size_kb"], rec["_launches"] = w_kb, f_kb, [nw, nf]
            for name, rows in ((ROUND + "_pmc_write_site_apply.csv", wrows), (ROUND + "_pmc_fetch_site_apply.csv", frows)):
                with open(os.path.join(ROOT, "gpurun_out", name), "w", newline="") as fh:
                    wr = csv.writer(fh)
                    wr.writerow(["Kernel_Name", "Counter_Name", "Counter_Value_KB", "Grid_Size", "Workgroup_Size", "LDS_Block_Size",
                                 "VGPR_Count", "SGPR_Count"])
                    wr.writerows(rows)
    for key, (script, args) in OPERATIONS.items():
        w_kb, wtop = operation_pass("WRITE_SIZE", os.path.join(out, "write"), script, args)
        f_kb, ftop = operation_pass("FETCH_SIZE", os.path.join(out, "fetch"), script, args)
        rec[key] = w_kb * 1024 + 2 * f_kb * 1024
        rec["_detail_" + key] = {"write_size_kb": w_kb, "fetch_size_kb": f_kb, "write_bytes_top_kernels": wtop,
                                 "fetch_bytes_top_kernels_uncorrected": ftop}
    json.dump(rec, open(os.path.join(ROOT, "gpurun_out", ROUND + "_pmc_traffic.json"), "w"), indent=1)
    shutil.rmtree(out, ignore_errors=True)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
