#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (scripts/profile_bench.sh) into the tracked files under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (verbatim)
  profiles/<tag>_pmc_summary.csv    per-kernel median of every collected counter
  profiles/pmc_traffic.json         HBM bytes per launch of the tracker kernel, corrected as
                                    MI355X_MICROARCH.md §HBM prescribes (FETCH_SIZE x2 on gfx950;
                                    FETCH/WRITE_SIZE are in KB) — read by bench.py
"""
import csv
import glob
import json
import os
import shutil
import statistics
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    stats = sorted(glob.glob(os.path.join(src, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
    if stats:
        shutil.copy(stats[-1], os.path.join(dst, f"{tag}_kernel_stats.csv"))  # the newest run (gpurun_out keeps earlier ones)
    vals = defaultdict(lambda: defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(src, "pmc_*", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
    with open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid", "workgroup", "counter", "dispatches", "median", "min", "max"])
        for k in sorted(vals):
            if "at::native" in k or "rocclr" in k:
                continue
            for c in sorted(vals[k]):
                v = vals[k][c]
                w.writerow([k, meta[k][0], meta[k][1], c, len(v), statistics.median(v), min(v), max(v)])
    klt = [k for k in vals if "klt_basic_inverse_pipelined_kernel" in k] or [k for k in vals if "klt_track_kernel" in k]
    if klt:
        v = vals[klt[0]]
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fetch_kb, write_kb = statistics.median(v["FETCH_SIZE"]), statistics.median(v["WRITE_SIZE"])
            out = {
                "source": f"profiles/{tag}_pmc_summary.csv (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes)",
                "kernel": klt[0],
                "fetch_size_kb_median": fetch_kb, "write_size_kb_median": write_kb,
                "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE counts 128-B requests at 64 B "
                              "(MI355X_MICROARCH.md §HBM); calibrated there for 16 B/lane streams, this kernel issues unaligned "
                              "8 B/lane window loads, so the x2 makes this an upper estimate",
                "klt_config2_bytes_per_launch": int((2 * fetch_kb + write_kb) * 1024),
                "klt_config2_bytes_per_launch_uncorrected": int((fetch_kb + write_kb) * 1024),
            }
            with open(os.path.join(dst, "pmc_traffic.json"), "w") as f:
                json.dump(out, f, indent=1)
            print(json.dumps(out, indent=1))
    print(open(os.path.join(dst, f"{tag}_kernel_stats.csv")).read()[:600])


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r1")
