#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/<workload>/ (scripts/profile_all.sh) into the tracked files under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summaries, one block per workload (hot-path kernels only)
  profiles/<tag>_pmc_summary.csv    per workload and kernel: median / min / max of every collected counter
  profiles/pmc_traffic.json         per workload, for its dominant kernel: HBM bytes per launch corrected as
                                    MI355X_MICROARCH.md section HBM prescribes (FETCH_SIZE x2 on gfx950; FETCH/WRITE_SIZE
                                    are in KB), VALU wave-instructions per launch, LDS bank-conflict fraction — read by bench.py
"""
import csv
import glob
import json
import os
import statistics
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOT = ("klt_", "hamming_", "match_epilogue", "cosine_", "direct_track", "downsample", "brief_kernel", "harris")
DOMINANT = {"config2": "klt_basic_inverse_pipelined_kernel", "config5_shard": "klt_basic_inverse_pipelined_kernel", "config1": "klt_basic_inverse_pipelined_kernel",
            "config3": "klt_", "config4": "klt_", "hamming": "hamming_match_mfma_kernel<8, false>"}


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    workloads = sorted(d for d in os.listdir(src) if os.path.isdir(os.path.join(src, d)))
    traffic_path = os.path.join(dst, "pmc_traffic.json")
    try:
        traffic = json.load(open(traffic_path))
    except Exception:
        traffic = {}
    traffic.setdefault("workloads", {})
    try:
        build = json.load(open(os.path.join(src, "build_info.json")))  # written by scripts/profile_all.sh on the GPU box
    except Exception:
        build = {}
    with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w", newline="") as fs, open(os.path.join(dst, f"{tag}_pmc_summary.csv"), "w", newline="") as fp:
        ws, wp = csv.writer(fs), csv.writer(fp)
        ws.writerow(["workload", "Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        wp.writerow(["workload", "kernel", "grid", "workgroup", "vgpr", "sgpr", "lds_bytes", "counter", "dispatches", "median", "min", "max"])
        for w in workloads:
            stats = sorted(glob.glob(os.path.join(src, w, "trace", "*", "*_kernel_stats.csv")), key=os.path.getmtime)
            avg_ns = {}
            if stats:
                for r in csv.DictReader(open(stats[-1])):
                    if any(h in r["Name"] for h in HOT):
                        ws.writerow([w] + [r.get(k, "") for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev")])
                        avg_ns[r["Name"]] = float(r["AverageNs"])
            vals = defaultdict(lambda: defaultdict(list))
            meta = {}
            for f in glob.glob(os.path.join(src, w, "pmc_*", "*", "*counter_collection.csv")):
                for r in csv.DictReader(open(f)):
                    k = r["Kernel_Name"]
                    if not any(h in k for h in HOT):
                        continue
                    vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"], r["LDS_Block_Size"])
            for k in sorted(vals):
                for c in sorted(vals[k]):
                    v = vals[k][c]
                    wp.writerow([w, k] + list(meta[k]) + [c, len(v), statistics.median(v), min(v), max(v)])
            dom = [k for k in vals if DOMINANT.get(w, "klt_") in k]
            if dom:
                k = max(dom, key=lambda kk: sum(len(x) for x in vals[kk].values()))
                v = {c: statistics.median(x) for c, x in vals[k].items()}
                entry = {"kernel": k, "source": f"profiles/{tag}_pmc_summary.csv (rocprofv3 --pmc, separate passes; scripts/profile_all.sh {tag})",
                         "source_hash": build.get("source_hash"), "mllvm": build.get("mllvm")}
                if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                    entry["fetch_size_kb_median"], entry["write_size_kb_median"] = v["FETCH_SIZE"], v["WRITE_SIZE"]
                    entry["bytes_per_launch"] = int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
                    entry["bytes_per_launch_uncorrected"] = int((v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024)
                    entry["correction"] = ("bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: gfx950 FETCH_SIZE counts 128-B requests at 64 B "
                                           "(MI355X_MICROARCH.md section HBM; calibrated there for 16 B/lane streams — an upper estimate for this kernel's narrower loads)")
                if "SQ_INSTS_VALU" in v:
                    entry["valu_insts_per_launch"] = int(v["SQ_INSTS_VALU"])
                if "SQ_INSTS_SALU" in v:
                    entry["salu_insts_per_launch"] = int(v["SQ_INSTS_SALU"])
                if "SQ_INSTS_LDS" in v:
                    entry["lds_insts_per_launch"] = int(v["SQ_INSTS_LDS"])
                if v.get("SQ_INSTS_MFMA"):
                    entry["mfma_insts_per_launch"] = int(v["SQ_INSTS_MFMA"])
                    entry["mfma_busy_cycles_per_launch"] = int(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0))
                if v.get("GRBM_GUI_ACTIVE"):
                    entry["gui_active_cycles_sum_over_xcds"] = int(v["GRBM_GUI_ACTIVE"])
                if v.get("SQ_LDS_IDX_ACTIVE"):
                    entry["lds_bank_conflict_frac"] = v.get("SQ_LDS_BANK_CONFLICT", 0.0) / v["SQ_LDS_IDX_ACTIVE"]
                if v.get("SQ_WAVE_CYCLES"):
                    entry["wait_any_frac_of_wave_cycles"] = v.get("SQ_WAIT_ANY", 0.0) / v["SQ_WAVE_CYCLES"]
                if v.get("TCC_HIT_sum") is not None and (v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0)) > 0:
                    entry["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
                name = [n for n in avg_ns if DOMINANT.get(w, "klt_") in n]
                if name:
                    entry["trace_average_ns"] = avg_ns[max(name, key=lambda n: avg_ns[n])]
                traffic["workloads"][w] = entry
    if "config2" in traffic["workloads"] and "bytes_per_launch" in traffic["workloads"]["config2"]:
        traffic["klt_config2_bytes_per_launch"] = traffic["workloads"]["config2"]["bytes_per_launch"]  # round-1 key, kept for older readers
    with open(traffic_path, "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic["workloads"], indent=1))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r2")
