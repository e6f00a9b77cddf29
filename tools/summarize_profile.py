#!/usr/bin/env python3
"""Condense a tools/profile_gpu.sh output directory into a per-kernel table:
calls, average duration (kernel-trace stats) and PMC counters per dispatch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]


def short(name):
    for key in ("scan_filter12_kernel", "scan_ticket_gate_kernel", "scan_ticket_sort_kernel", "cut_flags_kernel", "cut_offsets_kernel", "kmc_decode_kernel", "pack_rows12_kernel", "scan_filter_kernel", "scan_probe_kernel", "scan_hits_kernel", "scan_bin_gate_kernel", "scan_bin_kernel", "iso_cover_kernel<false>",
                "iso_cover_kernel<true>", "iso_genotype_kernel", "ref_scan_kernel", "rows_kernel", "map_insert_kernel", "genotype_kernel",
                "cover_kernel", "blk_pop_kernel", "summary_kernel"):
        if key in name:
            return key + ("<35,43>" if ("ILi35ELi43E" in name or "<35, 43" in name) else "")
    return name[:60]


print("== kernel-trace stats (rocprofv3 --kernel-trace --stats) ==")
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    print("%-34s %8s %14s %14s %8s" % ("kernel", "calls", "avg_ns", "total_ns", "pct"))
    for r in rows[:12]:
        print("%-34s %8s %14.0f %14.0f %8s" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]), float(r["TotalDurationNs"]), r.get("Percentage", "")))

print()
print("== PMC counters, mean per dispatch (separate passes) ==")
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if not any(s in k for s in ("scan_filter", "scan_probe", "scan_hits", "scan_bin", "iso_cover_kernel<false>", "iso_genotype")):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-24s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))

# ---- HBM traffic of the filter kernel, calibrated (MI355X_MICROARCH.md, HBM: FETCH_SIZE is uncalibrated for
# access widths other than 16 B/lane -> calibrate on a known byte count in the kernel's own pattern) ----
import json
cal = []
for f in glob.glob(os.path.join(out, "cal_FETCH_SIZE", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "scan_filter" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            cal.append(float(r["Counter_Value"]))
k = next((x for x in acc if x.startswith("scan_filter12_kernel")), None) or next((x for x in acc if x.startswith("scan_filter_kernel")), "scan_filter_kernel")
compact = k.startswith("scan_filter12")
if cal and k in acc and "FETCH_SIZE" in acc[k] and "WRITE_SIZE" in acc[k]:
    rows = None
    try:
        rows = json.load(open(os.path.join(out, "bench_stats.json")))["config"]["kmers_per_gpu"]
        bf_bits = json.load(open(os.path.join(out, "bench_stats.json")))["config"]["bf_bits"]
    except Exception:
        pass
    if rows:
        stream = (12.0 if compact else 16.0) * rows      # what the kernel streams per row: packed rows, or hi + lo of the SoA table
        fetch_cal_kb = sum(cal) / len(cal)
        factor = stream / (fetch_cal_kb * 1024.0)
        fetch_kb = sum(acc[k]["FETCH_SIZE"]) / len(acc[k]["FETCH_SIZE"])
        write_kb = sum(acc[k]["WRITE_SIZE"]) / len(acc[k]["WRITE_SIZE"])
        traffic = factor * fetch_kb * 1024.0 + write_kb * 1024.0
        print()
        print("== filter-kernel HBM traffic per launch ==")
        print("stream-only FETCH_SIZE %.6g KB for a known %.6g B stream -> calibration factor %.3f" % (fetch_cal_kb, stream, factor))
        print("FETCH_SIZE %.6g KB, WRITE_SIZE %.6g KB -> %.4g B per launch (algorithmic %.4g B)" % (fetch_kb, write_kb, traffic, 44.0 * rows))
        json.dump({"kernel": k, "units_per_launch": rows, "bf_bits": bf_bits, "hbm_bytes_per_launch": traffic, "fetch_size_kb": fetch_kb,
                   "write_size_kb": write_kb, "fetch_calibration_factor": factor, "calibration": "FETCH_SIZE of the same kernel with only its %d B/row stream (scan_ablate=3)" % (12 if compact else 16),
                   "algorithmic_bytes_per_launch": 44.0 * rows}, open(os.path.join(out, "traffic_scan_filter12.json" if compact else "traffic_scan_filter.json"), "w"), indent=1)
