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
    for key in ("scan_filter_kernel", "scan_hits_kernel", "call_isolated_kernel", "ref_scan_kernel", "rows_kernel",
                "map_insert_kernel", "genotype_kernel", "cover_kernel", "blk_pop_kernel", "summary_kernel"):
        if key in name:
            return key + ("<35,43>" if "ILi35ELi43E" in name else "")
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
    if not any(s in k for s in ("scan_filter", "scan_hits", "call_isolated")):
        continue
    print(k)
    for c in sorted(acc[k]):
        v = acc[k][c]
        print("    %-24s mean %.6g  (n=%d)" % (c, sum(v) / len(v), len(v)))
