#!/usr/bin/env python3
"""Condense tools/traffic_c4.sh's PMC passes into profiles/traffic_scan_c4.json's content: HBM bytes per launch group (2^27 rows) of the
sub-slice form's three kernels over the whole-genome table.  FETCH_SIZE / WRITE_SIZE are in KB (1024 B)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out = sys.argv[1]
KERNELS = ("scan_sub_sort_kernel", "scan_sub_gate_kernel", "scan_sub_probe_kernel")


def sums(sub, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for k in KERNELS:
                if k in r["Kernel_Name"]:
                    acc[k][0] += float(r["Counter_Value"])
                    acc[k][1] += 1
    return acc


bench = json.load(open(os.path.join(out, "bench_fetch.json")))
cfg = bench["config"]
rows, group = cfg["kmers_per_gpu"], 1 << 27
fetch, write, cal = sums("fetch", "FETCH_SIZE"), sums("write", "WRITE_SIZE"), sums("cal", "FETCH_SIZE")
n_groups = rows / group                                     # launch groups per scan, the last one partial
disp = fetch[KERNELS[0]][1]
scans = disp / -(-rows // group)                             # scans the command ran (timed step + the kernel-time repetitions)
print("rows %d = %.2f launch groups of 2^27; %d dispatches per kernel = %.0f scans" % (rows, n_groups, disp, scans))
# calibration: pass one without its stores reads its row stream and nothing else worth counting (12 B per row, compact rows)
factor = None
if cal[KERNELS[0]][1]:
    fetch_cal = cal[KERNELS[0]][0] * 1024.0 / (cal[KERNELS[0]][1] / -(-rows // group))   # bytes per scan as counted
    factor = 12.0 * rows / fetch_cal
    print("calibration: pass one's row stream counted as %.4g B per scan against %.4g B known -> FETCH_SIZE x %.4f on its 8-byte loads" % (fetch_cal, 12.0 * rows, factor))
parts = {}
for k in KERNELS:
    f = fetch[k][0] * 1024.0 / scans / n_groups              # per launch group of 2^27 rows
    w = write[k][0] * 1024.0 / max(write[k][1] / -(-rows // group), 1) / n_groups
    # pass one and pass two read 8-byte-per-lane streams (rows, tickets): the calibrated factor; the probe kernel's reads are divergent 16-byte
    # pieces of 64-byte records and 12-byte rows: counted as they are (x 1)
    corr = factor if (factor and k != "scan_sub_probe_kernel") else 1.0
    parts[k] = {"fetch_bytes": f, "fetch_correction": corr, "write_bytes": w, "hbm_bytes": f * corr + w}
    print("%-24s FETCH %.4g B x %.3f + WRITE %.4g B = %.4g B per launch group" % (k, f, corr, w, f * corr + w))
total = sum(p["hbm_bytes"] for p in parts.values())
res = {"kernel": "scan_sub_sort_kernel<35,43> + scan_sub_gate_kernel<4> + scan_sub_probe_kernel<35,43>", "rows": rows, "units_per_launch": group, "bf_bits": cfg["bf_bits"],
       "panel_variants": cfg["panel_variants"], "hbm_bytes_per_launch": total, "parts": parts, "algorithmic_bytes_per_launch": 44.0 * group,
       "source": "tools/traffic_c4.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in passes of their own over bench.py --workload c4 (KB = 1024 B), summed over the "
                 "scan's launch groups and divided by rows / 2^27",
       "corrections": "FETCH_SIZE x %s for pass one's and pass two's 8-byte-per-lane streams (factor measured on pass one's own row stream, --scan-ablate 256); x 1 for the probe "
                      "kernel's divergent reads" % ("%.4f" % factor if factor else "1 (calibration pass missing)")}
json.dump(res, open(os.path.join(out, "traffic_scan_c4.json"), "w"), indent=1)
print("total %.4g B per launch group against %.4g B algorithmic" % (total, 44.0 * group))
