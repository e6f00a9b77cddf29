#!/usr/bin/env python3
"""Condense tools/traffic_blocks.sh's PMC passes into profiles/traffic_blocks_<workload>.json's content: HBM bytes per record loop
(main.cpp:522-579 on the resident panel: cut, tiers 1-3, likelihoods), kernel by kernel.  FETCH_SIZE / WRITE_SIZE are in KB (1024 B).
FETCH_SIZE is taken as counted (x 1): the loop's reads are mostly divergent pieces of records, rows and panel arrays, for which the
counter is uncalibrated; on gfx950 it counts a wide coalesced stream at half its bytes (MI355X_MICROARCH.md), so the figure is a lower bound
where a kernel streams (cut_flags, genotype: 16-40 B per record)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, wl = sys.argv[1], sys.argv[2]
LOOP = ("cut_flags_kernel", "tile_reduce_kernel", "part_scan_kernel", "tile_rescan_kernel", "flag_scatter_kernel", "panel_lone_kernel", "fw_walk_kernel", "fw_order_kernel",
        "fw_picks_kernel", "fw_eval_kernel", "fw_slide_kernel", "fb_compact_kernel", "cover_blocks_kernel", "fw_finish_kernel", "genotype_kernel", "ref_pack_kernel")


def sums(sub, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for k in LOOP:
                if k in r["Kernel_Name"] and "index" not in r["Kernel_Name"]:
                    name = k + ("<2>" if ("<2>" in r["Kernel_Name"] or "<1>" in r["Kernel_Name"]) and k.startswith("fw_") else "")
                    acc[name][0] += float(r["Counter_Value"])
                    acc[name][1] += 1
    return acc


bench = json.load(open(os.path.join(out, "bench_fetch.json")))
fetch, write = sums("fetch", "FETCH_SIZE"), sums("write", "WRITE_SIZE")
loops = fetch["cut_flags_kernel"][1]                      # one block cut per record loop the command ran (c5: diploid + haploid jobs)
print("record loops in the command: %d" % loops)
parts, total = {}, 0.0
for k in sorted(fetch):
    if k.endswith("<2>"):                                 # index-time instantiations (MODE 1 / 2): not the call-time loop
        continue
    f = fetch[k][0] * 1024.0 / loops
    w = write[k][0] * 1024.0 / max(loops, 1) if k in write else 0.0
    if k == "genotype_kernel":                            # (bench.py's kernel timing runs the -v form once more per repetition: per dispatch, one per loop)
        f = fetch[k][0] * 1024.0 / fetch[k][1]
        w = write[k][0] * 1024.0 / max(write[k][1], 1)
    parts[k] = {"fetch_bytes": f, "write_bytes": w, "dispatches_per_loop": fetch[k][1] / loops}
    total += f + w
    print("%-24s FETCH %.4g B + WRITE %.4g B  (%.1f dispatches per loop)" % (k, f, w, fetch[k][1] / loops))
rb = bench.get("roofline_blocks") or {}
res = {"workload": wl, "hbm_bytes_per_launch": total, "parts": parts, "algorithmic_bytes_per_launch": rb.get("algorithmic_bytes_per_launch"),
       "units_per_launch": rb.get("units_per_launch"), "panel_variants": bench["config"]["panel_variants"], "haploid_included": wl == "c5",
       "source": "tools/traffic_blocks.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in passes of their own over bench.py --workload %s (KB = 1024 B), per record loop" % wl,
       "corrections": "FETCH_SIZE x 1 (divergent reads; a lower bound where a kernel streams: gfx950 counts wide coalesced streams at half their bytes)"}
json.dump(res, open(os.path.join(out, "traffic_blocks_%s.json" % wl), "w"), indent=1)
print("total %.4g B per record loop%s" % (total, (" against %.4g B algorithmic" % rb["algorithmic_bytes_per_launch"]) if rb.get("algorithmic_bytes_per_launch") else ""))
