#!/usr/bin/env python3
"""Condense tools/traffic_blocks.sh's (or traffic_c4.sh's) PMC passes into profiles/traffic_blocks_<workload>.json's content: HBM bytes per
record loop (main.cpp:522-579 on the resident panel: cut, tiers 1-3, likelihoods), kernel by kernel.  FETCH_SIZE / WRITE_SIZE are in KB
(1024 B).  FETCH_SIZE is taken as counted (x 1): the loop's reads are mostly divergent pieces of records, rows and panel arrays, for which the
counter is uncalibrated; on gfx950 it counts a wide coalesced stream at half its bytes (MI355X_MICROARCH.md), so the figure is a lower
bound where a kernel streams (cut_flags, genotype: 16-40 B per record).
Which dispatches belong to a call-time loop: the call-time instantiations only (<0>, <false>, <true>; not <1> / <2> / *_index_*); the loops
the command ran = dispatches of fw_finish_kernel (one per mg_cover_blocks_device); the block cut's kernels also run once at index time, so
theirs is the mean per dispatch; of genotype_kernel's dispatches (bench.py's kernel timing runs the -v form too) the GT / GQ-only ones are
those that write least."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

out, wl = sys.argv[1], sys.argv[2]
CUT = ("cut_flags_kernel", "tile_reduce_kernel", "part_scan_kernel", "tile_rescan_kernel", "flag_scatter_kernel")
LOOP = ("panel_lone_kernel", "fw_walk_kernel", "fw_snp_kernel", "fw_order_kernel", "fw_chain_kernel", "fw_picks_kernel", "fw_eval_kernel", "fw_slide_kernel", "fb_compact_kernel",
        "cover_blocks_kernel", "fw_finish_kernel", "ref_pack_kernel")


def rows_of(sub, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            n = r["Kernel_Name"]
            if "index" in n or "<1>" in n or "<2>" in n:
                continue
            for k in CUT + LOOP + ("genotype_kernel",):
                if k + "(" in n or k + "<" in n:
                    acc[k].append(float(r["Counter_Value"]) * 1024.0)
    return acc


bench = json.load(open(os.path.join(out, "bench_fetch.json")))
fetch, write = rows_of("fetch", "FETCH_SIZE"), rows_of("write", "WRITE_SIZE")
loops = len(fetch["fw_finish_kernel"])
print("call-time record loops in the command: %d" % loops)
parts, total = {}, 0.0
for k in sorted(fetch):
    f, w = fetch[k], write.get(k, [])
    if k in CUT:                     # one cut per loop (its kernels also run at index time): per dispatch x dispatches per cut
        per = len(f) / max(len(fetch["cut_flags_kernel"]), 1)
        fb, wb = sum(f) / len(f) * per, (sum(w) / len(w) * per if w else 0.0)
    elif k == "genotype_kernel":     # the GT / GQ-only dispatches: the half that writes least
        nk = max(1, min(loops, len(f)))
        fb, wb = sum(sorted(f)[:nk]) / nk, (sum(sorted(w)[:nk]) / nk if w else 0.0)
    else:
        fb, wb = sum(f) / max(loops, 1), sum(w) / max(loops, 1)
    parts[k] = {"fetch_bytes": fb, "write_bytes": wb, "dispatches": len(f)}
    total += fb + wb
    print("%-24s FETCH %.4g B + WRITE %.4g B  (%d dispatches)" % (k, fb, wb, len(f)))
rb = bench.get("roofline_blocks") or {}
res = {"workload": wl, "hbm_bytes_per_launch": total, "parts": parts, "algorithmic_bytes_per_launch": rb.get("algorithmic_bytes_per_launch"),
       "units_per_launch": rb.get("units_per_launch"), "panel_variants": bench["config"]["panel_variants"], "haploid_included": wl == "c5",
       "source": "tools/traffic_blocks.py over rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes (each in a pass of its own) of bench.py --workload %s (KB = 1024 B), per record loop" % wl,
       "corrections": "FETCH_SIZE x 1 (divergent reads; a lower bound where a kernel streams: gfx950 counts wide coalesced streams at half their bytes)"}
json.dump(res, open(os.path.join(out, "traffic_blocks_%s.json" % wl), "w"), indent=1)
print("total %.4g B per record loop%s" % (total, (" against %.4g B algorithmic" % rb["algorithmic_bytes_per_launch"]) if rb.get("algorithmic_bytes_per_launch") else ""))
