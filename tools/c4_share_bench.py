#!/usr/bin/env python3
"""One GPU's share of BASELINE config C4 (3.75e8 of 3e9 k-mers, b=16) against a replicated index of --variants
SNPs: times the scan forms against each other (tickets by gate slice / coarse gate + row partition / direct).
Table per SURVEY 8(d): 20 % of the rows are windows around variant sites (a fifth of those centred = true hits),
the rest uniform random; drawn on the GPU (tests/big_cases.py).  Prints one JSON line per form."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from big_cases import DeviceTable, build_device_index  # noqa: E402
from malva_amd import Context, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--variants", type=float, default=8e7)
ap.add_argument("--rows", type=float, default=3.75e8)
ap.add_argument("--b", type=int, default=16)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--forms", default="tickets,legacy")
ap.add_argument("--layout", default="soa", choices=["soa", "compact"], help="table rows as SoA arrays (20 B) or packed 12-byte rows")
args = ap.parse_args()
K, R = 35, 43
n_vars, n_rows = int(args.variants), int(args.rows)
t0 = time.time()
panel = synth.snp_panel(n_vars, 4242, spacing=40)
plant = min(n_vars, n_rows // 5 * 2 // 15)          # 7.5 windows per planted variant on average -> 20 % of the rows
tab = DeviceTable(panel, n_rows, K, R, 9, plant_variants=plant)
print("[c4] panel + table: %.0f s, %d site rows (%.1f %%)" % (time.time() - t0, tab.n_site, 100.0 * tab.n_site / n_rows), file=sys.stderr)
forms = {"subs": [("use_sub", 1)], "subs_nostore": [("use_sub", 1), ("scan_ablate", 256)],
         "subs_split2": [("use_sub", 1), ("sub_split", 2)], "subs_split8": [("use_sub", 1), ("sub_split", 8)],
         "subs_norec": [("use_sub", 1), ("use_record_counters", 0)], "subs_probe1k": [("use_sub", 1), ("probe_grid", 1024)], "subs_probe4k": [("use_sub", 1), ("probe_grid", 4096)],
         "subs_hits4k": [("use_sub", 1), ("hits_grid", 4096)],
         "tickets": [("use_sub", 0), ("use_tickets", 1)],
         # timing only (wrong results): pass one without its ticket stores / without sorting the tile either
         "tickets_nostore": [("use_sub", 0), ("use_tickets", 1), ("scan_ablate", 256)], "tickets_nosort": [("use_sub", 0), ("use_tickets", 1), ("scan_ablate", 512)],
         "legacy": [("use_sub", 0), ("use_tickets", 0)], "direct": [("use_sub", 0), ("use_tickets", 0), ("use_partition", 0)],
         # smaller fine gates (more false positives, but inside the 256 MiB Infinity Cache): gate_log2 is fixed before the inserts
         "gate30": [("use_pregate", 0), ("gate_log2", 30)], "gate29": [("use_pregate", 0), ("gate_log2", 29)],
         "gate30k2": [("use_pregate", 0), ("gate_k", 2), ("gate_log2", 30)], "gate29k2": [("use_pregate", 0), ("gate_k", 2), ("gate_log2", 29)],
         "gate28k2": [("use_pregate", 0), ("gate_k", 2), ("gate_log2", 28)]}
for form in args.forms.split(","):
    with Context(K, R, args.b << 33) as ctx:
        for name, value in forms[form]:
            ctx.set_option(name, value)
        t0 = time.time()
        build_device_index(ctx, panel, K)
        build_s = time.time() - t0
        ms = []
        d_rows = None
        if args.layout == "compact":
            d_rows = torch.zeros(ctx.kmc_rows_bytes(n_rows) // 4, dtype=torch.int32, device="cuda:0")
            torch.cuda.synchronize()
            ctx.kmc_pack_rows_device(*tab.ptrs(), d_rows.data_ptr())
        for _ in range(args.reps + 1):
            ctx.counters_reset()
            ctx.synchronize()
            t0 = time.perf_counter()
            if d_rows is not None:
                ctx.kmc_scan_rows_device(d_rows.data_ptr(), n_rows)
            else:
                ctx.kmc_scan_device(*tab.ptrs())
            ctx.synchronize()
            ms.append(1e3 * (time.perf_counter() - t0))
        f, p, h, n_open, n_hit = ctx.scan_stats()
        first = min(n_rows, 1 << 27)
        print(json.dumps({"form": form, "layout": args.layout, "variants": n_vars, "rows": n_rows, "b": args.b, "gate_log2": ctx.get_option("gate_log2"),
                          "pregate_k": ctx.get_option("pregate_k"), "scan_tickets": ctx.get_option("scan_tickets"), "scan_subs": ctx.get_option("scan_subs"), "gate_grid": ctx.get_option("ticket_gate_grid"), "scan_bins": ctx.get_option("scan_bins"),
                          "spilled": ctx.get_option("scan_spilled"), "scan_ms_whole_table": round(min(ms[1:]), 3),
                          "chunk_rows": first, "chunk_ms_avg": {"filter": round(f, 3), "probe": round(p, 3), "hits": round(h, 3)},
                          "filter_frac_of_8TBs": round(44 * first / (f * 1e-3) / 8e12, 3),
                          "scan_frac_of_8TBs": round(44 * n_rows / (min(ms[1:]) * 1e-3) / 8e12, 3),
                          "open_rows_last_chunk": n_open, "bf_hit_rows": n_hit, "index_build_s": round(build_s, 1)}), flush=True)
