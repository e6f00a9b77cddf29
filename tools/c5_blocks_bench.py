#!/usr/bin/env python3
"""The record loop of BASELINE config C5 (indel / MNP clusters, 8 samples half unphased) by itself, for A/B of tier 2's options.

    python3 tools/c5_blocks_bench.py [--clusters 2.6e5] [--reps 8] [--haploid] [--set name=v,name=v ...]

Every --set is one configuration (comma-separated mg_set_option pairs; "base" = defaults).  Per configuration: the tiers' times
from mg_blocks_stats (HIP events inside the library), the record loop's wall time and a checksum of coverages / GT / GQ, which
must not depend on the configuration.  Under tools/prof_cmd.sh the kernel table gives the per-kernel split.
"""
import argparse
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clusters", type=float, default=2.6e5)
    ap.add_argument("--kmers", type=float, default=2e7)
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--haploid", action="store_true")
    ap.add_argument("--set", action="append", default=[])
    a = ap.parse_args()
    import torch
    import bench
    args = bench.parse_args(["--workload", "c5", "--clusters", str(a.clusters), "--kmers", str(a.kmers), "--cpu-sample", "0", "--plant-records", "2000"])
    job = bench.Job("c5", args, 0, 1, 0, torch, None, haploid=a.haploid)
    ctx = job.ctx
    ctx.counters_reset()
    job.scan(job.n_rows)
    configs = a.set or ["base"]
    first = None
    for cfg in configs:
        pairs = [] if cfg == "base" else [kv.split("=") for kv in cfg.split(",")]
        old = {}
        for name, value in pairs:
            old[name] = ctx.get_option(name)
            ctx.set_option(name, int(value))
        t2, t1, t3, wall = [], [], [], []
        for r in range(a.reps + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            job.rp.cut(ctx)
            job.rp.cover(ctx)
            job.rp.genotype(ctx, probs=False)
            e1.record()
            e1.synchronize()
            st = ctx.blocks_stats()
            if r >= 2:
                t1.append(st[0]); t2.append(st[1]); t3.append(st[2]); wall.append(e0.elapsed_time(e1))
        res = job.rp.results()
        crc = 0
        for key in ("cov", "g1", "g2", "gq", "overflow"):
            if key in res:
                crc = zlib.crc32(np.ascontiguousarray(res[key]).tobytes(), crc)
        first = crc if first is None else first
        print("%-44s tier1 %.3f tier2 %.3f tier3 %.3f loop %.3f ms  general %d kmers %d tier3 %d listed chains %d  crc %08x%s"
              % (cfg, np.mean(t1), np.mean(t2), np.mean(t3), np.mean(wall), st[3], st[5], st[6], ctx.get_option("blocks_listed_chains"), crc,
                 "" if crc == first else "  DIFFERS"), flush=True)
        for name, value in old.items():
            ctx.set_option(name, int(value))
    job.close()


if __name__ == "__main__":
    main()
