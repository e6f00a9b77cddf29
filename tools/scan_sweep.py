#!/usr/bin/env python3
"""A/B the scan kernels' launch parameters in ONE process on one index and one table
(cdna_hip_programming.md rule 24): interleaved rounds, median + min per variant.
usage: python tools/scan_sweep.py [--kmers 1e8] [--variants 1e6] opt=v1,v2 [opt2=...]"""
import itertools
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from malva_amd import BF_ALT, BF_CTX, Context, synth  # noqa: E402


def main():
    args = sys.argv[1:]
    n_rows, n_vars, b, rounds = int(1e8), int(1e6), 4, 7
    sweeps = {}
    while args:
        a = args.pop(0)
        if a == "--kmers":
            n_rows = int(float(args.pop(0)))
        elif a == "--variants":
            n_vars = int(float(args.pop(0)))
        elif a == "--b":
            b = int(args.pop(0))
        elif a == "--rounds":
            rounds = int(args.pop(0))
        else:
            k, v = a.split("=")
            sweeps[k] = [int(x) for x in v.split(",")]
    K, R = 35, 43
    dev = torch.device("cuda", 0)
    panel = synth.snp_panel(n_vars, seed=20261003)
    ctx = Context(K, R, b << 33)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    for k in [k for k in sweeps if k.startswith("gate_")]:      # index-time options: one value, before any insert
        ctx.set_option(k, sweeps.pop(k)[0])
    sig, _ = synth.snp_signature_rows(panel, K)
    rows = np.zeros((sig.shape[0], 40), dtype=np.uint8)
    rows[:, :K] = sig
    ctx.map_insert(rows[0::2]); ctx.bf_insert(BF_ALT, rows[1::2]); ctx.bf_finalize(BF_ALT)
    ctx.ref_scan(panel.genome.tobytes()); ctx.bf_finalize(BF_CTX)
    hi, lo, cnt = synth.kmer_table(panel, n_rows, K, R, seed=777)
    d_hi = torch.from_numpy(hi.view(np.int64)).to(dev)
    d_lo = torch.from_numpy(lo.view(np.int64)).to(dev)
    d_cnt = torch.from_numpy(cnt.view(np.int32)).to(dev)
    keys = list(sweeps)
    combos = list(itertools.product(*[sweeps[k] for k in keys])) or [()]
    times = {c: [] for c in combos}
    ref = None
    for rnd in range(rounds + 1):
        for c in combos:
            for k, v in zip(keys, c):
                ctx.set_option(k, v)
            ctx.counters_reset()
            ctx.kmc_scan_device(d_hi.data_ptr(), d_lo.data_ptr(), d_cnt.data_ptr(), n_rows)
            f, pr, h, n_open, nh = ctx.scan_stats()
            if rnd:
                times[c].append((f, pr, h, n_open, nh))
            else:   # results must not depend on the launch parameters
                d = torch.zeros(sum(ctx.counters_size()), dtype=torch.int32, device=dev)
                ctx.counters_export_device(d.data_ptr()); ctx.synchronize()
                ref = d if ref is None else ref
                assert "scan_ablate" in keys or torch.equal(ref, d), "counters differ for %s" % (c,)
    for c in combos:
        f = np.array([t[0] for t in times[c]]); pr = np.array([t[1] for t in times[c]]); h = np.array([t[2] for t in times[c]])
        print("%-36s filter median %.3f min %.3f | probe %.3f | hits %.3f ms | open %d hit %d | filter %.3g kmers/s, all %.3g" %
              (dict(zip(keys, c)), np.median(f), f.min(), np.median(pr), np.median(h), times[c][-1][3], times[c][-1][4],
               n_rows / (np.median(f) * 1e-3), n_rows / (np.median(f + pr + h) * 1e-3)), flush=True)


if __name__ == "__main__":
    main()
