#!/usr/bin/env python3
"""Rate of mg_kmc_scan when the table arrives in HOST buffers (pageable numpy arrays -> hipMemcpy -> scan):
the PCIe-inclusive figure DESIGN.md quotes beside the HBM-resident one.  Not a bench line."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (first HIP runtime in the process)
from malva_amd import BF_ALT, BF_CTX, Context, synth  # noqa: E402

n_rows, n_vars = int(1e8), int(1e6)
panel = synth.snp_panel(n_vars, seed=20261003)
ctx = Context(35, 43, 4 << 33)
sig, _ = synth.snp_signature_rows(panel, 35)
rows = np.zeros((sig.shape[0], 40), dtype=np.uint8)
rows[:, :35] = sig
ctx.map_insert(rows[0::2]); ctx.bf_insert(BF_ALT, rows[1::2]); ctx.bf_finalize(BF_ALT)
ctx.ref_scan(panel.genome.tobytes()); ctx.bf_finalize(BF_CTX)
hi, lo, cnt = synth.kmer_table(panel, n_rows, 35, 43, seed=777)
ctx.kmc_scan(hi[:1000], lo[:1000], cnt[:1000])
best = 1e9
for _ in range(3):
    ctx.counters_reset(); ctx.synchronize()
    t0 = time.perf_counter()
    ctx.kmc_scan(hi, lo, cnt)
    best = min(best, time.perf_counter() - t0)
print("mg_kmc_scan from host buffers: %.3f s for %d rows = %.3g rows/s = %.2f GB/s of table" % (best, n_rows, n_rows / best, 20 * n_rows / best / 1e9))
