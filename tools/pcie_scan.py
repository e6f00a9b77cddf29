#!/usr/bin/env python3
"""PCIe-inclusive rates of the host-fed scans (the figures DESIGN.md quotes beside the HBM-resident one; not a bench line):
  mg_kmc_scan          SoA table in host buffers, 20 B per row, pageable and pinned
  mg_kmc_scan_records  raw KMC database records, 10 B per 43-mer, decoded on the device, pageable and pinned
Both stream the table through two staging slots, the upload of one piece beside the scan of the previous one."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from malva_amd import BF_ALT, BF_CTX, Context, capi, synth  # noqa: E402

n_rows, n_vars = int(1e8), int(1e6)
panel = synth.snp_panel(n_vars, seed=20261003)
ctx = Context(35, 43, 4 << 33)
sig, _ = synth.snp_signature_rows(panel, 35)
rows = np.zeros((sig.shape[0], 40), dtype=np.uint8)
rows[:, :35] = sig
ctx.map_insert(rows[0::2]); ctx.bf_insert(BF_ALT, rows[1::2]); ctx.bf_finalize(BF_ALT)
ctx.ref_scan(panel.genome); ctx.bf_finalize(BF_CTX)
hi, lo, cnt = synth.kmer_table(panel, n_rows, 35, 43, seed=777)


def timed(f, what, bytes_per_row):
    f()
    best = 1e9
    for _ in range(3):
        ctx.counters_reset(); ctx.synchronize()
        t0 = time.perf_counter()
        f()
        best = min(best, time.perf_counter() - t0)
    print("%-44s %.1f ms for %d rows = %.3g rows/s = %.1f GB/s over the link" % (what, 1e3 * best, n_rows, n_rows / best, bytes_per_row * n_rows / best / 1e9), flush=True)


timed(lambda: ctx.kmc_scan(hi, lo, cnt), "mg_kmc_scan, pageable host buffers", 20)
bufs, ptrs = [], []
for a in (hi, lo, cnt):
    b, p = capi.host_alloc(a.nbytes)
    b[:] = a.view(np.uint8)
    bufs.append(b.view(a.dtype)); ptrs.append(p)
timed(lambda: ctx.kmc_scan(*bufs), "mg_kmc_scan, pinned host buffers", 20)
for p in ptrs:
    capi.host_free(p)

# the same rows as KMC database records (prefix 7 symbols, 9 suffix bytes, 1 counter byte): sorted by k-mer, one bin
order = np.lexsort((lo, hi))
hs, ls, cs = hi[order], lo[order], cnt[order]
prefix = ((hs << np.uint64(64 - 22)) | (ls >> np.uint64(22))) >> np.uint64(64 - 14)      # top 14 of the 86 bits
lut = np.searchsorted(prefix, np.arange(1 << 14, dtype=np.uint64)).astype(np.uint64)
suffix = ((hs & np.uint64((1 << 8) - 1)).astype(object) << 64 | ls.astype(object)) if False else None
rec = np.zeros((n_rows, 10), dtype=np.uint8)
v_lo = ls
v_hi = hs & np.uint64(0xFF)                      # suffix = low 72 bits: 8 from hi, 64 from lo
rec[:, 0] = v_hi.astype(np.uint8)
for j in range(8):
    rec[:, 1 + j] = ((v_lo >> np.uint64(8 * (7 - j))) & np.uint64(0xFF)).astype(np.uint8)
rec[:, 9] = np.minimum(cs, 255).astype(np.uint8)
ctx.kmc_set_lut(lut, 7, 9, 1, 1, 255, n_rows)
dh, dl, dc = ctx.kmc_decode_records(rec[:100000])
assert np.array_equal(dh, hs[:100000]) and np.array_equal(dl, ls[:100000])
timed(lambda: ctx.kmc_scan_records(rec), "mg_kmc_scan_records, pageable host buffer", 10)
pb, pp = capi.host_alloc(rec.nbytes)
pb[:] = rec.reshape(-1)
timed(lambda: ctx.kmc_scan_records(pb), "mg_kmc_scan_records, pinned host buffer", 10)
capi.host_free(pp)
