import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
from big_cases import DeviceTable, build_device_index, counters_tensor
from malva_amd import BF_ALT, BF_CTX, Context, synth
from oracle import capi as ocapi
K, R = 35, 43
n_vars, n_rows, bits = 100000, 10_000_000, 1 << 33
panel = synth.snp_panel(n_vars, 5)
tab = DeviceTable(panel, n_rows, K, R, 7)
ctx = Context(K, R, bits)
build_device_index(ctx, panel, K)
counters, n_bf, n_map = counters_tensor(ctx)
ctx.kmc_scan_device(*tab.ptrs()); ctx.synchronize()
got = counters.cpu().numpy().view(np.uint32).astype(np.int64)
ref, alt = tab.expected_sums(panel)
h = K // 2
w = synth.windows(panel.genome, panel.pos - h, K).copy(); w[:, h] = panel.pool[1::2]
rows = np.zeros((n_vars, 40), dtype=np.uint8); rows[:, :K] = w
slot = ctx.bf_index(BF_ALT, rows)
pos = ctx.bf_export_sparse(BF_ALT)[2]
rank = np.searchsorted(pos, slot)
want = np.zeros(n_bf, dtype=np.int64); np.add.at(want, rank, alt)
bad = np.flatnonzero((got[:n_bf] & 0xFFFF) != (want & 0xFFFF))
print("bad", bad.size, "of", n_bf)
inv = np.zeros(n_bf, dtype=np.int64); inv[rank] = np.arange(n_vars)
for r in bad[:8]:
    v = inv[r]
    print("rank", r, "var", v, "got", got[r], "want", want[r], "donor", panel.donor_gt[v], "ref", chr(panel.pool[2*v]), "alt", chr(panel.pool[2*v+1]))
    sel = np.flatnonzero(tab.site_var == v)
    print("   planted:", [(int(tab.site_hap[i]), int(tab.site_off[i]), int(tab.site_cnt[i])) for i in sel])
# oracle on the same table
obf, octx, omap = ocapi.BF(bits), ocapi.BF(bits), ocapi.KMAP()
sig, _ = synth.snp_signature_rows(panel, K)
r2 = np.zeros((sig.shape[0], 40), dtype=np.uint8); r2[:, :K] = sig
isr = np.zeros(r2.shape[0], dtype=np.uint8); isr[0::2] = 1
ocapi.add_kmers(obf, omap, r2, isr); obf.switch_mode(); ocapi.ref_scan(obf, octx, panel.genome.tobytes(), K, R); octx.switch_mode()
hi, lo, cnt = tab.host(0, n_rows)
ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, K, R)
oc = obf.counts().astype(np.int64)
print("device == oracle:", np.array_equal(got[:n_bf] & 0xFFFF, oc), "oracle bad vs want:", int(((oc & 0xFFFF) != (want & 0xFFFF)).sum()))
