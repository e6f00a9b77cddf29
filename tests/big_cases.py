"""BASELINE-sized inputs for the -m gpu config tests.  Tables of 1e8..4e8 rows are drawn ON THE GPU (torch is
plumbing here: random bits and a scatter), because numpy needs minutes for them; only the rows planted around
variant sites come from malva_amd.synth, with labels, so the test knows what every counter must hold."""
import numpy as np
import torch

from malva_amd import synth


class DeviceTable:
    """SoA k-mer table resident in HBM: d_hi / d_lo (int64 bit patterns of the u64 halves), d_cnt (int32)"""

    def __init__(self, panel, n_rows, k, ref_k, seed, plant_variants=None, device=0):
        t = synth.device_table(panel, n_rows, k, ref_k, seed, device, plant_variants)
        self.d_hi, self.d_lo, self.d_cnt, self.n, self.n_site = t["d_hi"], t["d_lo"], t["d_cnt"], t["n"], t["n_site"]
        self.site_where, self.site_cnt = t["site_where"], t["site_cnt"]
        self.site_var, self.site_hap, self.site_off = t["site_var"], t["site_hap"], t["site_off"]

    def ptrs(self, a=0, b=None):
        b = self.n if b is None else b
        return self.d_hi[a:b].data_ptr(), self.d_lo[a:b].data_ptr(), self.d_cnt[a:b].data_ptr(), b - a

    def host(self, a, b):
        return (self.d_hi[a:b].cpu().numpy().view(np.uint64), self.d_lo[a:b].cpu().numpy().view(np.uint64),
                self.d_cnt[a:b].cpu().numpy().view(np.uint32))

    def expected_sums(self, panel, a=0, b=None):
        """What the planted rows in [a, b) must add: per variant, the summed counts of the centred windows (offset 0)
        that carry the REF allele (-> exact-map value of the REF signature) and the ALT allele (-> counter of the ALT
        signature's filter slot, unless its context is in the reference).  Off-centre windows and the random rows
        do not touch a signature except by a 2^-70 accident."""
        b = self.n if b is None else b
        sel = (self.site_off == 0) & (self.site_where >= a) & (self.site_where < b)
        v, h = self.site_var[sel], self.site_hap[sel]
        allele = panel.donor_gt[v, h.astype(np.int64)]
        ref = np.zeros(panel.n, dtype=np.int64)
        alt = np.zeros(panel.n, dtype=np.int64)
        np.add.at(ref, v[allele == 0], self.site_cnt[sel][allele == 0])
        np.add.at(alt, v[allele == 1], self.site_cnt[sel][allele == 1])
        return ref, alt


def build_device_index(ctx, panel, k, batch=1 << 21):
    """REF signatures -> exact map (in variant order: the counter id of variant j's REF key is j), ALT -> bf"""
    from malva_amd import BF_ALT, BF_CTX
    stride = (k + 1 + 7) // 8 * 8
    for a in range(0, panel.n, batch):
        b = min(panel.n, a + batch)
        sub = synth.Panel(genome=panel.genome, pos=panel.pos[a:b], var_allele_off=panel.var_allele_off[a:b + 1] - panel.var_allele_off[a],
                          allele_off=panel.allele_off[2 * a:2 * b + 1] - panel.allele_off[2 * a], pool=panel.pool[2 * a:2 * b],
                          freq=panel.freq[2 * a:2 * b], present_mask=panel.present_mask[a:b], flags=panel.flags[a:b], donor_gt=panel.donor_gt[a:b])
        sig, _ = synth.snp_signature_rows(sub, k)
        rows = np.zeros((sig.shape[0], stride), dtype=np.uint8)
        rows[:, :k] = sig
        ctx.map_insert(rows[0::2])
        ctx.bf_insert(BF_ALT, rows[1::2])
    ctx.bf_finalize(BF_ALT)
    ctx.ref_scan(panel.genome)
    ctx.bf_finalize(BF_CTX)


def counters_tensor(ctx, device=0):
    """torch view of [bf counters | map counters] (u32 bit patterns as int32) + the split point"""
    from malva_amd.dist import alias_int32
    ptr, n_bf, n_map = ctx.counters_view()
    return alias_int32(ptr, n_bf + n_map, torch.device("cuda", device)), n_bf, n_map
