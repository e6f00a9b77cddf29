"""The records' own copies of their counters (MapSlot::cval / cbf, kmer_dev.h): a call-time lookup reads the counter on the
line of the record it has just walked to, instead of a second random line of vals[] / counts[].  The copies are kept by
the scan and are only trusted while nothing else has changed a counter; every way in and out of that state must leave
the lookups equal to the oracle's (bloom_filter.hpp:100-125, kmap.hpp:114-131, main.cpp:482-500)."""
import numpy as np
import pytest
import torch

from big_cases import DeviceTable, build_device_index
from gpu_util import pad_rows
from malva_amd import BF_ALT, Context, synth
from oracle import capi as ocapi

pytestmark = pytest.mark.gpu
K, R = 35, 43


def _oracle_index(panel, bits):
    obf, octx, omap = ocapi.BF(bits), ocapi.BF(bits), ocapi.KMAP()
    sig, _ = synth.snp_signature_rows(panel, K)
    rows = np.zeros((sig.shape[0], 40), dtype=np.uint8)
    rows[:, :K] = sig
    isr = np.zeros(rows.shape[0], dtype=np.uint8)
    isr[0::2] = 1
    ocapi.add_kmers(obf, omap, rows, isr)
    obf.switch_mode()
    ocapi.ref_scan(obf, octx, panel.genome.tobytes(), K, R)
    octx.switch_mode()
    return obf, octx, omap, rows, isr


def test_copies_follow_every_change_of_a_counter():
    n_vars, n_rows, bits = 30_000, 3_000_000, 1 << 26
    panel = synth.snp_panel(n_vars, 91)
    tab = DeviceTable(panel, n_rows, K, R, 5)
    obf, octx, omap, sig_rows, isr = _oracle_index(panel, bits)
    o = {"bf": obf, "ctx": octx, "map": omap}

    def oracle_reset():  # (the reference has no reset: a new sample is a new process, i.e. the index as built)
        o["bf"], o["ctx"], o["map"] = _oracle_index(panel, bits)[:3]

    with Context(K, R, bits) as ctx:
        ctx.set_option("use_record_counters", 2)  # (1, the default, keeps them only once the counter vectors have outgrown the caches)
        build_device_index(ctx, panel, K)
        ctx.reference_upload(panel.genome)

        def scan(a, b):
            ctx.kmc_scan_device(*tab.ptrs(a, b))
            ctx.synchronize()
            ocapi.kmc_scan_packed(o["ctx"], o["bf"], o["map"], *tab.host(a, b), K, R)

        def check(live):
            assert ctx.get_option("record_counters_live") == live
            cov, g1, g2, gq, st = ctx.call_isolated(panel.pos.astype(np.uint64), panel.var_allele_off, panel.allele_off, panel.pool, panel.freq,
                                                    panel.present_mask, panel.flags, 0.001, 200, False)
            ocov, og1, og2, ogq = ocapi.call_isolated(o["bf"], o["map"], panel.genome, panel.pos, panel.allele_off, panel.var_allele_off, panel.pool,
                                                      panel.freq, panel.present_mask, panel.flags, K, 0.001, 200, False)
            assert np.array_equal(cov, ocov) and np.array_equal(g1, og1) and np.array_equal(g2, og2) and np.array_equal(gq, ogq)
            assert np.array_equal(ctx.bf_export(BF_ALT)[3], o["bf"].counts())  # (the vectors stay the counters of record)
            return cov

        # a freshly built index: the first scan brings the copies up to date by itself, then keeps them
        assert ctx.get_option("record_counters_live") == 0
        scan(0, n_rows // 2)
        first = check(1)
        assert (first > 0).sum() > n_vars // 4
        scan(n_rows // 2, n_rows)  # a second scan on top: the copies accumulate like the vectors
        check(1)
        # per-k-mer increments (ASCII API) go to the vectors only: the lookups fall back to them ...
        rng = np.random.default_rng(3)
        pick = rng.choice(sig_rows.shape[0], size=4000, replace=False)
        big = rng.integers(60000, 65536, size=pick.size).astype(np.uint32)  # (the filter's cells are u16: the next scan wraps them)
        for r, ref, c in zip(sig_rows[pick], isr[pick], big):
            km = bytes(r[:K])
            (o["map"].increment if ref else o["bf"].increment)(km, int(c))
        pr = pad_rows(sig_rows[pick][:, :K])
        ctx.map_increment(pr[isr[pick] == 1], big[isr[pick] == 1].astype(np.int32))
        ctx.bf_increment(BF_ALT, pr[isr[pick] == 0], big[isr[pick] == 0])
        check(0)
        # ... until the next scan has published them into the records; its own additions wrap the u16 cells there as in the vector
        scan(0, n_rows)
        check(1)
        # a reset is a new epoch: every copy reads as zero without having been touched
        ctx.counters_reset()
        oracle_reset()
        assert ctx.get_option("record_counters_live") == 1
        scan(0, n_rows // 3)
        ref_cov = check(1)
        # the same answers without the copies
        ctx.set_option("use_record_counters", 0)
        assert np.array_equal(check(0), ref_cov)
        ctx.set_option("use_record_counters", 2)
        scan(n_rows // 3, n_rows)
        check(1)
        # a caller that holds the vector itself (an all-reduce by other means) may write it at any time: copies off for good
        ctx.counters_view()
        ctx.counters_reset()
        oracle_reset()
        scan(0, n_rows // 2)
        check(0)
    torch.cuda.synchronize()
