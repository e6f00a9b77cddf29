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


def test_lazy_vectors_sub_slice_form_and_the_resident_record_loop():
    """Scans of the sub-slice form add to the records' copies ALONE while those are current (MapView::lazy): vals[] / counts[]
    are brought up to date only when somebody asks for them (rec_collect_kernel).  Every way of asking -- an export, the
    per-k-mer reads, a per-k-mer increment (which ends the copies' validity), a hand-out of the vector -- and the record loop's
    device forms, which read the copies and leave the vectors lazy, must agree with the oracle; a reset forgets what the records
    were ahead by; use_record_counters = 0 and lazy_vectors = 0 give the same answers."""
    from gpu_util import map_values_by_key
    from malva_amd.resident import ResidentPanel
    from test_gpu_resident import oracle_blocks
    k, ref_k, bits = 35, 43, 1 << 30
    panel = synth.clustered_snp_panel(60_000, seed=43, n_contigs=2)
    args = oracle_blocks(panel, k)
    hi, lo, cnt = synth.flat_kmer_table(panel, 600_000, k, ref_k, seed=8, max_records=8_000)
    half = len(hi) // 2

    def oracle_index():
        obf, octx, omap = ocapi.BF(bits), ocapi.BF(bits), ocapi.KMAP()
        ocapi.index_blocks(obf, omap, panel.genome, **args, haploid=False, k=k)
        obf.switch_mode()
        for b, l in zip(panel.contig_base, panel.contig_len):
            ocapi.ref_scan(obf, octx, panel.genome[int(b):int(b) + int(l)].tobytes(), k, ref_k)
        octx.switch_mode()
        return obf, octx, omap
    o = dict(zip(("bf", "ctx", "map"), oracle_index()))
    dev = torch.device("cuda", 0)
    d_hi, d_lo = (torch.from_numpy(a.view(np.int64)).to(dev) for a in (hi, lo))
    d_cnt = torch.from_numpy(cnt.view(np.int32)).to(dev)
    torch.cuda.synchronize()
    with Context(k, ref_k, bits) as ctx:
        for name, value in [("use_record_counters", 2), ("gate_log2", 14), ("use_sub", 1), ("sub_min_log2", 11), ("sub_words_log2", 3)]:
            ctx.set_option(name, value)
        ctx.reference_upload(panel.genome)
        rp = ResidentPanel(panel, 0, haploid=False)
        assert rp.index(ctx).sum() == 0
        ctx.bf_finalize(BF_ALT)
        for b, l in zip(panel.contig_base, panel.contig_len):
            ctx.ref_scan_resident(int(b), int(l))
        ctx.bf_finalize(1)

        def scan(a, b):
            ctx.kmc_scan_device(d_hi[a:b].data_ptr(), d_lo[a:b].data_ptr(), d_cnt[a:b].data_ptr(), b - a)
            ctx.synchronize()
            ocapi.kmc_scan_packed(o["ctx"], o["bf"], o["map"], hi[a:b], lo[a:b], cnt[a:b], k, ref_k)

        def loop_equals_oracle():
            rp.call_step(ctx)
            got = rp.results()
            want = ocapi.cover_blocks(o["bf"], o["map"], panel.genome, **args, haploid=False, k=k)
            assert np.array_equal(got["cov"], want) and got["overflow"].sum() == 0
            g1, g2, gq = ocapi.genotype_panel(want, panel.freq, panel.var_allele_off, 0.001, 200, False)
            assert np.array_equal(got["g1"], g1) and np.array_equal(got["g2"], g2) and np.array_equal(got["gq"], gq)
            return want

        def vectors_equal_oracle():
            assert np.array_equal(ctx.bf_export(BF_ALT)[3], o["bf"].counts())
            assert map_values_by_key(ctx) == dict(o["map"].items())

        scan(0, half)
        assert ctx.get_option("scan_subs") == 32 and ctx.get_option("record_counters_live") == 1 and ctx.get_option("vectors_stale") == 1
        cov = loop_equals_oracle()                              # the record loop reads the copies ...
        assert (cov > 0).sum() > 4_000
        assert ctx.get_option("vectors_stale") == 1             # ... and leaves the vectors lazy
        scan(half, len(hi))                                     # a second lazy scan on top
        assert ctx.get_option("vectors_stale") == 1
        vectors_equal_oracle()                                  # an export collects them
        assert ctx.get_option("vectors_stale") == 0 and ctx.get_option("record_counters_live") == 1
        loop_equals_oracle()
        # a new sample: the reset forgets what the records were ahead by, stale or not
        scan(0, half)
        assert ctx.get_option("vectors_stale") == 1
        ctx.counters_reset()
        o.update(zip(("bf", "ctx", "map"), oracle_index()))
        assert ctx.get_option("vectors_stale") == 0
        vectors_equal_oracle()                                  # all zero
        scan(half, len(hi))
        loop_equals_oracle()
        # a per-k-mer increment goes to the vectors: they are collected first, and the copies are republished by the next scan
        sig, _ = synth.snp_signature_rows(synth.snp_panel(50, 5), k)
        some = pad_rows(sig[:8])
        key = ctx.map_export()[0][0]
        ctx.map_increment(pad_rows(np.frombuffer(key, dtype=np.uint8)[None, :k]), np.array([77], dtype=np.int32))
        o["map"].increment(key, 77)
        ctx.bf_increment(BF_ALT, some, np.arange(8, dtype=np.uint32))  # (random k-mers: no bit of theirs is set, nothing changes)
        assert ctx.get_option("record_counters_live") == 0 and ctx.get_option("vectors_stale") == 0
        vectors_equal_oracle()
        scan(0, half)
        assert ctx.get_option("vectors_stale") == 1
        loop_equals_oracle()
        vectors_equal_oracle()
        # the same answers with the adds going to the vectors too, and without the copies
        for name in ("lazy_vectors", "use_record_counters"):
            ctx.set_option(name, 0)
            ctx.counters_reset()
            o.update(zip(("bf", "ctx", "map"), oracle_index()))
            scan(0, len(hi))
            assert ctx.get_option("vectors_stale") == 0
            loop_equals_oracle()
            vectors_equal_oracle()
        # a caller that takes the vector itself gets it up to date
        ctx.set_option("lazy_vectors", 1)
        ctx.set_option("use_record_counters", 2)
        ctx.counters_reset()
        o.update(zip(("bf", "ctx", "map"), oracle_index()))
        scan(0, half)
        assert ctx.get_option("vectors_stale") == 1
        from malva_amd.dist import alias_int32
        ptr, n_bf, n_map = ctx.counters_view()
        view = alias_int32(ptr, n_bf + n_map, dev)
        torch.cuda.synchronize()
        assert np.array_equal(view[:n_bf].cpu().numpy().view(np.uint32) & 0xFFFF, o["bf"].counts())
    torch.cuda.synchronize()
