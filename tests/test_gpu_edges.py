"""Edge cases and full-size properties of the HIP path: empty and ragged inputs, maximum k-mer lengths,
filters with nothing in them, k-mers at contig ends, and BASELINE-sized runs checked through
size-independent properties (linearity, shard sums, an independent checksum of the expected hits)."""
import numpy as np
import pytest

from gpu_util import build_index_pair, map_values_by_key
from malva_amd import BF_ALT, BF_CTX, Context, MalvaError, synth
from malva_amd.capi import rows_of
from oracle import capi as ocapi

pytestmark = pytest.mark.gpu


def test_empty_inputs_everywhere():
    ctx = Context(35, 43, 1 << 16)
    empty = np.zeros((0, 40), dtype=np.uint8)
    ctx.bf_insert(BF_ALT, empty)
    ctx.map_insert(empty)
    assert ctx.bf_test(BF_ALT, empty).shape == (0,)
    assert ctx.map_size() == 0
    ctx.bf_finalize(BF_ALT)                                   # a filter with no set bit: popcount 0, no counters
    assert ctx.bf_info(BF_ALT) == (1 << 16, 0, 1)
    ctx.ref_scan(b"ACGT" * 100)                               # nothing can hit an empty bf
    ctx.bf_finalize(BF_CTX)
    assert ctx.bf_info(BF_CTX)[1] == 0
    z = np.zeros(0, dtype=np.uint64)
    ctx.kmc_scan(z, z, np.zeros(0, dtype=np.uint32))         # empty table
    hi, lo = synth.pack_ascii(synth.BASES[np.random.default_rng(1).integers(0, 4, size=(1000, 43))])
    ctx.kmc_scan(hi, lo, np.full(1000, 7, dtype=np.uint32))  # rows against an empty index: no effect, no crash
    assert ctx.counters_size() == (0, 0)
    km = [b"ACGTACGTACGTACGTACGTACGTACGTACGTACG"]
    assert ctx.bf_get_count(BF_ALT, rows_of(km))[0] == 0 and ctx.map_get_count(rows_of(km))[0] == 0
    cov = ctx.lookup_cover([], [], [0], [0, 0, 0])            # two allele slots without signatures
    assert list(cov) == [0, 0]
    g1, g2, gq, st, _, _ = ctx.genotype(cov, np.array([0.5, 0.5], np.float32), np.array([0, 2], np.uint32), 0.001, 200, False)
    assert (int(g1[0]), int(g2[0]), int(gq[0])) == (0, 0, 0)
    mode, size, pos, counts = ctx.bf_export_sparse(BF_ALT)
    assert pos.size == 0 and counts.size == 0
    ctx.close()


def test_ragged_rows_and_maximum_lengths():
    """rows of different lengths in one batch (1..128 bytes), longest supported k-mer, k == ref_k == 64 packed"""
    rng = np.random.default_rng(4)
    size = (1 << 20) + 7
    kmers = [bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), size=L)) for L in (1, 2, 3, 4, 8, 9, 16, 17, 32, 33, 64, 65, 127, 128)]
    ctx = Context(35, 43, size)
    obf = ocapi.BF(size)
    for km in kmers:
        obf.add_key(km)
    ctx.bf_insert(BF_ALT, rows_of(kmers, 136))
    obf.switch_mode(); ctx.bf_finalize(BF_ALT)
    assert np.array_equal(ctx.bf_export(BF_ALT)[2], obf.words())
    with pytest.raises(ValueError):
        rows_of([b"A" * 136], 136)                            # does not fit a row
    with pytest.raises(MalvaError):
        Context(200, 201, size)                               # k beyond MG_MAX_KMER
    ctx.close()
    ctx = Context(64, 64, 1 << 22)
    rows = synth.BASES[rng.integers(0, 4, size=(2000, 64))]
    hi, lo = synth.pack_ascii(rows)
    want = np.array([ocapi.lib().mo_bf_hash(bytes(r)) % (1 << 22) for r in rows], dtype=np.uint64)
    assert np.array_equal(ctx.packed_index(BF_ALT, hi, lo, 64), want)
    ctx.close()


def test_sparse_export_import_roundtrip_and_rejects_bad_positions():
    rng = np.random.default_rng(6)
    kms = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=35)) for _ in range(3000)]
    a = Context(35, 43, 1 << 30)
    a.bf_insert(BF_ALT, rows_of(kms)); a.bf_finalize(BF_ALT)
    a.bf_increment(BF_ALT, rows_of(kms), np.arange(3000, dtype=np.uint32) % 500)
    mode, size, pos, counts = a.bf_export_sparse(BF_ALT)
    assert np.all(np.diff(pos.astype(np.int64)) > 0) and pos.size == a.bf_info(BF_ALT)[1]
    b = Context(35, 43, 1 << 30)
    b.bf_import_sparse(BF_ALT, mode, size, pos, counts)
    assert np.array_equal(b.bf_get_count(BF_ALT, rows_of(kms)), a.bf_get_count(BF_ALT, rows_of(kms)))
    assert np.array_equal(b.bf_export(BF_ALT)[2], a.bf_export(BF_ALT)[2])
    with pytest.raises(MalvaError):
        b.bf_import_sparse(BF_CTX, 1, size, pos[::-1].copy(), counts)     # not ascending
    with pytest.raises(MalvaError):
        b.bf_import_sparse(BF_CTX, 1, size, np.array([size], np.uint64), np.zeros(1, np.uint16))  # out of range
    a.close(); b.close()


def test_variants_at_contig_ends_are_skipped_like_the_reference():
    """var_block.hpp:104: ref_pos < k or > len - k gives no signatures -> coverage 0 -> 0/0:0"""
    k = 35
    panel = synth.snp_panel(50, 9, spacing=30, first=20, tail=0)       # first and last variants sit within k of the ends
    glen = panel.genome.size
    panel.flags[:] = ((panel.pos >= k) & (panel.pos <= glen - k)).astype(np.uint8)
    assert panel.flags[0] == 0 and panel.flags[-1] == 0 and panel.flags.sum() > 10
    ctx = Context(k, 43, 1 << 20)
    rows, valid = synth.signature_rows(panel, k)
    obf, octx, omap = build_index_pair(ctx, panel, k, 43, 1 << 20, rows, valid)
    hi, lo, cnt = synth.kmer_table(synth.Panel(genome=panel.genome, pos=panel.pos[1:-1], var_allele_off=panel.var_allele_off[:-2],
                                               allele_off=panel.allele_off[:-4], pool=panel.pool[2:-2], freq=panel.freq[2:-2],
                                               present_mask=panel.present_mask[1:-1], flags=panel.flags[1:-1],
                                               donor_gt=panel.donor_gt[1:-1]), 4000, k, 43, 10)
    ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, 43)
    ctx.kmc_scan(hi, lo, cnt)
    ctx.reference_upload(panel.genome)
    cov, g1, g2, gq, st = ctx.call_isolated(panel.pos.astype(np.uint64), panel.var_allele_off, panel.allele_off, panel.pool, panel.freq,
                                            panel.present_mask, panel.flags, 0.001, 200, False)
    ocov, og1, og2, ogq = ocapi.call_isolated(obf, omap, panel.genome, panel.pos, panel.allele_off, panel.var_allele_off, panel.pool,
                                              panel.freq, panel.present_mask, panel.flags, k, 0.001, 200, False)
    assert np.array_equal(cov, ocov) and np.array_equal(g1, og1) and np.array_equal(g2, og2) and np.array_equal(gq, ogq)
    assert (int(g1[0]), int(g2[0]), int(gq[0])) == (0, 0, 0) and cov[:2].sum() == 0
    ctx.close()


def test_full_size_properties_b4():
    """BASELINE C3 shape at 1/5 scale with the real filter size (b=4: two 4-GiB filters): the oracle cannot run here in
    seconds, so the run is checked through properties that do not depend on size."""
    k, ref_k, bits = 35, 43, 4 << 33
    n_vars, n_rows = 200_000, 20_000_000
    panel = synth.snp_panel(n_vars, 77)
    hi, lo, cnt = synth.kmer_table(panel, n_rows, k, ref_k, 78)
    sig, _ = synth.snp_signature_rows(panel, k)
    rows = np.zeros((sig.shape[0], 40), dtype=np.uint8); rows[:, :k] = sig

    def fresh():
        c = Context(k, ref_k, bits)
        c.map_insert(rows[0::2]); c.bf_insert(BF_ALT, rows[1::2]); c.bf_finalize(BF_ALT)
        c.ref_scan(panel.genome.tobytes()); c.bf_finalize(BF_CTX)
        return c

    def state(c):
        keys, vals = c.map_export()
        order = np.argsort(np.array(keys, dtype=object))
        return c.bf_export_sparse(BF_ALT)[3].astype(np.uint32), vals[order].astype(np.int64)

    whole = fresh(); whole.kmc_scan(hi, lo, cnt); w_bf, w_map = state(whole)
    # 1. an independent checksum: every exact-map value must equal the summed counts of the table rows whose centre
    #    k-mer IS that REF signature -- computed here with numpy on packed integers, no hashing involved
    ch, cl = synth.canonical_m(*synth.pack_ascii(sig[0::2]), k)
    key = {(int(a), int(b)): i for i, (a, b) in enumerate(zip(ch, cl))}
    mh, ml = synth._shr128(hi, lo, 2 * ((ref_k - k) - (ref_k - k) // 2))
    kh, kl = synth._mask(k)
    th, tl = synth.canonical_m(mh & kh, ml & kl, k)
    expect = np.zeros(n_vars, dtype=np.int64)
    sel = np.flatnonzero(np.isin(tl, cl))                      # cheap pre-filter on the low word
    for i in sel:
        j = key.get((int(th[i]), int(tl[i])))
        if j is not None:
            expect[j] += int(cnt[i])
    got = map_values_by_key(whole)
    canon_rows = synth.unpack_ascii(ch, cl, k)[:, :k]
    assert all(got[bytes(r)] == int(e) for r, e in zip(canon_rows, expect))
    assert expect.sum() > 0
    whole.close()
    # 2. linearity: scanning the table twice doubles every counter (u16 / u32 wrap)
    twice = fresh(); twice.kmc_scan(hi, lo, cnt); twice.kmc_scan(hi, lo, cnt); t_bf, t_map = state(twice); twice.close()
    assert np.array_equal((2 * w_bf) & 0xFFFF, t_bf) and np.array_equal(2 * w_map, t_map)
    # 3. shards: two halves on two contexts sum to the whole (what the all-reduce relies on)
    a = fresh(); a.kmc_scan(hi[: n_rows // 2], lo[: n_rows // 2], cnt[: n_rows // 2]); a_bf, a_map = state(a); a.close()
    b = fresh(); b.kmc_scan(hi[n_rows // 2:], lo[n_rows // 2:], cnt[n_rows // 2:]); b_bf, b_map = state(b); b.close()
    assert np.array_equal((a_bf + b_bf) & 0xFFFF, w_bf) and np.array_equal(a_map + b_map, w_map)


def test_calls_from_another_host_thread():
    """the ABI has no thread affinity: every entry point makes the context's device current for the calling thread
    (a thread that never touched HIP sits on device 0) and puts the previous one back.  Build on this thread, scan and
    read back on another, compare with the oracle."""
    import threading
    from gpu_util import build_index_pair, map_values_by_key
    k, ref_k, bits = 35, 43, 1 << 20
    panel = synth.snp_panel(800, 61)
    hi, lo, cnt = synth.kmer_table(panel, 30000, k, ref_k, 62)
    ctx = Context(k, ref_k, bits)
    obf, octx, omap = build_index_pair(ctx, panel, k, ref_k, bits)
    ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, ref_k)
    got = {}

    def work():
        try:
            ctx.kmc_scan(hi, lo, cnt)
            got["counts"] = ctx.bf_export(BF_ALT)[3]
            got["map"] = map_values_by_key(ctx)
        except Exception as e:          # surfaced on the main thread below
            got["error"] = e
    t = threading.Thread(target=work)
    t.start(); t.join()
    assert "error" not in got, got.get("error")
    assert np.array_equal(got["counts"], obf.counts()) and got["map"] == dict(omap.items())
    ctx.close()


def test_index_isolated_and_cut_blocks_refuse_what_they_cannot_do():
    """error behaviour of the round-2 entry points: wrong state, missing reference, flanks outside the reference, empty batches"""
    panel = synth.snp_panel(50, 77)
    args = (panel.pos.astype(np.uint64), panel.var_allele_off, panel.allele_off, panel.pool, panel.present_mask, panel.flags)
    with Context(35, 43, 1 << 20) as ctx:
        with pytest.raises(MalvaError):                                   # no reference uploaded yet
            ctx.index_isolated(*args)
        ctx.reference_upload(panel.genome)
        assert ctx.index_isolated(*(a[:0] if i != 1 else a[:1] for i, a in enumerate(args))).size == 0    # empty batch: no-op
        far = args[0].copy()
        far[3] = len(panel.genome) - 5                                    # flagged eligible, right flank leaves the reference
        with pytest.raises(MalvaError):
            ctx.index_isolated(far, *args[1:])
        assert not ctx.index_isolated(*args).any() and ctx.map_size() == 50
        ctx.bf_finalize(BF_ALT)
        with pytest.raises(MalvaError):                                   # `bf` is in read mode now
            ctx.index_isolated(*args)
        assert len(ctx.cut_blocks([], [], [], [])) == 0
        assert list(ctx.cut_blocks([10, 20, 100], [1, 1, 1], [1, 1, 1], [0, 0, 0])) == [0, 2, 3]     # 10 + 18 >= 20 near; 20 + 18 < 100
        assert list(ctx.cut_blocks([10, 20, 21], [1, 1, 1], [1, 1, 1], [0, 1, 1])) == [0, 1, 3]       # another sequence cuts
