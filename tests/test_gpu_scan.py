"""H10/H11: reference-context scan and KMC scan on the device vs the oracle.
Full counter state is compared: every bf counter (rank order) and every exact-map
value, for tiny filters (collisions, context hits) and full-size ones, with the
cache-resident summaries on and off, specialised (35,43) and generic (k, ref_k)."""
import numpy as np
import pytest

from gpu_util import build_index_pair, map_values_by_key
from malva_amd import BF_ALT, BF_CTX, Context, MalvaError, synth
from oracle import capi as ocapi

pytestmark = pytest.mark.gpu


def _scan_case(k, ref_k, bf_bits, n_vars, n_rows, seed, use_summary=1, genome_edit=None, options=(), after=None):
    panel = synth.snp_panel(n_vars, seed)
    if genome_edit:
        genome_edit(panel.genome)
    ctx = Context(k, ref_k, bf_bits)
    ctx.set_option("use_summary", use_summary)
    for name, value in options:
        ctx.set_option(name, value)
    obf, octx, omap = build_index_pair(ctx, panel, k, ref_k, bf_bits)
    # index parity first: same bits in both filters
    _, _, words, _ = ctx.bf_export(BF_ALT)
    assert np.array_equal(words, obf.words())
    _, _, cwords, _ = ctx.bf_export(BF_CTX)
    assert np.array_equal(cwords, octx.words())
    hi, lo, cnt = synth.kmer_table(panel, n_rows, k, ref_k, seed + 3)
    ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, ref_k)
    ctx.kmc_scan(hi, lo, cnt)
    _, _, _, counts = ctx.bf_export(BF_ALT)
    assert np.array_equal(counts, obf.counts())
    assert map_values_by_key(ctx) == dict(omap.items())
    _, _, _, n_open, n_hits = ctx.scan_stats()
    if after:
        after(ctx)
    ctx.close()
    return obf, octx, n_hits


def test_scan_default_k35_r43_full_size_filter():
    obf, octx, n_hits = _scan_case(35, 43, 1 << 33, 3000, 120000, 11)
    assert obf.counts().any() and n_hits > 0


def test_scan_tiny_filter_forces_collisions_and_context_hits():
    # 2^17-bit filters: ~5% of random k-mers hit bf by collision, the reference scan fills
    # context_bf, so the context test decides many increments
    obf, octx, n_hits = _scan_case(35, 43, 1 << 17, 3000, 150000, 12)
    assert octx.popcount() > 100 and n_hits > 5000


def test_scan_without_summaries_is_identical():
    _scan_case(35, 43, 1 << 17, 2000, 80000, 13, use_summary=0)


@pytest.mark.parametrize("k,ref_k,bits,n_bins,bin_cap", [(35, 43, 1 << 33, 32, 0), (35, 43, 1 << 17, 4, 0), (31, 41, (1 << 18) + 77, 9, 0),
                                                         (35, 43, 1 << 33, 32, 4), (35, 43, 1 << 17, 4, 1)])
def test_scan_partitioned_second_level(k, ref_k, bits, n_bins, bin_cap):
    """the partition form (option use_partition; off by default since the ticket and sub-slice forms took the index sizes it was built
    for: on what is left to it the plain filter kernel is faster): the coarse gate's survivors binned by fine-gate slice; forced here with a 2^10-bit
    coarse gate in front of a fine one of up to 2^14 bits (slices of 2^9 bits).  bin_cap > 0 shrinks the bin segments so that most
    rows overflow them and take the spill list."""
    def check(ctx):
        assert ctx.get_option("pregate_k") > 0 and ctx.get_option("scan_bins") == n_bins
        assert (ctx.get_option("scan_spilled") > 0) == (bin_cap > 0)
    _scan_case(k, ref_k, bits, 3000, 150000, 31, after=check,
               options=[("use_pregate", 2), ("pregate_log2", 10), ("gate_log2", 14), ("scan_bin_cap", bin_cap), ("use_partition", 1)])
    def direct(ctx):
        assert ctx.get_option("scan_bins") == 0
    _scan_case(k, ref_k, bits, 3000, 150000, 31, after=direct,
               options=[("use_pregate", 2), ("pregate_log2", 10), ("gate_log2", 14), ("use_partition", 0)])


@pytest.mark.parametrize("k,ref_k,bits,gate_log2,slices,bin_cap", [(35, 43, 1 << 33, 14, 32, 0), (35, 43, 1 << 17, 14, 32, 0), (35, 63, 1 << 20, 13, 16, 0),
                                                                  (31, 41, (1 << 18) + 77, 14, 32, 0), (35, 43, 1 << 33, 12, 8, 0),
                                                                  (35, 43, 1 << 33, 11, 4, 0),      # fewer slices than XCDs: slices shared
                                                                  (35, 43, 1 << 33, 17, 256, 0),    # the most slices the form takes (a 512 MiB gate at full size)
                                                                  (35, 43, 1 << 33, 14, 32, 16), (35, 43, 1 << 17, 13, 16, 16)])
def test_scan_ticket_form(k, ref_k, bits, gate_log2, slices, bin_cap):
    """whole-genome-sized indexes file an 8-byte ticket per table row under the slice of the fine gate it will probe and
    then walk the slices out of L2 (scan_ticket_kernel / scan_ticket_gate_kernel); forced here on small gates: slices of
    2^9 bits (pregate_log2 = 10), a fine gate of 2^gate_log2 bits.  bin_cap > 0 shrinks the segments so that most
    tickets take the spill list.  Counters must equal the oracle's, as in every other form."""
    shift = 6                                    # the gate holds one bit per 2^shift filter bits, at most 2^gate_log2 of them
    while ((bits + (1 << shift) - 1) >> shift) > (1 << gate_log2):
        shift += 1
    words = (((bits + (1 << shift) - 1) >> shift) + 63) // 64
    slices = (words + 7) // 8 if slices else 0   # slices of 8 words (pregate_log2 = 10)

    def check(ctx):
        assert ctx.get_option("scan_tickets") == slices >= 2 and ctx.get_option("scan_bins") == 0
        assert (ctx.get_option("scan_spilled") > 0) == (bin_cap > 0)
    _scan_case(k, ref_k, bits, 3000, 150000, 31, after=check,
               options=[("pregate_log2", 10), ("gate_log2", gate_log2), ("use_tickets", 1), ("ticket_min_log2", 11), ("scan_bin_cap", bin_cap)])


@pytest.mark.parametrize("k,ref_k,bits,gate_log2,words_log2,bin_cap", [(35, 43, 1 << 33, 14, 3, 0), (35, 43, 1 << 17, 14, 3, 0), (35, 63, 1 << 20, 13, 2, 0),
                                                                       (31, 41, (1 << 18) + 77, 14, 3, 0),   # the last sub-slice is a partial one
                                                                       (35, 43, 1 << 33, 16, 0, 0),          # 1,024 bins: the most the form takes
                                                                       (35, 43, 1 << 33, 12, 5, 0),          # 2 bins shared by several workgroups each
                                                                       (35, 43, 1 << 33, 14, 3, 16), (35, 43, 1 << 17, 13, 2, 16)])
def test_scan_sub_slice_form(k, ref_k, bits, gate_log2, words_log2, bin_cap):
    """whole-genome-sized indexes file an 8-byte ticket per table row under an LDS-sized sub-slice of the fine gate
    (scan_sub_sort_kernel), then one workgroup per sub-slice copies it into LDS and answers that bin's tickets from there
    (scan_sub_gate_kernel); the probe kernel walks the per-bin regions of surviving row numbers.  Forced here on small gates:
    sub-slices of 2^words_log2 words.  bin_cap > 0 shrinks the segments so that most tickets take the spill list (answered from
    the gate in global memory).  Counters must equal the oracle's, as in every other form."""
    shift = 6
    while ((bits + (1 << shift) - 1) >> shift) > (1 << gate_log2):
        shift += 1
    words = (((bits + (1 << shift) - 1) >> shift) + 63) // 64
    n_bins = (words + (1 << words_log2) - 1) >> words_log2

    def check(ctx):
        assert ctx.get_option("scan_subs") == n_bins >= 2 and ctx.get_option("scan_bins") == 0 and ctx.get_option("scan_tickets") == 0
        assert (ctx.get_option("scan_spilled") > 0) == (bin_cap > 0)
    _scan_case(k, ref_k, bits, 3000, 150000, 31, after=check,
               options=[("gate_log2", gate_log2), ("use_sub", 1), ("sub_min_log2", 11), ("sub_words_log2", words_log2), ("scan_bin_cap", bin_cap)])


@pytest.mark.parametrize("form", ["direct", "tickets", "partition"])
@pytest.mark.parametrize("hit_entries", [0, 1])
def test_scan_hit_list_with_and_without_the_filter_entries(form, hit_entries):
    """The probe kernel hands the hit kernel each row's filter entry (counter index, record) with the row (use_hit_entries = 1,
    the default); without it -- what a record table beyond 2^30 records falls back to -- the hit kernel hashes the centre k-mer
    again and walks to the entry itself.  Both, with the records' counter copies kept (use_record_counters = 2), in the direct,
    ticket and partition forms: counters equal the oracle's."""
    options = [("use_hit_entries", hit_entries), ("use_record_counters", 2)]
    if form == "tickets":
        options += [("pregate_log2", 10), ("gate_log2", 14), ("use_tickets", 1), ("ticket_min_log2", 11)]
    if form == "partition":
        options += [("use_pregate", 2), ("pregate_log2", 10), ("gate_log2", 14), ("use_partition", 1)]

    def check(ctx):
        assert ctx.get_option("use_hit_entries") == hit_entries
        assert (ctx.get_option("scan_tickets") > 0) == (form == "tickets") and (ctx.get_option("scan_bins") > 0) == (form == "partition")
    _scan_case(35, 43, 1 << 20, 3000, 150000, 37, after=check, options=options)


@pytest.mark.parametrize("k,ref_k", [(31, 41), (35, 63), (21, 22), (33, 64), (17, 17), (11, 19), (16, 24), (9, 9)])   # below 17 the hash takes XXH3's short-input branches
def test_scan_generic_k(k, ref_k):
    _scan_case(k, ref_k, (1 << 18) + 77, 1500, 60000, 100 + k)


def test_ref_scan_with_non_acgt_bases_and_odd_modulus():
    def edit(g):
        rng = np.random.default_rng(3)
        g[:50] = ord("N")
        for p in rng.integers(100, len(g) - 100, size=300):
            g[p] = rng.choice(np.frombuffer(b"NWMRYK", dtype=np.uint8))
    _scan_case(35, 43, 1000003, 2000, 50000, 14, genome_edit=edit)


def test_scan_is_linear_and_shards_sum():
    """size-independent properties used at full size by bench.py: scanning a table twice doubles
    every counter (mod 2^16 / 2^32); scanning two halves on two contexts and summing equals the whole."""
    k, ref_k, bits = 35, 43, 1 << 22
    panel = synth.snp_panel(2000, 21)
    hi, lo, cnt = synth.kmer_table(panel, 100000, k, ref_k, 22)
    res = []
    for parts in ([slice(0, None)], [slice(0, None), slice(0, None)], [slice(0, 50000)], [slice(50000, None)]):
        ctx = Context(k, ref_k, bits)
        build_index_pair(ctx, panel, k, ref_k, bits)
        for s in parts:
            ctx.kmc_scan(hi[s], lo[s], cnt[s])
        _, _, _, counts = ctx.bf_export(BF_ALT)
        keys, vals = ctx.map_export()
        order = np.argsort(np.array(keys, dtype=object))
        res.append((counts.astype(np.uint32), vals[order].astype(np.int64)))
        ctx.close()
    whole, twice, a, b = res
    assert np.array_equal((2 * whole[0]) & 0xFFFF, twice[0]) and np.array_equal(2 * whole[1], twice[1])
    assert np.array_equal((a[0] + b[0]) & 0xFFFF, whole[0]) and np.array_equal(a[1] + b[1], whole[1])


def test_counters_view_aliases_and_export_import_roundtrip():
    """the exchange step's two forms: export/import copies, and the zero-copy contiguous view"""
    import torch
    from malva_amd.dist import alias_int32
    k, ref_k, bits = 35, 43, 1 << 22
    panel = synth.snp_panel(1500, 41)
    hi, lo, cnt = synth.kmer_table(panel, 60000, k, ref_k, 42)
    ctx = Context(k, ref_k, bits)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)   # one stream for torch's ops and the context's: ordered
    build_index_pair(ctx, panel, k, ref_k, bits)
    ctx.kmc_scan(hi, lo, cnt)
    n_bf, n_map = ctx.counters_size()
    buf = torch.zeros(n_bf + n_map, dtype=torch.int32, device="cuda")
    ctx.counters_export_device(buf.data_ptr()); ctx.synchronize()
    before = map_values_by_key(ctx)
    ptr, a, b = ctx.counters_view()
    assert (a, b) == (n_bf, n_map)
    view = alias_int32(ptr, a + b, torch.device("cuda", 0))
    assert torch.equal(view, buf)                       # same content, now contiguous
    view *= 2                                           # what an in-place all-reduce of two equal shards would leave
    torch.cuda.synchronize()
    after = map_values_by_key(ctx)
    assert all(after[k_] == 2 * v for k_, v in before.items())
    ctx.counters_import_device(buf.data_ptr()); ctx.synchronize()
    assert map_values_by_key(ctx) == before
    ctx.kmc_scan(hi, lo, cnt)                           # the joined arrays keep working as scan targets
    assert all(v2 == 2 * v for (_, v), (_, v2) in zip(sorted(before.items()), sorted(map_values_by_key(ctx).items())))
    ctx.map_insert(rows_of_random(k))                   # growing the map un-joins transparently
    assert ctx.map_size() == len(before) + 5
    ctx.close()


def rows_of_random(k):
    rng = np.random.default_rng(99)
    from malva_amd.capi import rows_of
    return rows_of([bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=k)) for _ in range(5)])


@pytest.mark.parametrize("pregate_log2,expect_k", [(10, 0), (20, 4)])
def test_coarse_gate_is_chosen_or_skipped_at_finalize(pregate_log2, expect_k):
    """gate sized automatically (2^25 bits for this small index, i.e. larger than the coarse gate given here): a
    coarse gate of 2^10 bits would be saturated by 6,000 entries and is skipped; one of 2^20 bits is used with 4 bits
    per entry (direct two-level form: 64 slices are more than the partitioned form takes).  Same counters either way."""
    def check(ctx):
        assert ctx.get_option("pregate_k") == expect_k and ctx.get_option("scan_bins") == 0
    _scan_case(35, 43, 1 << 33, 3000, 100000, 71, after=check, options=[("pregate_log2", pregate_log2)])


@pytest.mark.parametrize("k,ref_k,n_rows,bits", [(35, 43, 120001, 1 << 33), (35, 43, 150003, 1 << 17), (31, 41, 60002, (1 << 18) + 77), (33, 44, 50000, 1 << 20),
                                                 (17, 33, 50001, 1 << 20)])
def test_scan_compact_rows_equals_oracle(k, ref_k, n_rows, bits):
    """12-byte rows (count << 2 ref_k | k-mer): packed on the device from the SoA table, scanned by scan_filter12_kernel;
    every counter equals the oracle's scan of the SoA rows"""
    import torch
    panel = synth.snp_panel(3000, 70 + k)
    hi, lo, cnt = synth.kmer_table(panel, n_rows, k, ref_k, 71)
    cnt[:] = 1 + (cnt * 37) % ((1 << (96 - 2 * ref_k)) - 1)          # the whole range a packed row can hold (ref_k 44: 1..254)
    with Context(k, ref_k, bits) as ctx:
        obf, octx, omap = build_index_pair(ctx, panel, k, ref_k, bits)
        ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, ref_k)
        dev = torch.device("cuda", 0)
        d_hi, d_lo = (torch.from_numpy(a.view(np.int64)).to(dev) for a in (hi, lo))
        d_cnt = torch.from_numpy(cnt.view(np.int32)).to(dev)
        d_rows = torch.zeros(ctx.kmc_rows_bytes(n_rows) // 4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.kmc_pack_rows_device(d_hi.data_ptr(), d_lo.data_ptr(), d_cnt.data_ptr(), n_rows, d_rows.data_ptr())
        w = d_rows.cpu().numpy().view(np.uint32).reshape(-1, 3)[:n_rows].astype(np.uint64)
        kb = 2 * ref_k - 64
        assert np.array_equal(w[:, 0] | (w[:, 1] << np.uint64(32)), lo) and np.array_equal(w[:, 2] & np.uint64((1 << kb) - 1), hi)
        assert np.array_equal(w[:, 2] >> np.uint64(kb), cnt.astype(np.uint64))
        ctx.kmc_scan_rows_device(d_rows.data_ptr(), n_rows)
        ctx.synchronize()
        assert np.array_equal(ctx.bf_export(BF_ALT)[3], obf.counts())
        assert map_values_by_key(ctx) == dict(omap.items())
        # a count that does not fit is refused, not truncated
        d_cnt[7] = 1 << (96 - 2 * ref_k)
        torch.cuda.synchronize()
        with pytest.raises(MalvaError):
            ctx.kmc_pack_rows_device(d_hi.data_ptr(), d_lo.data_ptr(), d_cnt.data_ptr(), n_rows, d_rows.data_ptr())
    with Context(35, 63, 1 << 20) as ctx:                              # 126-bit k-mers leave no room for a count
        with pytest.raises(MalvaError):
            ctx.kmc_scan_rows_device(1 << 20, 4)


@pytest.mark.parametrize("k,ref_k,n_rows,bits,gate_log2,bin_cap", [(35, 43, 150003, 1 << 33, 14, 0), (35, 43, 120001, 1 << 17, 11, 0), (31, 41, 60002, (1 << 18) + 77, 12, 0),
                                                                  (35, 43, 150003, 1 << 33, 14, 16), (33, 44, 4, 1 << 20, 12, 0), (35, 43, 2049, 1 << 20, 12, 0)])
def test_scan_compact_rows_ticket_form(k, ref_k, n_rows, bits, gate_log2, bin_cap):
    """the ticket form over 12-byte rows (what a whole-genome index takes when the table is resident in its compact form):
    pass one reads the packed rows, the open list holds row numbers, the probe kernel fetches the 12 bytes of each.
    Forced on small gates as in test_scan_ticket_form; counters equal the oracle's."""
    import torch
    panel = synth.snp_panel(3000, 90 + k)
    hi, lo, cnt = synth.kmer_table(panel, n_rows, k, ref_k, 91)
    cnt[:] = 1 + (cnt * 37) % ((1 << (96 - 2 * ref_k)) - 1)
    with Context(k, ref_k, bits) as ctx:
        for name, value in [("pregate_log2", 10), ("gate_log2", gate_log2), ("use_tickets", 1), ("ticket_min_log2", 11), ("scan_bin_cap", bin_cap)]:
            ctx.set_option(name, value)
        obf, octx, omap = build_index_pair(ctx, panel, k, ref_k, bits)
        ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, ref_k)
        dev = torch.device("cuda", 0)
        d_hi, d_lo = (torch.from_numpy(a.view(np.int64)).to(dev) for a in (hi, lo))
        d_cnt = torch.from_numpy(cnt.view(np.int32)).to(dev)
        d_rows = torch.zeros(ctx.kmc_rows_bytes(n_rows) // 4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.kmc_pack_rows_device(d_hi.data_ptr(), d_lo.data_ptr(), d_cnt.data_ptr(), n_rows, d_rows.data_ptr())
        ctx.kmc_scan_rows_device(d_rows.data_ptr(), n_rows)
        ctx.synchronize()
        assert ctx.get_option("scan_tickets") >= 2
        assert (ctx.get_option("scan_spilled") > 0) == (bin_cap > 0)
        assert np.array_equal(ctx.bf_export(BF_ALT)[3], obf.counts())
        assert map_values_by_key(ctx) == dict(omap.items())


@pytest.mark.parametrize("k,ref_k,n_rows,bits,gate_log2,words_log2,bin_cap", [(35, 43, 150003, 1 << 33, 14, 3, 0), (35, 43, 120001, 1 << 17, 11, 1, 0),
                                                                             (31, 41, 60002, (1 << 18) + 77, 12, 2, 0), (35, 43, 150003, 1 << 33, 16, 0, 16),
                                                                             (33, 44, 4, 1 << 20, 12, 2, 0), (35, 43, 16385, 1 << 20, 12, 2, 0)])
def test_scan_compact_rows_sub_slice_form(k, ref_k, n_rows, bits, gate_log2, words_log2, bin_cap):
    """the sub-slice form over 12-byte rows (what bench.py's whole-genome workload runs): pass one reads the packed rows, the
    regions hold row numbers, the probe kernel fetches the 12 bytes of each.  Forced on small gates; counters equal the oracle's."""
    import torch
    panel = synth.snp_panel(3000, 90 + k)
    hi, lo, cnt = synth.kmer_table(panel, n_rows, k, ref_k, 91)
    cnt[:] = 1 + (cnt * 37) % ((1 << (96 - 2 * ref_k)) - 1)
    with Context(k, ref_k, bits) as ctx:
        for name, value in [("gate_log2", gate_log2), ("use_sub", 1), ("sub_min_log2", 11), ("sub_words_log2", words_log2), ("scan_bin_cap", bin_cap)]:
            ctx.set_option(name, value)
        obf, octx, omap = build_index_pair(ctx, panel, k, ref_k, bits)
        ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, ref_k)
        dev = torch.device("cuda", 0)
        d_hi, d_lo = (torch.from_numpy(a.view(np.int64)).to(dev) for a in (hi, lo))
        d_cnt = torch.from_numpy(cnt.view(np.int32)).to(dev)
        d_rows = torch.zeros(ctx.kmc_rows_bytes(n_rows) // 4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        ctx.kmc_pack_rows_device(d_hi.data_ptr(), d_lo.data_ptr(), d_cnt.data_ptr(), n_rows, d_rows.data_ptr())
        ctx.kmc_scan_rows_device(d_rows.data_ptr(), n_rows)
        ctx.synchronize()
        assert ctx.get_option("scan_subs") >= 2
        assert (ctx.get_option("scan_spilled") > 0) == (bin_cap > 0)
        assert np.array_equal(ctx.bf_export(BF_ALT)[3], obf.counts())
        assert map_values_by_key(ctx) == dict(omap.items())
        f, p_, h, n_open, n_hits = ctx.scan_stats()
        assert n_open > 0 and f >= 0


@pytest.mark.parametrize("form,chunk_log2", [("direct", 14), ("tickets", 13), ("tickets", 15), ("compact", 12), ("compact-tickets", 14), ("partition", 15),
                                             ("subs", 15), ("compact-subs", 14), ("compact-subs", 16)])
def test_scan_in_many_chunks(form, chunk_log2):
    """A table longer than one launch group (2^27 rows) is scanned chunk by chunk, lists and list counters reused.  Launch
    groups of 2^12..2^15 rows put 5 to 37 chunks into a 150,003-row table (the last one partial): every counter equals the
    oracle's for every form of the scan.  (Running the probe and hit kernels of chunk i on a second stream beside the passes
    of chunk i + 1 was built on top of this and measured at the C4 share: 7.43 ms against 7.24 -- the tail kernels slow down
    by what they would have taken alone; dropped.)"""
    import torch
    k, ref_k, bits, n_rows = 35, 43, 1 << 33, 150003
    options = [("scan_chunk_log2", chunk_log2)]
    if "tickets" in form:
        options += [("pregate_log2", 10), ("gate_log2", 14), ("use_tickets", 1), ("ticket_min_log2", 11)]
    if form == "partition":
        options += [("pregate_log2", 10), ("gate_log2", 14), ("use_tickets", 0), ("use_partition", 1)]
    if "subs" in form:
        options += [("gate_log2", 14), ("use_sub", 1), ("sub_min_log2", 11), ("sub_words_log2", 3)]
    panel = synth.snp_panel(3000, 201)
    hi, lo, cnt = synth.kmer_table(panel, n_rows, k, ref_k, 202)
    with Context(k, ref_k, bits) as ctx:
        for name, value in options:
            ctx.set_option(name, value)
        obf, octx, omap = build_index_pair(ctx, panel, k, ref_k, bits)
        ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, ref_k)
        dev = torch.device("cuda", 0)
        d_hi, d_lo = (torch.from_numpy(a.view(np.int64)).to(dev) for a in (hi, lo))
        d_cnt = torch.from_numpy(cnt.view(np.int32)).to(dev)
        torch.cuda.synchronize()
        if form.startswith("compact"):
            d_rows = torch.zeros(ctx.kmc_rows_bytes(n_rows) // 4, dtype=torch.int32, device=dev)
            torch.cuda.synchronize()
            ctx.kmc_pack_rows_device(d_hi.data_ptr(), d_lo.data_ptr(), d_cnt.data_ptr(), n_rows, d_rows.data_ptr())
            ctx.kmc_scan_rows_device(d_rows.data_ptr(), n_rows)
        else:
            ctx.kmc_scan_device(d_hi.data_ptr(), d_lo.data_ptr(), d_cnt.data_ptr(), n_rows)
        ctx.synchronize()
        assert np.array_equal(ctx.bf_export(BF_ALT)[3], obf.counts())
        assert map_values_by_key(ctx) == dict(omap.items())
        _, _, _, n_open, n_hits = ctx.scan_stats()
        assert n_hits > 0
        if "tickets" in form:
            assert ctx.get_option("scan_tickets") >= 2
        if form == "partition":
            assert ctx.get_option("scan_bins") >= 2
        if "subs" in form:
            assert ctx.get_option("scan_subs") == 32


@pytest.mark.parametrize("k,ref_k", [(35, 43), (35, 63), (31, 42), (21, 29), (64, 64), (17, 18), (33, 64)])
def test_ref_scan_dense_hits_packed_resident_and_bytewise(k, ref_k):
    """H11 (main.cpp:383-401) with MANY hits, so that the context insert is exercised as much as the centre test: `bf` holds
    every 7th k-mer of the reference itself.  Three ways -- the host contig (packed on the fly), the contig inside the
    resident reference at a non-zero, unaligned offset (mg_ref_scan_resident), and the byte-wise kernel for every window
    (use_packed_ref_scan = 0) -- against the oracle: N runs, IUPAC codes, odd ref_k - k (the reference's sliding quirk in the
    first k windows), a contig end."""
    bits = (1 << 22) + 5
    rng = np.random.default_rng(k * 100 + ref_k)
    g = synth.random_genome(260_000, 99 + k).copy()
    g[1000:1400] = ord("N")
    for p in rng.integers(2000, len(g) - 2000, size=400):
        g[p] = rng.choice(np.frombuffer(b"NWMRYK", dtype=np.uint8))
    g[-5:] = ord("N")
    text = g.tobytes()
    rows = np.zeros((len(range(0, len(g) - k, 7)), 72), dtype=np.uint8)
    for i, p in enumerate(range(0, len(g) - k, 7)):
        rows[i, :k] = g[p:p + k]
    obf, octx = ocapi.BF(bits), ocapi.BF(bits)
    ocapi.add_kmers(obf, ocapi.KMAP(), rows, np.zeros(rows.shape[0], dtype=np.uint8))
    obf.switch_mode()
    ocapi.ref_scan(obf, octx, text, k, ref_k)
    want = octx.set_positions()
    assert len(want) > 20_000
    prefix = synth.random_genome(12_345, 5).tobytes()          # another sequence in front: the contig starts at an odd offset
    for mode in ("host", "resident", "bytewise"):
        with Context(k, ref_k, bits) as ctx:
            if mode == "bytewise":
                ctx.set_option("use_packed_ref_scan", 0)
            ctx.bf_insert(BF_ALT, rows)
            ctx.bf_finalize(BF_ALT)
            if mode == "resident":
                ctx.reference_upload(prefix + text)
                ctx.ref_scan_resident(len(prefix), len(text))
            else:
                ctx.ref_scan(text)
            ctx.bf_finalize(BF_CTX)
            got = ctx.bf_export_sparse(BF_CTX)[2]
            assert np.array_equal(got, want), mode


@pytest.mark.parametrize("use_set", [0, 2])
def test_context_filter_blocks_rows_with_and_without_the_position_set(use_set):
    """main.cpp:496-498: a row whose ref_k-mer is in context_bf does NOT increment `bf`.  The table here is made of exactly
    such rows -- every reference window whose context the index put into context_bf (its centre k-mer hits `bf`: with a
    small filter, mostly Bloom false positives) -- beside donor windows and random rows, so the hit kernel's context test
    decides hundreds of rows.  use_ctx_set = 2 answers it from the hash set of context_bf's set positions (what a
    whole-genome filter does by itself), 0 from the bit array: counters equal the oracle's either way."""
    k, ref_k, bits = 35, 43, (1 << 20) + 7
    panel = synth.snp_panel(3000, 5)
    with Context(k, ref_k, bits) as ctx:
        ctx.set_option("use_ctx_set", use_set)
        obf, octx, omap = build_index_pair(ctx, panel, k, ref_k, bits)
        g = panel.genome
        blocked = [p for p in range(0, len(g) - ref_k + 1) if octx.test_key(g[p:p + ref_k].tobytes())]
        assert len(blocked) > 300
        whi, wlo = synth.pack_ascii(synth.windows(g, np.array(blocked), ref_k))
        hi, lo, cnt = synth.kmer_table(panel, 60000, k, ref_k, seed=9)
        hi = np.concatenate([hi, whi]); lo = np.concatenate([lo, wlo]); cnt = np.concatenate([cnt, np.full(len(blocked), 7, dtype=np.uint32)])
        ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, ref_k)
        ctx.kmc_scan(hi, lo, cnt)
        assert (ctx.get_option("ctx_set_log2") > 0) == (use_set == 2)
        assert np.array_equal(ctx.bf_export(BF_ALT)[3], obf.counts())
        assert map_values_by_key(ctx) == dict(omap.items())
        # the blocked rows alone, on fresh counters: nothing may reach `bf` through them
        ctx.counters_reset()
        ctx.kmc_scan(whi, wlo, np.full(len(blocked), 7, dtype=np.uint32))
        assert not ctx.bf_export(BF_ALT)[3].any()
