"""General variant blocks enumerated ON THE DEVICE (mg_cover_blocks: chains, haplotype picks, signature assembly,
lookup, coverage) against the oracle's block model + set_coverages on clustered synthetic panels."""
import numpy as np
import pytest

import vcf_synth
from block_util import pack_blocks
from malva_amd import BF_ALT, BF_CTX, Context
from malva_amd.capi import rows_of
from oracle import capi as ocapi
from oracle import pipeline
from oracle.model import VCFReader, flatten_vk, read_fasta

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flat", [1, 0])
@pytest.mark.parametrize("seed,haploid,k,ref_k", [(31, False, 35, 43), (32, True, 35, 43), (33, False, 31, 41), (34, False, 21, 29),
                                                  (35, True, 63, 64), (36, False, 35, 63)])
def test_device_enumeration_matches_oracle(tmp_path, seed, haploid, k, ref_k, flat):
    """flat = 1: the tiers as the library deals the records out (csrc/block_pipeline.h); flat = 0: every general record
    through the workgroup-per-record kernel"""
    _run_case(tmp_path, seed, haploid, k, ref_k, dense=False, flat=flat)


@pytest.mark.parametrize("flat", [1, 0])
@pytest.mark.parametrize("seed,haploid", [(41, False), (42, True), (43, False)])
def test_dense_clusters_hit_the_device_capacities(tmp_path, seed, haploid, flat):
    """6-15 variants inside 22 bp with overlapping deletions: long chains, many chains per side, long unphased runs.
    Whatever the device does not hand back must still equal the oracle, and it must hand back only a minority."""
    _run_case(tmp_path, seed, haploid, 35, 43, dense=True, flat=flat)


@pytest.mark.parametrize("seed,haploid,dense,limit", [(51, False, False, 0), (52, True, False, 1), (53, False, True, 2)])
def test_direct_evaluation_when_the_pick_set_overflows(tmp_path, seed, haploid, dense, limit):
    """the kernel keeps a chain's distinct haplotype picks in an LDS set and evaluates each once; with the set limited
    to 0..2 picks nearly every chain overflows it and every sample's pick is evaluated directly: same coverages"""
    _run_case(tmp_path, seed, haploid, 35, 43, dense=dense, set_limit=limit, flat=0)


@pytest.mark.parametrize("flat", [1, 0])
@pytest.mark.parametrize("sp_default", [1 << 14, 0, 1 | (1 << 14)])
@pytest.mark.parametrize("seed,haploid,dense", [(37, False, False), (38, True, False), (44, False, True)])
def test_sparse_genotype_layout(tmp_path, seed, haploid, dense, flat, sp_default):
    """the panel's genotypes handed over as the entries other than one default word (mg_cover_blocks_sparse): same
    coverages, whether the default is 0|0 phased, 0/0 unphased (an unphased panel: the phase bit of a homozygous
    genotype still decides how the sample's other genotypes along a chain combine) or any other word (1|0 here).
    flat = 0 sends every record through the workgroup kernel, whose sample walk then runs over the ENTRIES of a chain's
    members (and once over the sample that has none)."""
    _run_case(tmp_path, seed, haploid, 35, 43, dense=dense, flat=flat, sparse=True, sp_default=sp_default)


def _run_case(tmp_path, seed, haploid, k, ref_k, dense, set_limit=None, flat=1, sparse=False, sp_default=1 << 14):
    prefix = str(tmp_path / "case")
    contigs, records = vcf_synth.make_case(prefix, seed, haploid=haploid, k=k, n_clusters=40 if dense else 120, vcf_strip_chr=True,
                                           dense=dense, n_samples=4 if dense else 5)
    table = str(tmp_path / "donor.txt")
    vcf_synth.donor_table(contigs, records, ref_k, seed, table)
    opt = pipeline.Options(haploid=haploid, k=k, ref_k=ref_k, bf_size=1 << 24, strip_chr=True)
    fa, vcf = prefix + ".fa", prefix + ".vcf"
    idx = pipeline.index(fa, vcf, opt)
    kmers = [(l.split()[0].encode(), int(l.split()[1])) for l in open(table)]
    pipeline.scan(idx, kmers, opt)
    refs = read_fasta(fa, True)
    # the same index and counters on the device, through the ASCII batch calls
    ctx = Context(k, ref_k, opt.bf_size)
    ctx.set_option("use_flat_tier", flat)
    ctx.set_option("use_packed_pool", seed % 2)   # signature k-mers assembled from the packed allele pool (odd seeds) or from its bytes
    if set_limit is not None:
        ctx.set_option("blocks_set_limit", set_limit)
    bits = idx.bf.set_positions()
    ctx.bf_import_sparse(BF_ALT, 1, opt.bf_size, bits, idx.bf.counts())
    ctx.bf_import_sparse(BF_CTX, 1, opt.bf_size, idx.context_bf.set_positions(), idx.context_bf.counts())
    items = list(idx.ref_bf.items())
    ctx.map_import([k_ for k_, _ in items], np.array([v for _, v in items], dtype=np.int32))
    names = list(refs)
    base, off = {}, 0
    for n in names:
        base[n] = off; off += len(refs[n])
    ctx.reference_upload("".join(refs[n] for n in names).encode())
    blocks, want = [], []
    reader = VCFReader(vcf, "-")
    vbs = []
    last = ""
    for vb, reference, used in pipeline._blocks(reader, opt, refs, False):
        if vb is None:
            break
        # the block's contig: the one whose sequence it was evaluated against
        name = next((n for n in names if refs[n] is reference or refs[n] == reference), names[0])
        blocks.append((vb, name))
        km = vb.extract_kmers(reference, haploid)
        ks, is_ref, so, ao = flatten_vk(km, [len(v.alts) + 1 for v in vb.variants])
        w = ocapi.lookup_weights(idx.bf, idx.ref_bf, ocapi.rows_from_kmers(ks)[0], np.array(is_ref, np.uint8)) if ks else np.zeros(0, np.int32)
        want.append(ocapi.set_coverages(w, so, ao))
    want = np.concatenate(want)
    args = pack_blocks(blocks, base, {n: len(refs[n]) for n in names})
    cov, ovf = ctx.cover_blocks(**args, haploid=haploid, sparse=sparse, sp_default=sp_default)
    ok = np.repeat(ovf == 0, np.diff(np.array(args["var_allele_off"])))
    assert ok.mean() > (0.6 if dense else 0.9), "too many variants fell back: %.2f" % (1 - ok.mean())
    assert np.array_equal(cov[ok], want[ok])
    assert (want[ok] > 0).sum() > 50
    # (with 16 chains per side and 32 members per chain side the dense clusters mostly fit: what still comes back is
    #  unphased runs beyond 2^10 mixes and windows clipped by a contig end)
    # variants flagged overflow must be genuinely beyond a device capacity or clipped by a contig end -- never wrong
    print("fallback variants: %d of %d" % (int(ovf.sum()), len(ovf)))
    ctx.close()


@pytest.mark.parametrize("haploid", [False, True])
def test_positions_beyond_2_to_the_25_float_near(tmp_path, haploid):
    """Chain walks on the device at positions where the reference's `are_near` runs in float (var_block.hpp:417-423: the
    int sum is promoted by ceil((float)k / 2)): 35 Mb contig, clusters beyond 2^25 with gaps around the threshold.  The
    blocks are cut by the oracle (float too); inside them the device must find the oracle's chains -- an integer
    walk finds others (tests/test_host_enumerator_cpu.py shows 781 blocks against 666 on the same case)."""
    k, ref_k, bits = 35, 43, 1 << 24
    prefix = str(tmp_path / "far")
    seq, records, pairs = vcf_synth.make_far_case(prefix, 12, haploid=haploid)
    assert pairs > 30
    opt = pipeline.Options(haploid=haploid, k=k, ref_k=ref_k, bf_size=bits)
    refs = {"1": seq}
    blocks, vks = [], []
    for vb, reference, _ in pipeline._blocks(VCFReader(prefix + ".vcf", "-"), opt, refs, False):
        if vb is None:
            break
        blocks.append((vb, "1"))
        vks.append(vb.extract_kmers(reference, haploid))
    assert sum(len(vb.variants) > 1 for vb, _ in blocks) > 200
    # an index straight from the signatures (no reference scan needed here) with made-up weights
    obf, omap = ocapi.BF(bits), ocapi.KMAP()
    for km in vks:
        for per in km.values():
            for a, sigs in per.items():
                for sig in sigs:
                    for s_ in sig:
                        (omap if a == 0 else obf).add_key(s_.encode())
    obf.switch_mode()
    for km in vks:
        for per in km.values():
            for a, sigs in per.items():
                for sig in sigs:
                    for s_ in sig:
                        w = 1 + ocapi.xxh3_64(s_.encode()) % 97
                        if a == 0:
                            omap.increment(s_.encode(), w)
                        else:
                            obf.increment(s_.encode(), w)
    want = []
    for (vb, _), km in zip(blocks, vks):
        ks, is_ref, so, ao = flatten_vk(km, [len(v.alts) + 1 for v in vb.variants])
        w = ocapi.lookup_weights(obf, omap, ocapi.rows_from_kmers(ks)[0], np.array(is_ref, np.uint8)) if ks else np.zeros(0, np.int32)
        want.append(ocapi.set_coverages(w, so, ao))
    want = np.concatenate(want)
    with Context(k, ref_k, bits) as ctx:
        ctx.bf_import_sparse(BF_ALT, 1, bits, obf.set_positions(), obf.counts())
        ctx.bf_import_sparse(BF_CTX, 1, bits, np.zeros(0, np.uint64), np.zeros(0, np.uint16))
        items = list(omap.items())
        ctx.map_import([k_ for k_, _ in items], np.array([v for _, v in items], dtype=np.int32))
        ctx.reference_upload(seq.encode())
        args = pack_blocks(blocks, {"1": 0}, {"1": len(seq)})
        cov, ovf = ctx.cover_blocks(**args, haploid=haploid)
    ok = np.repeat(ovf == 0, np.diff(np.array(args["var_allele_off"])))
    assert ok.mean() > 0.95
    assert np.array_equal(cov[ok], want[ok])
    assert (want[ok] > 0).sum() > 1000


@pytest.mark.parametrize("k,ref_k,n_bad", [(35, 43, 0), (35, 43, 40), (21, 29, 7), (64, 64, 0), (15, 23, 0)])
def test_lone_variants_indexed_on_the_device(k, ref_k, n_bad):
    """mg_index_isolated: extract_kmers + add_kmers_to_bf (var_block.hpp:95-219 with comb = {v}; main.cpp:122-144) for blocks of
    one short variant, against the oracle's add_key calls over the same signatures: same `bf` bits, same exact-map keys.
    Variants with an N in their window (n_bad of them) and every variant when k < 17 come back flagged and untouched."""
    from malva_amd import synth
    bits = 1 << 24
    panel = synth.snp_panel(4000, 300 + k)
    genome = panel.genome.copy()
    rng = np.random.default_rng(k)
    bad = rng.choice(panel.n, size=n_bad, replace=False) if n_bad else np.zeros(0, dtype=np.int64)
    for v in bad:
        genome[int(panel.pos[v]) - 3] = ord("N")
    sig, valid = synth.snp_signature_rows(synth.Panel(genome=genome, pos=panel.pos, var_allele_off=panel.var_allele_off, allele_off=panel.allele_off,
                                                      pool=panel.pool, freq=panel.freq, present_mask=panel.present_mask, flags=panel.flags,
                                                      donor_gt=panel.donor_gt), k)
    with Context(k, ref_k, bits) as ctx:
        ctx.reference_upload(genome)
        ovf = ctx.index_isolated(panel.pos.astype(np.uint64), panel.var_allele_off, panel.allele_off, panel.pool, panel.present_mask, panel.flags)
        want_ovf = np.zeros(panel.n, dtype=np.uint8)
        want_ovf[bad] = 1
        if k < 17:
            want_ovf[:] = 1
        assert np.array_equal(ovf, want_ovf)
        obf, omap = ocapi.BF(bits), ocapi.KMAP()
        rows = np.zeros((sig.shape[0], 72), dtype=np.uint8)
        rows[:, :k] = sig
        keep = np.repeat(want_ovf == 0, 2)                       # signature rows come in (REF, ALT) pairs per variant
        is_ref = np.zeros(rows.shape[0], dtype=np.uint8)
        is_ref[0::2] = 1
        ocapi.add_kmers(obf, omap, rows[keep], is_ref[keep])
        ctx.bf_finalize(BF_ALT)
        obf.switch_mode()
        assert np.array_equal(ctx.bf_export(BF_ALT)[2], obf.words())
        keys, vals = ctx.map_export()
        assert sorted(keys) == sorted(k_ for k_, _ in omap.items()) and not vals.any()
        assert (want_ovf == 0).sum() == 0 or len(keys) > 0


@pytest.mark.parametrize("k", [35, 21, 64])
def test_blocks_cut_on_the_device(k):
    """mg_cut_blocks: the cut test of the record loops (main.cpp:341, 547 -- not near the block's last record, or another
    sequence) for a batch of records, against the oracle's are_near (float arithmetic) record by record: positions below
    and far above 2^24, gaps around the threshold, contig changes, a batch of one, an empty batch."""
    rng = np.random.default_rng(k)
    n = 200_000
    gaps = rng.integers(1, 2 * k, size=n)
    pos = np.cumsum(gaps).astype(np.int64)
    pos[n // 2:] += (1 << 27) - pos[n // 2]                      # second half starts at 2^27: float spacing 8..16
    pos[n // 4:n // 2] += (1 << 24) - 5000 - pos[n // 4]         # second quarter straddles 2^24
    pos = np.maximum.accumulate(pos)
    ref_size = rng.integers(1, 12, size=n).astype(np.uint32)
    min_size = np.minimum(ref_size, rng.integers(1, 12, size=n)).astype(np.uint32)
    contig = np.sort(rng.integers(0, 5, size=n)).astype(np.uint32)
    want = [0]
    for i in range(1, n):
        if contig[i] != contig[i - 1] or not ocapi.are_near(int(pos[i - 1]), int(ref_size[i - 1]), int(min_size[i - 1]), 0, k, int(pos[i])):
            want.append(i)
    want.append(n)
    with Context(k, k, 1 << 20) as ctx:
        got = ctx.cut_blocks(pos, ref_size, min_size, contig)
        assert np.array_equal(got, np.array(want, dtype=np.uint32))
        assert 0.05 < (len(want) - 1) / n < 0.95
        assert np.array_equal(ctx.cut_blocks(pos[:1], ref_size[:1], min_size[:1], contig[:1]), np.array([0, 1], dtype=np.uint32))
        assert len(ctx.cut_blocks(pos[:0], ref_size[:0], min_size[:0], contig[:0])) == 0
        # 1,025 records: one more than the offsets kernel's tile
        assert np.array_equal(ctx.cut_blocks(pos[:1025], ref_size[:1025], min_size[:1025], contig[:1025]), np.array([w for w in want if w < 1025] + [1025], dtype=np.uint32))


@pytest.mark.parametrize("flat", [1, 0])
@pytest.mark.parametrize("seed,haploid,k,ref_k,dense", [(61, False, 35, 43, False), (62, True, 35, 43, False), (63, False, 31, 41, False),
                                                        (64, False, 35, 63, False), (65, False, 35, 43, True), (66, True, 35, 43, True)])
def test_index_time_enumeration_on_the_device(tmp_path, seed, haploid, k, ref_k, dense, flat):
    """mg_index_blocks: VB::extract_kmers + add_kmers_to_bf (main.cpp:349-350, 122-144) on the device.  The blocks `index`
    keeps (present variants only) go through it; what it hands back (overflow) is enumerated by the oracle's model and
    inserted through the batch calls, exactly as the CLI does with its host enumerator.  The resulting `bf` bits and
    exact-map key set must equal the index the oracle pipeline builds -- and most variants must stay on the device."""
    prefix = str(tmp_path / "case")
    vcf_synth.make_case(prefix, seed, haploid=haploid, k=k, n_clusters=40 if dense else 150, vcf_strip_chr=True, dense=dense,
                        n_samples=4 if dense else 5)
    opt = pipeline.Options(haploid=haploid, k=k, ref_k=ref_k, bf_size=1 << 24, strip_chr=True)
    fa, vcf = prefix + ".fa", prefix + ".vcf"
    idx = pipeline.index(fa, vcf, opt)
    refs = read_fasta(fa, True)
    names = list(refs)
    base, off = {}, 0
    for n in names:
        base[n] = off; off += len(refs[n])
    with Context(k, ref_k, opt.bf_size) as ctx:
        ctx.set_option("use_flat_tier", flat)
        ctx.reference_upload("".join(refs[n] for n in names).encode())
        blocks = []
        for vb, reference, used in pipeline._blocks(VCFReader(vcf, "-"), opt, refs, True):       # for_index: main.cpp:332
            if vb is None:
                break
            name = next((n for n in names if refs[n] is reference or refs[n] == reference), names[0])
            blocks.append((vb, name, reference))
        args = pack_blocks([(vb, name) for vb, name, _ in blocks], base, {n: len(refs[n]) for n in names})
        ovf = ctx.index_blocks(**args, haploid=haploid)
        assert ovf.mean() < (0.7 if dense else 0.1), "too many variants fell back: %.2f" % ovf.mean()
        # the host side of the contract: blocks holding a flagged variant are enumerated by the model and inserted in batch
        ref_rows, alt_rows = [], []
        bo = args["blk_var_off"]
        for b, (vb, name, reference) in enumerate(blocks):
            if not ovf[bo[b]:bo[b + 1]].any():
                continue
            for per in vb.extract_kmers(reference, haploid).values():
                for a, sigs in per.items():
                    for sig in sigs:
                        (ref_rows if a == 0 else alt_rows).extend(km.encode() for km in sig)
        if ref_rows:
            ctx.map_insert(rows_of(ref_rows, 136))
        if alt_rows:
            ctx.bf_insert(BF_ALT, rows_of(alt_rows, 136))
        ctx.bf_finalize(BF_ALT)
        assert np.array_equal(ctx.bf_export_sparse(BF_ALT)[2], idx.bf.set_positions()) and idx.bf.popcount() > 200
        keys, vals = ctx.map_export()
        assert sorted(keys) == sorted(k_ for k_, _ in idx.ref_bf.items()) and len(keys) > 200 and not vals.any()
        assert ctx.map_size() == len(keys) == len(set(keys))
        # and the index works: reference scan + finalize, then the scan's counters equal the oracle's
        for n in dict.fromkeys(name for _, name, _ in blocks):
            ctx.ref_scan(refs[n].encode())
        ctx.bf_finalize(BF_CTX)
        assert np.array_equal(ctx.bf_export_sparse(BF_CTX)[2], idx.context_bf.set_positions())
