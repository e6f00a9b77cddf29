"""The synthetic-input generator (packing, canonical form, table shape) on CPU."""
import numpy as np

from malva_amd import synth
from oracle import capi as ocapi
from oracle.kmc_standin import canonical_acgt


def test_pack_roundtrip_and_canonical():
    rng = np.random.default_rng(0)
    for n in (17, 31, 32, 33, 35, 43, 63, 64):
        rows = synth.BASES[rng.integers(0, 4, size=(300, n))]
        hi, lo = synth.pack_ascii(rows)
        assert np.array_equal(synth.unpack_ascii(hi, lo, n)[:, :n], rows)
        ch, cl = synth.canonical_m(hi, lo, n)
        can = synth.unpack_ascii(ch, cl, n)[:, :n]
        for r, c in zip(rows, can):
            assert canonical_acgt(bytes(r)) == bytes(c)
            assert ocapi.canonical(bytes(r)) == bytes(c)        # the oracle's canonical agrees on pure ACGT


def test_signature_rows_match_the_general_enumerator():
    """synth.signature_rows (used to build indexes in tests/bench) == VB.extract_kmers on lone variants"""
    from oracle.model import VB, Variant
    k = 35
    panel = synth.mixed_panel(60, 3, k=k)
    rows, valid = synth.signature_rows(panel, k)
    genome = panel.genome.tobytes().decode()
    for v in range(panel.n):
        A = panel.n_alleles(v)
        var = Variant(seq_name="1", ref_pos=int(panel.pos[v]), ref_sub=bytes(panel.allele(v, 0)).decode(),
                      alts=[bytes(panel.allele(v, a)).decode() for a in range(1, A)])
        var.ref_size = len(var.ref_sub)
        var.set_sizes()
        var.is_present = bool(panel.flags[v] & 1)
        present = [a for a in range(A) if (int(panel.present_mask[v]) >> a) & 1]
        var.genotypes = [(a, a) for a in present]       # one haploid panel sample per present allele
        var.phasing = [True] * len(present)
        vb = VB(k, 0.001)
        vb.add_variant(var)
        got = vb.extract_kmers(genome, True)[0]
        a0 = int(panel.var_allele_off[v])
        want = {a: [[bytes(rows[a0 + a]).decode()]] for a in range(A) if valid[a0 + a]}
        # duplicate allele strings collapse onto the first index in the reference (get_allele_index)
        assert got == want or not var.is_present, (v, got, want)


def test_kmer_table_shape_and_counts():
    panel = synth.snp_panel(500, 5)
    hi, lo, cnt = synth.kmer_table(panel, 20000, 35, 43, 6)
    assert hi.shape == lo.shape == cnt.shape == (20000,)
    assert cnt.min() >= 2 and cnt.max() <= 63
    ch, cl = synth.canonical_m(hi, lo, 43)
    assert np.array_equal(ch, hi) and np.array_equal(cl, lo)      # rows are canonical, as KMC lists them
    assert int(hi.max()) < (1 << 22)                                # 86 bits


def test_threaded_oracle_scan_equals_the_single_threaded_one():
    """bench.py's all-cores CPU figure: the oracle's scan loop over slices of the table with atomic counter adds.
    Wrapping sums commute, so every counter must equal the single-threaded restatement's; a tiny filter makes
    many rows collide on the same counters and heavy counts make the 16-bit cells wrap."""
    k, ref_k, bits = 35, 43, 1 << 16
    panel = synth.snp_panel(1500, 5)
    sig, valid = synth.signature_rows(panel, k)
    is_ref = np.zeros(sig.shape[0], dtype=np.uint8)
    is_ref[panel.var_allele_off[:-1]] = 1
    rows = np.zeros((int(valid.sum()), 40), dtype=np.uint8)
    rows[:, :k] = sig[valid]
    hi, lo, cnt = synth.kmer_table(panel, 60000, k, ref_k, 6)
    cnt = (cnt.astype(np.uint32) * 977) % 60000 + 1
    res = []
    for threads in (0, 1, 5):
        obf, octx, omap = ocapi.BF(bits), ocapi.BF(bits), ocapi.KMAP()
        ocapi.add_kmers(obf, omap, rows, is_ref[valid])
        obf.switch_mode()
        ocapi.ref_scan(obf, octx, panel.genome.tobytes(), k, ref_k)
        octx.switch_mode()
        if threads:
            assert ocapi.kmc_scan_packed_mt(octx, obf, omap, hi, lo, cnt, k, ref_k, threads) == threads
        else:
            ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, ref_k)
        res.append((obf.counts().copy(), dict(omap.items())))
    assert res[0][0].any() and any(res[0][1].values())
    for r in res[1:]:
        assert np.array_equal(r[0], res[0][0]) and r[1] == res[0][1]
