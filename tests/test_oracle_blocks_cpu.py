"""The C oracle's block enumerator (oracle/malva_oracle.c: mo_cut_blocks, mo_cover_blocks, mo_index_blocks -- the form that
runs at bench sizes) against the oracle's Python model (oracle/model.py, pinned end to end by the reference's haploid
golden): same block cuts, same coverages, same index, on clustered panels with SNPs, MNPs, indels, alleles of k bases and
more, overlapping records, unphased genotypes, N / IUPAC in the reference, and positions beyond 2^25."""
import numpy as np
import pytest

import vcf_synth
from block_util import model_blocks, model_coverages, pack_blocks
from oracle import capi as ocapi
from oracle import pipeline


def _weights(idx_bf, idx_map, blocks, haploid):
    """made-up counters for every signature k-mer of the panel (no scan needed to exercise lookup + coverage)"""
    for vb, _, reference in blocks:
        for per in vb.extract_kmers(reference, haploid).values():
            for a, sigs in per.items():
                for sig in sigs:
                    for s_ in sig:
                        w = 1 + ocapi.xxh3_64(s_.encode()) % 97
                        if a == 0:
                            idx_map.increment(s_.encode(), w)
                        else:
                            idx_bf.increment(s_.encode(), w)


@pytest.mark.parametrize("seed,haploid,k,dense", [(71, False, 35, False), (72, True, 35, False), (73, False, 31, False), (74, False, 21, False),
                                                  (75, True, 63, False), (76, False, 35, True), (77, True, 35, True)])
def test_c_enumerator_equals_the_python_model(tmp_path, seed, haploid, k, dense):
    prefix = str(tmp_path / "case")
    vcf_synth.make_case(prefix, seed, haploid=haploid, k=k, n_clusters=30 if dense else 100, vcf_strip_chr=True, dense=dense,
                        n_samples=4 if dense else 5)
    opt = pipeline.Options(haploid=haploid, k=k, ref_k=k + 8, bf_size=1 << 24, strip_chr=True)
    fa, vcf = prefix + ".fa", prefix + ".vcf"
    # ---- index: same bf bits, same exact-map keys ----
    idx = pipeline.index(fa, vcf, opt)
    blocks_i, refs, names, base = model_blocks(fa, vcf, opt, True)
    args = pack_blocks([(vb, n) for vb, n, _ in blocks_i], base, {n: len(refs[n]) for n in names})
    reference = "".join(refs[n] for n in names).encode()
    bf, kmap = ocapi.BF(opt.bf_size), ocapi.KMAP()
    n_added = ocapi.index_blocks(bf, kmap, reference, **args, haploid=haploid, k=k)
    bf.switch_mode()
    assert n_added > 200
    assert np.array_equal(bf.set_positions(), idx.bf.set_positions())
    assert sorted(k_ for k_, _ in kmap.items()) == sorted(k_ for k_, _ in idx.ref_bf.items())
    # canon as the model's get_allele_index gives it
    assert np.array_equal(ocapi.allele_canon(args["var_allele_off"], args["allele_off"], args["pool"]), np.array(args["canon"], dtype=np.uint8))
    # ---- call: same cuts, same coverages ----
    blocks_c, _, _, _ = model_blocks(fa, vcf, opt, False)
    _weights(idx.bf, idx.ref_bf, blocks_c, haploid)
    args = pack_blocks([(vb, n) for vb, n, _ in blocks_c], base, {n: len(refs[n]) for n in names})
    want = model_coverages(blocks_c, idx.bf, idx.ref_bf, haploid)
    stats = {}
    got = ocapi.cover_blocks(idx.bf, idx.ref_bf, reference, **args, haploid=haploid, k=k, stats=stats)
    assert np.array_equal(got, want)
    assert (want > 0).sum() > 50 and stats["kmers"] >= stats["signatures"] > 100
    # the cut: contig ids in the order the record loop sees them (the first block's sequence is `last_seq_name`'s first value)
    cid = np.repeat([names.index(n) for _, n, _ in blocks_c], np.diff(args["blk_var_off"])).astype(np.uint32)
    off, bc = ocapi.cut_blocks(args["pos"], args["ref_size"], args["min_size"], cid, k)
    assert np.array_equal(off, np.array(args["blk_var_off"], dtype=np.uint32))
    assert [names[c] for c in bc] == [n for _, n, _ in blocks_c]


@pytest.mark.parametrize("haploid", [False, True])
def test_c_enumerator_beyond_2_to_the_25(tmp_path, haploid):
    """positions where are_near runs in float (var_block.hpp:417-423): the C walk and the Python walk agree"""
    k = 35
    prefix = str(tmp_path / "far")
    seq, records, pairs = vcf_synth.make_far_case(prefix, 12, haploid=haploid, span=400_000, n_clusters=150)
    assert pairs > 10
    opt = pipeline.Options(haploid=haploid, k=k, ref_k=43, bf_size=1 << 24)
    from oracle.model import VCFReader
    refs = {"1": seq}
    blocks = []
    for vb, reference, _ in pipeline._blocks(VCFReader(prefix + ".vcf", "-"), opt, refs, False):
        if vb is None:
            break
        blocks.append((vb, "1", reference))
    bf, kmap = ocapi.BF(opt.bf_size), ocapi.KMAP()
    for vb, _, reference in blocks:
        for per in vb.extract_kmers(reference, haploid).values():
            for a, sigs in per.items():
                for sig in sigs:
                    for s_ in sig:
                        (kmap if a == 0 else bf).add_key(s_.encode())
    bf.switch_mode()
    _weights(bf, kmap, blocks, haploid)
    args = pack_blocks([(vb, n) for vb, n, _ in blocks], {"1": 0}, {"1": len(seq)})
    want = model_coverages(blocks, bf, kmap, haploid)
    got = ocapi.cover_blocks(bf, kmap, seq.encode(), **args, haploid=haploid, k=k)
    assert np.array_equal(got, want) and (want > 0).sum() > 300
    off, _ = ocapi.cut_blocks(args["pos"], args["ref_size"], args["min_size"], np.zeros(len(args["pos"]), np.uint32), k)
    assert np.array_equal(off, np.array(args["blk_var_off"], dtype=np.uint32))
