"""Flat-array forms of variant blocks shared by the block tests: the oracle's Python model (oracle/model.py VB objects)
-> the arrays mg_cover_blocks / mg_index_blocks and the C oracle's mo_cover_blocks / mo_index_blocks take."""
import numpy as np

from oracle import capi as ocapi
from oracle import pipeline
from oracle.model import VCFReader, flatten_vk, read_fasta


def pack_blocks(blocks, contig_base, contig_len):
    """[(VB, contig name)] -> flat arrays of mg_cover_blocks"""
    bb, bl, bo = [], [], [0]
    pos, rs, ms, pr, vo, ao, pool, canon, gts = [], [], [], [], [0], [0], bytearray(), [], []
    n_samples = None
    for vb, name in blocks:
        bb.append(contig_base.get(name, 0)); bl.append(contig_len.get(name, 0))
        for v in vb.variants:
            pos.append(v.ref_pos); rs.append(v.ref_size); ms.append(v.min_size if v.alts else v.ref_size); pr.append(int(v.is_present))
            alleles = [v.ref_sub] + v.alts
            for a, al in enumerate(alleles):
                pool += al.encode(); ao.append(len(pool)); canon.append(v.get_allele_index(al))
            vo.append(vo[-1] + len(alleles))
            g = np.zeros(len(v.genotypes), dtype=np.uint16)
            for s, ((a1, a2), ph) in enumerate(zip(v.genotypes, v.phasing)):
                assert a1 < len(alleles) and a2 < len(alleles)
                g[s] = a1 | (a2 << 7) | (int(ph) << 14)
            gts.append(g)
        bo.append(len(pos))
    n_samples = max((len(g) for g in gts), default=0)
    gt = np.zeros((len(pos), n_samples), dtype=np.uint16)
    for i, g in enumerate(gts):
        gt[i, :len(g)] = g           # non-present variants carry no genotypes; their rows are never read
    return dict(blk_ref_base=bb, blk_ref_len=bl, blk_var_off=bo, pos=pos, ref_size=rs, min_size=ms, present=pr, var_allele_off=vo,
                allele_off=ao, pool=np.frombuffer(bytes(pool), dtype=np.uint8), canon=canon, gt=gt, n_samples=n_samples)


def model_blocks(fa, vcf, opt, for_index):
    """the record loop of oracle/pipeline.py -> ([(VB, contig name, reference string)], refs, names, base)"""
    refs = read_fasta(fa, opt.strip_chr)
    names = list(refs)
    base, off = {}, 0
    for n in names:
        base[n] = off; off += len(refs[n])
    blocks = []
    for vb, reference, used in pipeline._blocks(VCFReader(vcf, opt.samples), opt, refs, for_index):
        if vb is None:
            break
        name = next((n for n in names if refs[n] is reference or refs[n] == reference), names[0])
        blocks.append((vb, name, reference))
    return blocks, refs, names, base


def model_coverages(blocks, bf, ref_bf, haploid):
    """extract_kmers (Python model) + lookup + set_coverages per block -> coverage per allele slot, concatenated"""
    want = []
    for vb, _, reference in blocks:
        km = vb.extract_kmers(reference, haploid)
        ks, is_ref, so, ao = flatten_vk(km, [len(v.alts) + 1 for v in vb.variants])
        w = ocapi.lookup_weights(bf, ref_bf, ocapi.rows_from_kmers(ks)[0], np.array(is_ref, np.uint8)) if ks else np.zeros(0, np.int32)
        want.append(ocapi.set_coverages(w, so, ao))
    return np.concatenate(want) if want else np.zeros(0, np.uint32)
