"""End-to-end restatement of `malva-geno index` and `malva-geno call`.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Restates main.cpp:251-419
(index_main) and main.cpp:421-594 (call_main) over the C oracle (BF / KMAP /
likelihoods) and the Python block model, with the on-disk index replaced by
in-memory objects and the KMC database by an iterable of (ref_k-mer, count).
"""
import math
from dataclasses import dataclass
from typing import Dict, Iterable, List, Tuple

import numpy as np

from . import capi
from .model import VB, VCFReader, Variant, flatten_vk, read_fasta


@dataclass
class Options:
    """argument_parser.hpp:51-66 defaults; bf_size is in bits (-b N => N * 2^33)."""
    k: int = 35
    ref_k: int = 43
    error_rate: float = 0.001
    samples: str = "-"
    freq_key: str = "AF"
    max_coverage: int = 200
    bf_size: int = 1 << 35
    strip_chr: bool = False
    uniform: bool = False
    verbose: bool = False
    haploid: bool = False


@dataclass
class Index:
    context_bf: capi.BF
    bf: capi.BF
    ref_bf: capi.KMAP
    used_seq_names: List[str]


def _blocks(reader: VCFReader, opt: Options, refs: Dict[str, str], for_index: bool):
    """The record loop shared by main.cpp:309-370 and :522-579: yields
    (VB, reference string of `last_seq_name`, used-name bookkeeping).
    Keeps the reference's control flow, including that `last_seq_name` is only
    refreshed when a block is flushed."""
    vb = VB(opt.k, opt.error_rate)
    last_seq_name = ""
    used = []
    for v in reader.records(opt.freq_key, opt.uniform):
        if last_seq_name == "":
            last_seq_name = v.seq_name
            used.append(last_seq_name)
        if for_index:
            if (not v.has_alts) or (not v.is_present):
                continue
        elif not v.has_alts:
            continue
        if vb.empty():
            vb.add_variant(v)
            continue
        if (not vb.is_near_to_last(v)) or last_seq_name != v.seq_name:
            yield vb, refs.get(last_seq_name, ""), used
            vb = VB(opt.k, opt.error_rate)
            if last_seq_name != v.seq_name:
                last_seq_name = v.seq_name
                used.append(last_seq_name)
        vb.add_variant(v)
    if not vb.empty():
        yield vb, refs.get(last_seq_name, ""), used
    else:
        yield None, "", used


def index(fasta_path: str, vcf_path: str, opt: Options) -> Index:
    refs = read_fasta(fasta_path, opt.strip_chr)
    reader = VCFReader(vcf_path, opt.samples)
    bf = capi.BF(opt.bf_size)
    ref_bf = capi.KMAP()
    context_bf = capi.BF(opt.bf_size)
    used: List[str] = []
    for vb, reference, used in _blocks(reader, opt, refs, True):
        if vb is None:
            break
        kmers = vb.extract_kmers(reference, opt.haploid)
        for per in kmers.values():              # add_kmers_to_bf, main.cpp:122-144
            for a, sigs in per.items():
                for sig in sigs:
                    for km in sig:
                        (ref_bf if a == 0 else bf).add_key(km.encode())
    bf.switch_mode()                             # main.cpp:378
    for name in used:                            # main.cpp:383-401
        capi.ref_scan(bf, context_bf, refs.get(name, "").encode(), opt.k, opt.ref_k)
    context_bf.switch_mode()                     # main.cpp:404
    return Index(context_bf, bf, ref_bf, list(used))


def header_text(reader: VCFReader, verbose: bool) -> str:
    """print_cleaned_header, main.cpp:190-219, as htslib renders it."""
    lines = list(reader.header_lines)
    pass_line = '##FILTER=<ID=PASS,Description="All filters passed">'
    if not any(l.startswith("##FILTER=<ID=PASS,") or l.startswith("##FILTER=<ID=PASS>") for l in lines):
        lines.insert(1 if lines and lines[0].startswith("##fileformat") else 0, pass_line)

    def append(line):
        tag = line[: line.index(",")] + ","       # "##FORMAT=<ID=GT,"
        if not any(l.startswith(tag) or l.startswith(tag[:-1] + ">") for l in lines):
            lines.append(line)

    append('##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">')
    append('##FORMAT=<ID=GQ,Number=1,Type=Integer,Description="Genotype Quality">')
    if verbose:
        append('##INFO=<ID=COVS,Number=R,Type=Integer,Description="Allele coverages">')
        append('##INFO=<ID=GTS,Number=.,Type=String,Description="Genotypes Likelihood">')
    lines.append("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tDONOR")
    return "\n".join(lines) + "\n"


def _fmt_float(q: float) -> str:
    """ostream << float with default precision 6"""
    s = "%.6g" % q
    return s


def genotype_block(vb: VB, idx: Index, reference: str, opt: Options, raw: List = None) -> List[str]:
    """extract_kmers + set_coverages + genotype + output_variants for one block
    (main.cpp:556-559).  Returns the VCF lines.  If `raw` is a list, appends per
    variant (coverages, [(g1,g2,value)...], best index, GQ) for parity checks."""
    kmers = vb.extract_kmers(reference, opt.haploid)
    nal = [len(v.alts) + 1 for v in vb.variants]
    ks, is_ref, sig_off, al_off = flatten_vk(kmers, nal)
    if ks:
        rows, _ = capi.rows_from_kmers(ks)
        w = capi.lookup_weights(idx.bf, idx.ref_bf, rows, np.array(is_ref, dtype=np.uint8))
    else:
        w = np.zeros(0, dtype=np.int32)
    cov = capi.set_coverages(w, sig_off, al_off)
    # set_coverages only touches alleles that have signatures (main.cpp:157-181);
    # the others keep their initial 0 -- identical to an empty signature list.
    out = []
    a0 = 0
    for v in vb.variants:
        A = len(v.alts) + 1
        v.coverages = [int(x) for x in cov[a0:a0 + A]]
        a0 += A
        gts = capi.genotype(v.coverages, v.frequencies if v.frequencies else [0.0] * A, vb.error_rate,
                            opt.max_coverage, opt.haploid)
        bi, gq, norm = capi.select_gt([g[2] for g in gts])

        def gname(g):
            return str(g[0]) if g[1] < 0 else "%d/%d" % (g[0], g[1])

        best = gname(gts[bi]) if bi >= 0 else ("0" if opt.haploid else "0/0")
        info = "."
        if opt.verbose:                          # var_block.hpp:358-391
            info = "COVS=" + ",".join(str(c) for c in v.coverages)
            info += ";GTS=" + ",".join("%s:%s" % (gname(g), _to_string(norm[i])) for i, g in enumerate(gts))
        qual = "." if math.isnan(v.quality) else _fmt_float(v.quality)
        out.append("%s\t%d\t%s\t%s\t%s\t%s\t%s\t%s\tGT:GQ\t%s:%d" % (
            v.seq_name, v.ref_pos + 1, v.idx, v.ref_sub, ",".join(v.alts), qual, v.filter, info, best, gq))
        if raw is not None:
            raw.append((list(v.coverages), gts, bi, gq))
    return out


def _to_string(x: float) -> str:
    """std::to_string(double) == printf("%f")"""
    if math.isnan(x):
        return "-nan" if math.copysign(1.0, x) < 0 else "nan"
    return "%f" % x


def scan(idx: Index, kmers: Iterable[Tuple[bytes, int]], opt: Options):
    """KMC scan, main.cpp:482-500"""
    ks, cs = [], []
    for km, c in kmers:
        ks.append(km); cs.append(c)
    if not ks:
        return
    rows, _ = capi.rows_from_kmers(ks)
    capi.kmc_scan(idx.context_bf, idx.bf, idx.ref_bf, rows, np.array(cs, dtype=np.uint32), opt.k, opt.ref_k)


def call(fasta_path: str, vcf_path: str, idx: Index, kmers: Iterable[Tuple[bytes, int]], opt: Options,
         raw: List = None) -> str:
    refs = read_fasta(fasta_path, opt.strip_chr)
    scan(idx, kmers, opt)
    hdr_reader = VCFReader(vcf_path, "-")
    text = [header_text(hdr_reader, opt.verbose)]
    reader = VCFReader(vcf_path, opt.samples)
    for vb, reference, _ in _blocks(reader, opt, refs, False):
        if vb is None:
            break
        for line in genotype_block(vb, idx, reference, opt, raw):
            text.append(line + "\n")
    return "".join(text)
