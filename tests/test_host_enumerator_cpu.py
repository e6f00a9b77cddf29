"""The C++ host enumerator (malva_amd/host/block.hpp, io.hpp) against the oracle's block model:
`malva-geno dump-kmers` must list exactly the blocks, variants and signature k-mers that
oracle/model.py derives (no GPU involved)."""
import os
import subprocess

import pytest

import vcf_synth
from oracle import pipeline
from oracle.model import VCFReader, read_fasta

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin", "malva-geno")


def oracle_dump(fa, vcf, opt, for_index):
    refs = read_fasta(fa, opt.strip_chr)
    reader = VCFReader(vcf, opt.samples)
    out = []
    for vb, reference, _ in pipeline._blocks(reader, opt, refs, for_index):
        if vb is None:
            break
        kmers = vb.extract_kmers(reference, opt.haploid)
        lone = len(vb.variants) == 1 and vb.variants[0].ref_size < opt.k and all(len(a) < opt.k for a in vb.variants[0].alts)
        out.append("BLOCK %d%s" % (len(vb.variants), " lone" if lone else ""))
        for vi, v in enumerate(vb.variants):
            out.append("VAR %s %d %s%s present=%d" % (v.seq_name, v.ref_pos + 1, v.ref_sub, "".join(" " + a for a in v.alts), int(v.is_present)))
            for a in sorted(kmers.get(vi, {})):
                for l in sorted(",".join(sig) for sig in kmers[vi][a]):
                    out.append("SIG %d %s" % (a, l))
    return "\n".join(out) + "\n"


def cli_dump(fa, vcf, opt, for_index, pool=None):
    if not os.path.exists(BIN):
        pytest.fail("bin/malva-geno not built: run `make cli`")
    cmd = [BIN, "dump-kmers", "-k", str(opt.k), "-r", str(opt.ref_k), "-f", opt.freq_key]
    if opt.haploid:
        cmd.append("-1")
    if opt.strip_chr:
        cmd.append("-p")
    cmd += [fa, vcf, "index" if for_index else "call"]
    env = dict(os.environ) if pool is None else dict(os.environ, MALVA_GENO_VCF_POOL=str(pool))   # record decoding inline / by the thread pool
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("for_index", [True, False])
def test_haploid_example(golden_dir, for_index):
    opt = pipeline.Options(haploid=True, bf_size=1 << 33)
    fa, vcf = os.path.join(golden_dir, "haploid.fa"), os.path.join(golden_dir, "haploid.vcf.gz")
    assert cli_dump(fa, vcf, opt, for_index) == oracle_dump(fa, vcf, opt, for_index)


@pytest.mark.parametrize("seed,haploid,k,strip", [(1, False, 35, False), (2, True, 35, True), (3, False, 31, True), (4, False, 21, False),
                                                  (5, True, 63, False)])
def test_clustered_variants(tmp_path, seed, haploid, k, strip):
    prefix = str(tmp_path / "case")
    vcf_synth.make_case(prefix, seed, haploid=haploid, k=k)
    opt = pipeline.Options(haploid=haploid, k=k, ref_k=k + 8, strip_chr=strip)
    for for_index in (True, False):
        got = cli_dump(prefix + ".fa", prefix + ".vcf", opt, for_index)
        want = oracle_dump(prefix + ".fa", prefix + ".vcf", opt, for_index)
        assert got == want
    assert "SIG 2 " in want and " lone" in want          # multi-allelic and lone blocks occur in every case
    if seed in (1, 4):
        assert "," in want.split("SIG", 1)[1]            # ... and so does the sliding (allele >= k) signature


@pytest.mark.parametrize("haploid", [False, True])
def test_positions_beyond_2_to_the_25_float_near(tmp_path, haploid):
    """Beyond 2^24 the reference's `are_near` no longer computes in integers (float promotion, var_block.hpp:417-423):
    block cuts and chain walks of the C++ host enumerator must follow it -- the case holds dozens of neighbour pairs on
    which the float answer and the exact one differ, in both directions."""
    prefix = str(tmp_path / "far")
    _, records, pairs = vcf_synth.make_far_case(prefix, 11, haploid=haploid)
    assert pairs > 30
    opt = pipeline.Options(haploid=haploid)
    for for_index in (True, False):
        want = oracle_dump(prefix + ".fa", prefix + ".vcf", opt, for_index)
        assert cli_dump(prefix + ".fa", prefix + ".vcf", opt, for_index) == want
    assert want.count("BLOCK") > 400


def _write_case(tmp_path, gt_rows, samples, fmt="GT", seed=7, spacing=9):
    import numpy as np
    rng = np.random.default_rng(seed)
    seq = "".join(rng.choice(list("ACGT"), size=400 + spacing * len(gt_rows)))
    fa, vcf = str(tmp_path / "g.fa"), str(tmp_path / "g.vcf")
    with open(fa, "w") as fh:
        fh.write(">1\n%s\n" % seq)
    with open(vcf, "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##INFO=<ID=AF,Number=A,Type=Float,Description=\"af\">\n"
                 "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n##contig=<ID=1,length=%d>\n" % len(seq))
        fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples) + "\n")
        for i, cols in enumerate(gt_rows):
            pos = 200 + spacing * i
            ref = seq[pos]
            alts = [b for b in "ACGT" if b != ref][:2]
            fh.write("1\t%d\t.\t%s\t%s\t.\t.\tAF=0.2,0.1\t%s\t%s\n" % (pos + 1, ref, ",".join(alts), fmt, "\t".join(cols)))
    return fa, vcf


@pytest.mark.parametrize("haploid", [False, True])
def test_gt_column_forms(tmp_path, haploid):
    """What the one-pass GT reader has to take: missing alleles, mixed ploidy inside a record (the padded array the
    reference gets from htslib), multi-digit and signed tokens, extra sub-fields, GT not first in FORMAT."""
    samples = ["S%d" % i for i in range(6)]
    rows = [
        ["0|1", "1|0", "0/1", "./.", ".", "2|1"],
        ["0", "1", "2", ".", "0", "1"],                        # haploid columns: the reference pairs each with the NEXT sample's value
        ["0|1", "1", "0/0", "1|2", "2", "0"],                  # mixed ploidy: padded with vector_end
        ["0|1:9:x", "1|1:3", "0/2:.", ".:.", "1|0", "0|0:7"],  # extra sub-fields
        ["00|1", "+1|0", "1|02", "0|0", "0|1", "2|2"],         # atoi forms
        ["0|1|1", "1|0|0", "0|0|0", "0|0|1", "1|1|1", "0|2|0"],  # triploid: only the first two are used
    ]
    fa, vcf = _write_case(tmp_path, rows, samples)
    opt = pipeline.Options(haploid=haploid, k=21, ref_k=29)
    for for_index in (True, False):
        want = oracle_dump(fa, vcf, opt, for_index)
        assert cli_dump(fa, vcf, opt, for_index, pool=0) == want
        assert cli_dump(fa, vcf, opt, for_index, pool=1) == want
    # GT as the second FORMAT key
    rows2 = [["5:" + c.split(":")[0] for c in r] for r in rows[:3]]
    fa, vcf = _write_case(tmp_path, rows2, samples, fmt="DP:GT")
    assert cli_dump(fa, vcf, opt, False) == oracle_dump(fa, vcf, opt, False)


@pytest.mark.parametrize("haploid,phased", [(True, True), (False, True), (False, False)])
def test_large_panel_sparse_sample_walk(tmp_path, haploid, phased):
    """A panel of 3,000 samples where most are all-reference: the enumerator walks only the samples that carry a
    non-reference allele on the chain, and must still find every pick the reference's all-samples loop finds
    (including the all-reference pick, and picks that exist only through unphased mixing)."""
    import numpy as np
    rng = np.random.default_rng(3)
    n_s, n_v = 3000, 90      # one block of 90 variants: enumerated by the thread pool (>= 64)
    samples = ["P%d" % i for i in range(n_s)]
    sep = "|" if phased else "/"
    rows = []
    for v in range(n_v):
        carriers = set(rng.choice(n_s, size=int(rng.integers(0, 12)), replace=False).tolist())
        if v % 17 == 3:
            carriers = set(range(n_s)) - set(rng.choice(n_s, size=5, replace=False).tolist())   # a common variant
        cols = []
        for s in range(n_s):
            if s in carriers:
                a, b = int(rng.integers(0, 3)), int(rng.integers(1, 3))
                cols.append(str(b) if haploid else "%d%s%d" % (a, sep, b))
            else:
                cols.append("0" if haploid else "0%s0" % sep)
        rows.append(cols)
    fa, vcf = _write_case(tmp_path, rows, samples, spacing=7)
    opt = pipeline.Options(haploid=haploid, k=21, ref_k=29)
    want = oracle_dump(fa, vcf, opt, True)
    assert cli_dump(fa, vcf, opt, True) == want            # 3,000 sample columns: the pool decodes
    assert cli_dump(fa, vcf, opt, True, pool=0) == want
