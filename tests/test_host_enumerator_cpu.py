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


def cli_dump(fa, vcf, opt, for_index):
    if not os.path.exists(BIN):
        pytest.fail("bin/malva-geno not built: run `make cli`")
    cmd = [BIN, "dump-kmers", "-k", str(opt.k), "-r", str(opt.ref_k), "-f", opt.freq_key]
    if opt.haploid:
        cmd.append("-1")
    if opt.strip_chr:
        cmd.append("-p")
    cmd += [fa, vcf, "index" if for_index else "call"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
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
