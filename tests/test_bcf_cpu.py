"""BCF input: malva_amd/host/io.hpp translates BCF2 records into the VCF lines they encode, so the panel
decoder sees one format.  Host code only (`malva-geno dump-kmers` prints every block's variants and signature k-mers):
the same panel as text and as BCF (written by tests/bcf_writer.py, an independent reading of the published layout --
parity unpinned, no htslib here) must enumerate identically."""
import os
import subprocess

import pytest

import bcf_writer
import vcf_synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin", "malva-geno")


def _dump(args):
    r = subprocess.run([BIN, "dump-kmers"] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("with_idx", [False, True])
def test_haploid_example_as_bcf(tmp_path, golden_dir, with_idx):
    fa, vcf = os.path.join(golden_dir, "haploid.fa"), os.path.join(golden_dir, "haploid.vcf.gz")
    bcf = str(tmp_path / "haploid.bcf")
    bcf_writer.vcf_to_bcf(vcf, bcf, with_idx=with_idx)
    assert open(bcf, "rb").read(4) == b"\x1f\x8b\x08\x04"                      # BGZF
    for mode in ("index", "call"):
        want = _dump(["-1", "-k", "35", "-f", "AF", fa, vcf, mode])
        assert _dump(["-1", "-k", "35", "-f", "AF", fa, bcf, mode]) == want and want.count("VAR ") > 400


@pytest.mark.parametrize("seed,haploid", [(3, False), (4, True)])
def test_clustered_multiallelic_panel_as_bcf(tmp_path, seed, haploid):
    """phased and unphased genotypes, missing alleles, multi-allelic records, indels, several contigs"""
    prefix = str(tmp_path / "case")
    vcf_synth.make_case(prefix, seed, haploid=haploid, k=35, n_clusters=120, vcf_strip_chr=True)
    bcf = prefix + ".bcf"
    bcf_writer.vcf_to_bcf(prefix + ".vcf", bcf, with_idx=bool(seed & 1))
    args = ["-k", "35", "-p"] + (["-1"] if haploid else [])
    want = _dump(args + [prefix + ".fa", prefix + ".vcf", "call"])
    assert _dump(args + [prefix + ".fa", bcf, "call"]) == want and want.count("SIG ") > 500


def test_truncated_bcf_is_an_error(tmp_path, golden_dir):
    bcf = str(tmp_path / "t.bcf")
    bcf_writer.vcf_to_bcf(os.path.join(golden_dir, "haploid.vcf.gz"), bcf)
    data = open(bcf, "rb").read()
    open(bcf, "wb").write(data[:len(data) * 2 // 3])
    r = subprocess.run([BIN, "dump-kmers", "-1", os.path.join(golden_dir, "haploid.fa"), bcf, "call"], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


@pytest.mark.parametrize("group", [None, "70000", "1"])
def test_bgzipped_vcf_is_inflated_member_by_member(tmp_path, group):
    """a .vcf.gz as bgzip writes it (BGZF: what tabix needs, what panels ship as): the reader inflates the members of a group side
    by side, each to its place; lines cross members and groups (`group`: inflated bytes per group, forced small); the text
    panel, the BGZF one and the BGZF one through zlib's gzread (MALVA_GENO_NO_BGZF) must enumerate identically; a member
    with a flipped bit is an error"""
    prefix = str(tmp_path / "case")
    vcf_synth.make_case(prefix, 9, haploid=False, k=35, n_clusters=300, n_samples=40, vcf_strip_chr=True)
    gz = prefix + ".bgzf.vcf.gz"
    data = open(prefix + ".vcf", "rb").read()
    with open(gz, "wb") as out:
        bcf_writer._bgzf(data, out)
    assert len(data) > 3 * 0xFF00                                              # several members
    args = ["-k", "35", "-p", prefix + ".fa"]
    want = _dump(args + [prefix + ".vcf", "call"])
    env = dict(os.environ, MALVA_GENO_VCF_POOL="1")
    if group:
        env["MALVA_GENO_BGZF_GROUP"] = group
    for e in (env, dict(env, MALVA_GENO_NO_BGZF="1")):
        r = subprocess.run([BIN, "dump-kmers"] + args + [gz, "call"], capture_output=True, text=True, timeout=900, env=e)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout == want and want.count("SIG ") > 1000
    bad = bytearray(open(gz, "rb").read())
    bad[len(bad) // 2] ^= 0x10
    open(gz, "wb").write(bytes(bad))
    r = subprocess.run([BIN, "dump-kmers"] + args + [gz, "call"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0
