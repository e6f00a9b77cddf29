"""End to end through the C++ driver on the GPU: `malva-geno index` + `malva-geno call`
(bin/malva-geno over libmalva_hip.so) against the reference's golden VCF and against the oracle
pipeline on clustered synthetic panels."""
import os
import re
import shutil
import subprocess
import time

import pytest

import vcf_synth
from oracle import kmc_standin, pipeline

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "bin", "malva-geno")


def run_cli(args, **kw):
    if not os.path.exists(BIN):
        pytest.fail("bin/malva-geno not built: run `make cli`")
    r = subprocess.run([BIN] + args, capture_output=True, text=True, timeout=900, **kw)
    assert r.returncode == 0, r.stderr[-3000:]
    if os.environ.get("MALVA_CLI_LOG"):            # phase timings of the CLI (stderr), for tools/ and DESIGN.md
        with open(os.environ["MALVA_CLI_LOG"], "a") as fh:
            fh.write("== malva-geno %s\n%s\n" % (" ".join(args[:1]), r.stderr))
    return r.stdout


def test_haploid_example_byte_identical(tmp_path, golden_dir):
    """README.md:131-140: MALVA -1 -k 35 -r 43 -b 1 -f AF haploid.fa haploid.vcf haploid.fq"""
    fa = os.path.join(golden_dir, "haploid.fa")
    vcf = str(tmp_path / "haploid.vcf.gz")
    shutil.copy(os.path.join(golden_dir, "haploid.vcf.gz"), vcf)
    prefix = str(tmp_path / "haploid_malva43.kmercount")
    with open(prefix + ".txt", "w") as fh:
        for km, c in kmc_standin.count_fastq(os.path.join(golden_dir, "haploid.fq"), 43):
            fh.write("%s\t%d\n" % (km.decode(), c))
    common = ["-1", "-k", "35", "-r", "43", "-b", "1", "-f", "AF", fa, vcf, prefix]
    run_cli(["index"] + common)
    out = run_cli(["call"] + common)
    assert out == open(os.path.join(golden_dir, "haploid.malva.vcf")).read()
    # every block through the resident record loop (the default), or lone records through the fused mg_call_isolated: same bytes
    assert run_cli(["call"] + common, env=dict(os.environ, MALVA_GENO_ISO_PATH="1")) == out


@pytest.mark.parametrize("seed,haploid,verbose,k,ref_k", [(11, False, True, 35, 43), (12, True, True, 35, 43), (13, False, False, 31, 45),
                                                            (14, False, True, 35, 63)])
def test_clustered_panel_matches_oracle(tmp_path, seed, haploid, verbose, k, ref_k):
    prefix = str(tmp_path / "case")
    contigs, records = vcf_synth.make_case(prefix, seed, haploid=haploid, k=k, n_clusters=60, vcf_strip_chr=True)
    table = str(tmp_path / "donor.kmers")
    vcf_synth.donor_table(contigs, records, ref_k, seed, table + ".txt")
    opt = pipeline.Options(haploid=haploid, verbose=verbose, k=k, ref_k=ref_k, bf_size=1 << 33, strip_chr=True)
    idx = pipeline.index(prefix + ".fa", prefix + ".vcf", opt)
    kmers = [(l.split()[0].encode(), int(l.split()[1])) for l in open(table + ".txt")]
    want = pipeline.call(prefix + ".fa", prefix + ".vcf", idx, kmers, opt)
    args = ["-k", str(k), "-r", str(ref_k), "-b", "1", "-p"] + (["-1"] if haploid else []) + (["-v"] if verbose else [])
    args += [prefix + ".fa", prefix + ".vcf", table]
    run_cli(["index"] + args)
    got = run_cli(["call"] + args)
    assert run_cli(["call"] + args, env=dict(os.environ, MALVA_GENO_ISO_PATH="1")) == got    # (lone records through the fused entry instead of tier 1)
    assert got.count("\n") == want.count("\n")
    strip = lambda s: re.sub(r";GTS=[^\t]*", "", s)         # (the likelihood list only: COVS before it and GT:GQ behind it stay)
    assert strip(got) == strip(want)                 # header, records, COVS, GT, GQ: identical
    if got != want:                                   # GTS holds printf("%f") of doubles that may differ in the last bit of exp()
        for a, b in zip(got.split("\n"), want.split("\n")):
            if a != b:
                fa = [float(x.split(":")[1]) for x in a.split("GTS=")[1].split("\t")[0].split(",")]
                fb = [float(x.split(":")[1]) for x in b.split("GTS=")[1].split("\t")[0].split(",")]
                assert all(abs(x - y) <= 1.000001e-6 or (x != x and y != y) for x, y in zip(fa, fb)), (a, b)
    assert sum(1 for l in got.split("\n") if l and not l.startswith("#") and not l.endswith(":0")) > 20
    # the k-mer dump parsed by many threads in 700-byte tasks (a line belongs to the task holding its first byte),
    # with Windows line ends and blank lines thrown in, and through the gzip reader: same output, byte for byte
    env = dict(os.environ, MALVA_GENO_TABLE_TASK="700")
    assert run_cli(["call"] + args, env=env) == got
    lines = open(table + ".txt").read().split("\n")
    with open(table + ".txt", "w") as fh:
        fh.write("\r\n".join(lines[:50]) + "\r\n\n\n" + "\n".join(lines[50:]))
    assert run_cli(["call"] + args, env=env) == got
    import gzip
    with open(table + ".txt", "rb") as src:
        data = src.read()
    with gzip.open(table + ".txt", "wb") as dst:
        dst.write(data)
    assert run_cli(["call"] + args) == got
    # --gpus 3: the table sharded over three contexts (pieces of the gzip stream go round them), one exchange, the
    # record batches split between them -- on a one-GPU box the three contexts share the device
    # (MALVA_GENO_SHARE_DEVICE) and the exchange is the kernel sum instead of RCCL; same bytes out
    import torch
    share = {} if torch.cuda.device_count() >= 3 else {"MALVA_GENO_SHARE_DEVICE": "1"}
    env3 = dict(os.environ, MALVA_GENO_BATCH="61", **share)
    assert run_cli(["call", "--gpus", "3"] + args, env=env3) == got
    with open(table + ".txt", "wb") as dst:                      # and through the mapped, multi-threaded reader
        dst.write(data)
    assert run_cli(["call", "-g", "3"] + args, env=dict(env3, MALVA_GENO_TABLE_TASK="700")) == got


def test_sars_cov2_panel_config_c1(tmp_path, golden_dir):
    """BASELINE config C1: example/reference_sarsCov2.fasta + example/sars_cov2.vcf.gz (15,154 records x 27,934 haploid
    samples, 3,479 multi-allelic), k=35 r=43 b=1, the haploid example's reads as the sample.  The compiled reference,
    run on these inputs during the survey, emits 15,154 records with exactly two non-reference
    calls: 17747 C>T 1:94 and 17858 A>G 1:100."""
    fa = os.path.join(golden_dir, "reference_sarsCov2.fasta")
    vcf = str(tmp_path / "sars_cov2.vcf.gz")
    shutil.copy(os.path.join(golden_dir, "sars_cov2.vcf.gz"), vcf)
    prefix = str(tmp_path / "sample.kmercount")
    with open(prefix + ".txt", "w") as fh:
        for km, c in kmc_standin.count_fastq(os.path.join(golden_dir, "haploid.fq"), 43):
            fh.write("%s\t%d\n" % (km.decode(), c))
    common = ["-1", "-k", "35", "-r", "43", "-b", "1", "-f", "AF", fa, vcf, prefix]
    run_cli(["index"] + common)
    out = run_cli(["call"] + common)
    recs = [l.split("\t") for l in out.split("\n") if l and not l.startswith("#")]
    assert len(recs) == 15154
    nonref = [(r[1], r[3], r[4], r[9]) for r in recs if not r[9].startswith("0:")]
    assert nonref == [("17747", "C", "T", "1:94"), ("17858", "A", "G", "1:100")]


def test_c5_like_panel_many_batches(tmp_path):
    """BASELINE config C5 shape: indel/MNP-heavy clustered blocks, k=35 r=63, diploid and haploid, a few thousand
    records pushed through the driver in batches of 97 records (batch seams inside and between blocks)."""
    for seed, haploid in ((21, False), (22, True)):
        prefix = str(tmp_path / ("c5_%d" % seed))
        contigs, records = vcf_synth.make_case(prefix, seed, haploid=haploid, k=35, n_clusters=700, vcf_strip_chr=True)
        table = str(tmp_path / ("donor_%d.kmers" % seed))
        vcf_synth.donor_table(contigs, records, 63, seed, table + ".txt")
        opt = pipeline.Options(haploid=haploid, verbose=True, k=35, ref_k=63, bf_size=1 << 33, strip_chr=True)
        idx = pipeline.index(prefix + ".fa", prefix + ".vcf", opt)
        kmers = [(l.split()[0].encode(), int(l.split()[1])) for l in open(table + ".txt")]
        want = pipeline.call(prefix + ".fa", prefix + ".vcf", idx, kmers, opt)
        args = ["-k", "35", "-r", "63", "-b", "1", "-p", "-v"] + (["-1"] if haploid else []) + [prefix + ".fa", prefix + ".vcf", table]
        run_cli(["index"] + args)
        got = run_cli(["call"] + args, env=dict(os.environ, MALVA_GENO_BATCH="97"))
        assert got == want
        assert got.count("\n") > 1500


@pytest.mark.parametrize("haploid", [False, True])
def test_blocks_cut_on_the_device_or_on_the_host(tmp_path, haploid):
    """`index` and `call` cut their blocks on the device (mg_cut_blocks over batches of kept records); with batches of 7
    records nearly every block straddles a batch seam, with MALVA_GENO_HOST_CUT=1 the cuts are made record by record on
    the host as in round 1: byte-identical output, equal to the oracle's.  The panel's first record is a symbolic-only
    one on another sequence than the records behind it: the reference then flushes the first block under THAT name
    (main.cpp:319-323, 341-356), and so must both cutters."""
    prefix = str(tmp_path / "cut")
    contigs, records = vcf_synth.make_case(prefix, 77, haploid=haploid, k=35, n_clusters=160, vcf_strip_chr=True)
    lines = open(prefix + ".vcf").read().split("\n")
    first = next(i for i, l in enumerate(lines) if l and not l.startswith("#"))
    n_samples = len(lines[first].split("\t")) - 9
    gt0 = "0" if haploid else "0|0"
    lines.insert(first, "\t".join(["2", "50", ".", contigs["chr2"][49], "<DEL>", ".", ".", "AF=0.1", "GT"] + [gt0] * n_samples))
    open(prefix + ".vcf", "w").write("\n".join(lines))
    table = str(tmp_path / "donor.kmers")
    vcf_synth.donor_table(contigs, records, 43, 78, table + ".txt")
    opt = pipeline.Options(haploid=haploid, verbose=True, bf_size=1 << 33, strip_chr=True)
    idx = pipeline.index(prefix + ".fa", prefix + ".vcf", opt)
    kmers = [(l.split()[0].encode(), int(l.split()[1])) for l in open(table + ".txt")]
    want = pipeline.call(prefix + ".fa", prefix + ".vcf", idx, kmers, opt)
    args = ["-b", "1", "-p", "-v"] + (["-1"] if haploid else []) + [prefix + ".fa", prefix + ".vcf", table]
    outs = []
    for env in ({}, {"MALVA_GENO_CUT_BATCH": "7"}, {"MALVA_GENO_HOST_CUT": "1"}):
        run_cli(["index"] + args, env=dict(os.environ, **env))
        outs.append(run_cli(["call"] + args, env=dict(os.environ, **env)))
    assert outs[0] == want and outs[1] == want and outs[2] == want
    assert want.count("\n") > 300
    r = subprocess.run([BIN, "call"] + args, capture_output=True, text=True, timeout=900, env=dict(os.environ, MALVA_GENO_CUT_BATCH="7"))
    assert r.returncode == 0 and " block(s) cut on the device in " in r.stderr and r.stdout == want
    n_batches = int(r.stderr.split(" block(s) cut on the device in ")[1].split()[0])
    assert n_batches > 40                                       # batches of 7 kept records
    r = subprocess.run([BIN, "call"] + args, capture_output=True, text=True, timeout=900, env=dict(os.environ, MALVA_GENO_HOST_CUT="1"))
    assert r.returncode == 0 and "cut on the device" not in r.stderr
    # ... and `index` enumerates on the device too: lone short variants through mg_index_isolated, the rest through mg_index_blocks
    r = subprocess.run([BIN, "index"] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0
    lone = r.stderr.split(" lone variant(s): ")[1].split()
    assert int(lone[0]) > 20 and int(lone[5]) <= 3                # "<n> indexed on the device, <m> on the host" (N / IUPAC windows only)
    assert " general block(s): " in r.stderr


def test_device_capacity_overflow_falls_back_to_host_enumerator(tmp_path):
    """a run of 18 SNPs at consecutive positions with UNPHASED genotypes gives chains of up to 18 unphased members: 2^18
    haplotype mixes per sample, beyond the device kernel's 2^14 -- those blocks must come back flagged and be redone by the
    host enumerator, with the same output as the oracle; the phased runs (chains of up to 35 members) stay on the device;
    forcing the host enumerator for everything (MALVA_GENO_HOST_ENUM) must not change a byte either"""
    import numpy as np
    rng = np.random.default_rng(5)
    seq = "".join(rng.choice(list("ACGT"), size=3000))
    with open(tmp_path / "d.fa", "w") as fh:
        fh.write(">1\n" + seq + "\n")
    lines = ["##fileformat=VCFv4.2", '##INFO=<ID=AF,Number=A,Type=Float,Description="af">', '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
             "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tS0\tS1\tS2"]
    recs = []
    for start, n, sep in ((500, 18, "/"), (1500, 6, "|"), (2200, 30, "|")):
        for p in range(start, start + n):
            ref = seq[p]
            alt = "ACGT"[("ACGT".index(ref) + 1 + p % 3) % 4]
            gts = ["%d%s%d" % (rng.integers(0, 2), sep, rng.integers(0, 2)) for _ in range(3)]
            lines.append("1\t%d\t.\t%s\t%s\t.\t.\tAF=0.%d\tGT\t%s" % (p + 1, ref, alt, 1 + p % 8, "\t".join(gts)))
            recs.append(("1", p, ref, [alt]))
    vcf = str(tmp_path / "d.vcf")
    open(vcf, "w").write("\n".join(lines) + "\n")
    fa = str(tmp_path / "d.fa")
    table = str(tmp_path / "d.kmers")
    vcf_synth.donor_table({"1": seq}, recs, 43, 6, table + ".txt")
    opt = pipeline.Options(verbose=True, bf_size=1 << 33)
    idx = pipeline.index(fa, vcf, opt)
    kmers = [(l.split()[0].encode(), int(l.split()[1])) for l in open(table + ".txt")]
    want = pipeline.call(fa, vcf, idx, kmers, opt)
    args = ["-b", "1", "-v", fa, vcf, table]
    run_cli(["index"] + args)
    r = subprocess.run([BIN, "call"] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout == want
    assert "1 block(s) enumerated on the host" in r.stderr          # the unphased run, and only it
    forced = run_cli(["call"] + args, env=dict(os.environ, MALVA_GENO_HOST_ENUM="1"))
    assert forced == want


def test_index_file_is_the_references_container_and_interchanges_with_the_oracle(tmp_path):
    """`malva-geno index` writes <vcf>.c43.k35.malvax.zst in the reference's layout (main.cpp:406-412);
    the oracle's independent reader must find in it exactly the index the oracle pipeline builds itself, and an index
    file WRITTEN by the oracle must drive `malva-geno call` to the same bytes.  (Format unpinned: oracle/index_file.py.)"""
    import numpy as np
    from oracle import index_file
    prefix = str(tmp_path / "case")
    contigs, records = vcf_synth.make_case(prefix, 41, haploid=False, k=35, n_clusters=80, vcf_strip_chr=True)
    table = str(tmp_path / "donor.kmers")
    vcf_synth.donor_table(contigs, records, 43, 41, table + ".txt")
    args = ["-k", "35", "-r", "43", "-b", "1", "-p", "-v", prefix + ".fa", prefix + ".vcf", table]
    run_cli(["index"] + args)
    zst, hipz = prefix + ".vcf.c43.k35.malvax.zst", prefix + ".vcf.c43.k35.malvax.hipz"
    assert os.path.exists(zst) and os.path.exists(hipz)                  # both by default: the reference's container and the sparse one
    assert not [f for f in os.listdir(str(tmp_path)) if ".tmp." in f]
    opt = pipeline.Options(haploid=False, verbose=True, k=35, ref_k=43, bf_size=1 << 33, strip_chr=True)
    idx = pipeline.index(prefix + ".fa", prefix + ".vcf", opt)
    filters, keys = index_file.read_index(zst)
    for (mode, size, pos, counts), want in zip(filters, (idx.context_bf, idx.bf)):
        assert (mode, size) == (1, 1 << 33)
        assert np.array_equal(pos, want.set_positions())
        assert counts.size == pos.size and not counts.any()             # switch_mode leaves zeroed counters (bloom_filter.hpp:96)
    assert filters[1][2].size > 100                                      # (context_bf may well be empty on a panel this small)
    assert keys == dict(idx.ref_bf.items()) and len(keys) > 100
    from_own = run_cli(["call"] + args)                                  # (reads the sparse container: the newer of the two)
    os.remove(hipz)
    assert run_cli(["call"] + args) == from_own                          # the reference's container alone
    # one container on request, and nothing stale of the other kind left beside it
    run_cli(["index"] + args, env=dict(os.environ, MALVA_GENO_INDEX_FORMAT="hipz"))
    assert os.path.exists(hipz) and not os.path.exists(zst)
    assert run_cli(["call"] + args) == from_own
    run_cli(["index"] + args, env=dict(os.environ, MALVA_GENO_INDEX_FORMAT="zst"))
    assert os.path.exists(zst) and not os.path.exists(hipz)
    # an index written by the oracle (as the reference binary would have) beside an OLDER sparse file of some other
    # index: the newer file is the one read
    other = str(tmp_path / "other")
    vcf_synth.make_case(other, 43, haploid=False, k=35, n_clusters=20, vcf_strip_chr=True)
    run_cli(["index", "-k", "35", "-r", "43", "-b", "1", "-p", other + ".fa", other + ".vcf", table], env=dict(os.environ, MALVA_GENO_INDEX_FORMAT="hipz"))
    shutil.copy(other + ".vcf.c43.k35.malvax.hipz", hipz)
    os.utime(hipz, (1, 1))
    index_file.write_index(zst, idx.context_bf, idx.bf, idx.ref_bf)
    assert run_cli(["call"] + args) == from_own
    # ... and the same stale sparse file with a FRESH date (copied or restored without its times): its header names the size and
    # modification time of the VCF it was built from, which is not this one -- the reference's container is read
    now = time.time() + 5
    os.utime(hipz, (now, now))
    r = subprocess.run([BIN, "call"] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout == from_own and "another version of" in r.stderr
    assert sum(1 for l in from_own.split("\n") if l and not l.startswith("#") and not l.endswith(":0")) > 20


def test_bcf_panel_gives_the_same_records_as_the_vcf(tmp_path, golden_dir):
    """a .bcf panel (what htslib's bcf_open reads for the reference, main.cpp:261-272) through index + call:
    the records equal those of the text panel byte for byte (the header differs by what a BCF must declare: contigs, PASS)"""
    import bcf_writer
    fa = os.path.join(golden_dir, "haploid.fa")
    vcf = str(tmp_path / "haploid.vcf.gz")
    shutil.copy(os.path.join(golden_dir, "haploid.vcf.gz"), vcf)
    bcf = str(tmp_path / "haploid.bcf")
    bcf_writer.vcf_to_bcf(vcf, bcf, with_idx=True)
    prefix = str(tmp_path / "haploid_malva43.kmercount")
    with open(prefix + ".txt", "w") as fh:
        for km, c in kmc_standin.count_fastq(os.path.join(golden_dir, "haploid.fq"), 43):
            fh.write("%s\t%d\n" % (km.decode(), c))
    opts = ["-1", "-k", "35", "-r", "43", "-b", "1", "-f", "AF", fa]
    run_cli(["index"] + opts + [bcf, prefix])
    out = run_cli(["call"] + opts + [bcf, prefix])
    recs = lambda s: [l for l in s.split("\n") if l and not l.startswith("##")]
    want = open(os.path.join(golden_dir, "haploid.malva.vcf")).read()        # the reference's own golden
    assert recs(out) == recs(want) and len(recs(out)) == 419


@pytest.mark.parametrize("what", ["samples", "uniform", "error_cov", "freq_key", "all"])
def test_cli_options_match_oracle(tmp_path, what):
    """The options of argument_parser.hpp:86-159 no other test passes, each against oracle/pipeline.py byte for byte:
    -s <file>  a subset of the panel's samples, kept in VCF order whatever the file's order (main.cpp:264-272, 434-440: the
               index AND the call see only those samples' genotypes, so other alleles carry signatures);
    -u         uniform allele frequencies (variant.hpp:148-152; no record is then 'not present' through its frequencies);
    -e / -c    error rate and the coverage cap of VB::genotype (var_block.hpp:236-248: over-covered alleles -> 0/0:0);
    -f EUR_AF  another INFO key than AF (BASELINE config C2's flag) on a panel that carries both with different values."""
    seed, k, ref_k = 91, 35, 43
    prefix = str(tmp_path / "case")
    contigs, records = vcf_synth.make_case(prefix, seed, haploid=False, k=k, n_clusters=60, vcf_strip_chr=True, n_samples=7, second_freq_key="EUR_AF")
    table = str(tmp_path / "donor.kmers")
    vcf_synth.donor_table(contigs, records, ref_k, seed, table + ".txt")
    opt = pipeline.Options(haploid=False, verbose=True, k=k, ref_k=ref_k, bf_size=1 << 33, strip_chr=True)
    args = ["-k", str(k), "-r", str(ref_k), "-b", "1", "-p", "-v"]
    if what in ("samples", "all"):
        sfile = str(tmp_path / "keep.txt")
        with open(sfile, "w") as fh:
            fh.write("S5\nS1\n\nS2\n")                                # not the VCF's order, a blank line
        opt.samples = sfile
        args += ["-s", sfile]
    if what in ("uniform", "all"):
        opt.uniform = True
        args += ["-u"]
    if what in ("error_cov", "all"):
        opt.error_rate, opt.max_coverage = 0.01, 8
        args += ["-e", "0.01", "-c", "8"]
    if what in ("freq_key", "all"):
        opt.freq_key = "EUR_AF"
        args += ["-f", "EUR_AF"]
    idx = pipeline.index(prefix + ".fa", prefix + ".vcf", opt)
    kmers = [(l.split()[0].encode(), int(l.split()[1])) for l in open(table + ".txt")]
    want = pipeline.call(prefix + ".fa", prefix + ".vcf", idx, kmers, opt)
    args += [prefix + ".fa", prefix + ".vcf", table]
    run_cli(["index"] + args)
    got = run_cli(["call"] + args)
    strip = lambda s: re.sub(r";GTS=[^\t]*", "", s)         # (the likelihood list only: COVS before it and GT:GQ behind it stay)
    assert strip(got) == strip(want)                 # header, records, COVS, GT, GQ: identical
    for a, b in zip(got.split("\n"), want.split("\n")):   # GTS: printf("%f") of doubles, within the contract's 1e-6
        if a != b:
            fa = [float(x.split(":")[1]) for x in a.split("GTS=")[1].split("\t")[0].split(",")]
            fb = [float(x.split(":")[1]) for x in b.split("GTS=")[1].split("\t")[0].split(",")]
            assert all(abs(x - y) <= 1.000001e-6 or (x != x and y != y) for x, y in zip(fa, fb)), (a, b)
    recs = [l for l in got.split("\n") if l and not l.startswith("#")]
    assert sum(1 for l in recs if not l.endswith(":0")) > 20
    if what == "error_cov":
        assert any(l.endswith("\t0/0:0") and "COVS=" in l and max(int(x) for x in l.split("COVS=")[1].split(";")[0].split(",")) > 8 for l in recs), "no record exercised the coverage cap"
    # and the option really changes the result (the test would pass vacuously if the flag were ignored by both sides)
    base_opt = pipeline.Options(haploid=False, verbose=True, k=k, ref_k=ref_k, bf_size=1 << 33, strip_chr=True)
    base = pipeline.call(prefix + ".fa", prefix + ".vcf", pipeline.index(prefix + ".fa", prefix + ".vcf", base_opt), kmers, base_opt)
    assert strip(base) != strip(want)


@pytest.mark.parametrize("seed,haploid,dense,subset", [(21, False, False, False), (22, True, False, False), (23, False, True, True), (24, True, True, True)])
def test_sample_columns_decoded_on_the_device_give_the_same_records(tmp_path, seed, haploid, dense, subset):
    """VcfReader::defer_genotypes + mg_decode_gt_text (what a panel of >= 1024 samples gets by itself; forced here on a
    small one): index and call leave the sample columns to the device -- lone records take their presence mask from the
    raw-allele mask, general blocks their sparse entries, blocks handed back to the host enumerator rebuild the pairs from
    the entries -- and must write the bytes of the run in which the host decoded every column, which in turn are the oracle's."""
    prefix = str(tmp_path / "case")
    contigs, records = vcf_synth.make_case(prefix, seed, haploid=haploid, k=35, n_clusters=70, n_samples=9, vcf_strip_chr=True, dense=dense)
    table = str(tmp_path / "donor.kmers")
    vcf_synth.donor_table(contigs, records, 43, seed, table + ".txt")
    args = ["-k", "35", "-r", "43", "-b", "1", "-p", "-v"] + (["-1"] if haploid else [])
    samples = "-"
    if subset:
        samples = str(tmp_path / "keep.txt")
        open(samples, "w").write("S7\nS1\nS4\nS2\n")
        args += ["-s", samples]
    args += [prefix + ".fa", prefix + ".vcf", table]
    host = dict(os.environ, MALVA_GENO_GT_DEVICE="0", MALVA_GENO_VCF_POOL="1")
    dev = dict(os.environ, MALVA_GENO_GT_DEVICE="1", MALVA_GENO_VCF_POOL="1", MALVA_GENO_CUT_BATCH="37")   # (batch seams inside blocks of text)
    run_cli(["index"] + args, env=host)
    want = run_cli(["call"] + args, env=host)
    import numpy as np
    from oracle import index_file
    filt_host, keys_host = index_file.read_index(prefix + ".vcf.c43.k35.malvax.zst")
    run_cli(["index"] + args, env=dev)
    filt_dev, keys_dev = index_file.read_index(prefix + ".vcf.c43.k35.malvax.zst")
    assert keys_dev == keys_host and len(keys_dev) > 50                                 # the same index: keys, and both filters' set bits
    for a, b in zip(filt_dev, filt_host):
        assert a[:2] == b[:2] and np.array_equal(a[2], b[2])
    got = run_cli(["call"] + args, env=dev)
    assert got == want
    assert sum(1 for l in got.split("\n") if l and not l.startswith("#") and not l.endswith(":0")) > 20
    # blocks the device hands back (forced: all of them) rebuild the genotype pairs from the entries
    forced = run_cli(["call"] + args, env=dict(dev, MALVA_GENO_HOST_ENUM="1"))
    assert forced == want
    opt = pipeline.Options(haploid=haploid, verbose=True, k=35, ref_k=43, bf_size=1 << 33, strip_chr=True, samples=samples)
    idx = pipeline.index(prefix + ".fa", prefix + ".vcf", opt)
    kmers = [(l.split()[0].encode(), int(l.split()[1])) for l in open(table + ".txt")]
    strip = lambda s: re.sub(r";GTS=[^\t]*", "", s)
    assert strip(got) == strip(pipeline.call(prefix + ".fa", prefix + ".vcf", idx, kmers, opt))


@pytest.mark.parametrize("k,ref_k,haploid", [(71, 79, False), (65, 65, True)])
def test_k_above_64_goes_through_the_bytewise_forms(tmp_path, k, ref_k, haploid):
    """the reference takes any k (argument_parser.hpp:57-58); the packed kernels hold k-mers of up to 64 bases, so beyond that
    `index` and `call` use the host enumerator and the ASCII batch forms of every store call -- same records as the oracle"""
    prefix = str(tmp_path / "case")
    contigs, records = vcf_synth.make_case(prefix, 90 + k, haploid=haploid, k=k, n_clusters=40, vcf_strip_chr=True)
    table = str(tmp_path / "donor.kmers")
    vcf_synth.donor_table(contigs, records, ref_k, 90 + k, table + ".txt")
    opt = pipeline.Options(haploid=haploid, verbose=True, k=k, ref_k=ref_k, bf_size=1 << 33, strip_chr=True)
    idx = pipeline.index(prefix + ".fa", prefix + ".vcf", opt)
    kmers = [(l.split()[0].encode(), int(l.split()[1])) for l in open(table + ".txt")]
    want = pipeline.call(prefix + ".fa", prefix + ".vcf", idx, kmers, opt)
    args = ["-k", str(k), "-r", str(ref_k), "-b", "1", "-p", "-v"] + (["-1"] if haploid else []) + [prefix + ".fa", prefix + ".vcf", table]
    run_cli(["index"] + args)
    got = run_cli(["call"] + args)
    strip = lambda s: re.sub(r";GTS=[^\t]*", "", s)
    assert strip(got) == strip(want)
    assert sum(1 for l in got.split("\n") if l and not l.startswith("#") and not l.endswith(":0")) > 10
