"""Every BASELINE.json config that fits one GPU, at its stated size, against the oracle.

  C1  SARS-CoV-2 panel (15,154 records x 27,934 samples), k35 r43 b1, haploid: `bin/malva-geno` vs the oracle pipeline's
      full output (tests/golden/sars_cov2.oracle.malva.verbose.vcf.gz, tests/gen_c1_golden.py), byte for byte
  C2  chr20: inputs absent from the reference checkout (.MISSING_LARGE_BLOBS) -- cannot be run by anyone here
  C3  1e8 k-mers x 1e6 biallelic SNPs, b=4: the oracle cannot scan 1e8 rows in seconds, so the full-size run is held by
      size-independent properties (linearity, shard sums, a hash-free checksum of what every signature's counter must
      hold) and the oracle scans a 5e6-row / 2e5-variant sample with filters IT built (mo_add_kmers + mo_ref_scan)
  C4  one GPU's share of the whole-genome config (3.75e8 rows, b=16) against 1e7- and 8e7-SNP indexes: the same
      properties, with the large-index scan forms asserted to be the ones exercised
  C5  indel/MNP-heavy clustered panel, k35 r63 b=8, haploid and diploid: CLI output vs the oracle pipeline, byte for byte

Pins: the haploid / hom-likelihood branch and the haploid pick logic are held by the reference's own golden
(tests/test_gpu_cli.py::test_haploid_example_byte_identical).  The het likelihood branch (var_block.hpp:300-316), the
diploid phased / unphased pick fan-out (:709-728) and the u16 / u32 wraps are held ONLY by SURVEY Appendix B's vectors and by
the oracle's restatement: no reference-held fixture reaches them (example/chr20.malva.vcf has no inputs).  C3, C4 and C5's
diploid halves run in diploid mode and are therefore oracle-pinned, not reference-pinned."""
import gzip
import os
import shutil

import numpy as np
import pytest
import torch

import vcf_synth
from big_cases import DeviceTable, build_device_index, counters_tensor
from malva_amd import BF_ALT, BF_CTX, Context, synth
from oracle import capi as ocapi
from oracle import kmc_standin, pipeline
from test_gpu_cli import run_cli

pytestmark = pytest.mark.gpu
K, R = 35, 43


def _u32(t):
    return t.cpu().numpy().view(np.uint32).astype(np.int64)


def _scan(ctx, tab, a=0, b=None):
    torch.cuda.synchronize()            # torch's reads of the aliased counters are done before the library writes them
    ctx.kmc_scan_device(*tab.ptrs(a, b))
    ctx.synchronize()


def _snap(t):
    c = t.clone()
    torch.cuda.synchronize()
    return c


def _check_expected(ctx, panel, tab, counters, n_bf, a=0, b=None, n_check=None):
    """the hash-free checksum: counter of variant j's REF key == summed counts of its planted REF windows; counter of
    its ALT signature's filter slot == those of its ALT windows (summed over the signatures sharing the slot) unless
    the reference itself holds the ALT context"""
    n_check = panel.n if n_check is None else n_check
    ref, alt = tab.expected_sums(panel, a, b)
    got = _u32(counters)
    assert np.array_equal(got[n_bf:n_bf + n_check], ref[:n_check] & 0xFFFFFFFF)        # ids follow insertion order: variant j -> id j
    assert ref[:n_check].sum() > 0
    # ALT: slot of every ALT signature (the library's byte-wise hash, a different kernel from the scan's packed one)
    h = K // 2
    w = synth.windows(panel.genome, panel.pos[:n_check] - h, K).copy()
    w[:, h] = panel.pool[1:2 * n_check:2]
    rows = np.zeros((n_check, 40), dtype=np.uint8)
    rows[:, :K] = w
    slot = ctx.bf_index(BF_ALT, rows)
    pos = ctx.bf_export_sparse(BF_ALT)[2]
    rank = np.searchsorted(pos, slot)
    assert np.array_equal(pos[rank], slot)                                               # every ALT signature is a set bit
    want = np.zeros(n_bf, dtype=np.int64)
    np.add.at(want, rank, alt[:n_check])
    touched = np.zeros(n_bf, dtype=bool)
    touched[rank] = True
    # (variants beyond n_check have no planted rows: a slot they share with a checked signature gets nothing from them)
    # Bloom false positives are semantics: a random table row whose centre k-mer merely COLLIDES with a set
    # bit of `bf` is counted by the reference, so about n_rows * n_check / bits of the checked counters carry one
    # extra row's count (2..63).  Everything else must be exact, and the extras must be there and look like that.
    n_rows = (tab.n if b is None else b) - a
    fp = n_rows * n_check / ctx.bf_bits
    # The other way round, a planted ALT window whose 43-mer COLLIDES with a set bit of context_bf is (as in the
    # reference, main.cpp:496) not counted: context_bf's fill times the planted windows, a few dozen at most.
    extra = ((got[:n_bf] - want) & 0xFFFF)[touched]
    missing = extra >= 0x8000
    ctx_fill = ctx.bf_info(BF_CTX)[1] / ctx.bf_bits
    assert int(missing.sum()) <= 3 * ctx_fill * n_check + 8, int(missing.sum())
    n_extra = int(((extra != 0) & ~missing).sum())
    assert 0.7 * fp - 30 <= n_extra <= 1.2 * fp + 30, (n_extra, fp)
    assert extra[~missing].max() < 64 * 4 and (extra[(extra != 0) & ~missing] >= 2).all()
    assert (want > 0).sum() > n_check // 8


def test_c3_full_size_1e8_rows_1e6_snps_b4():
    n_vars, n_rows, bits = 1_000_000, 100_000_000, 4 << 33
    panel = synth.snp_panel(n_vars, 20261003)
    tab = DeviceTable(panel, n_rows, K, R, 7)
    with Context(K, R, bits) as ctx:
        build_device_index(ctx, panel, K)
        counters, n_bf, n_map = counters_tensor(ctx)
        assert n_map == n_vars and n_bf > 0.99 * n_vars
        _scan(ctx, tab)
        whole = _snap(counters)
        assert ctx.get_option("scan_bins") == 0 and ctx.get_option("pregate_k") == 0     # C3: one L2-resident gate, direct form
        _check_expected(ctx, panel, tab, whole, n_bf)
        # linearity: a second pass doubles every counter (u32 wrap; the reference's u16 cells are the low halves)
        _scan(ctx, tab)
        assert torch.equal(counters, whole * 2)
        # shards: two halves scanned separately sum to the whole (what the all-reduce relies on)
        ctx.counters_reset(); _scan(ctx, tab, 0, n_rows // 2); first = _snap(counters)
        ctx.counters_reset(); _scan(ctx, tab, n_rows // 2, n_rows)
        assert torch.equal(first + counters, whole)
        # the same table as compact 12-byte rows (what bench.py's headline scans: scan_filter12_kernel): identical counters from the
        # whole table, linear, and its two halves (chunk starts are whole quads of rows) sum to the whole
        d_rows = torch.zeros(ctx.kmc_rows_bytes(n_rows) // 4, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.kmc_pack_rows_device(*tab.ptrs(), d_rows.data_ptr())

        def scan12(a, b):
            torch.cuda.synchronize()
            ctx.kmc_scan_rows_device(d_rows.data_ptr() + 12 * a, b - a)
            ctx.synchronize()
        ctx.counters_reset(); scan12(0, n_rows)
        assert torch.equal(counters, whole)
        _check_expected(ctx, panel, tab, counters, n_bf)
        scan12(0, n_rows)
        assert torch.equal(counters, whole * 2)
        ctx.counters_reset(); scan12(0, n_rows // 2); first = _snap(counters)
        ctx.counters_reset(); scan12(n_rows // 2, n_rows)
        assert torch.equal(first + counters, whole)
        del d_rows
        # oracle: its own index (built by mo_add_kmers + mo_ref_scan, nothing imported from the device), a 5e6-row sample
        # of the same table, 2e5 variants genotyped
        ns, nv = 5_000_000, 200_000
        obf, octx, omap = ocapi.BF(bits), ocapi.BF(bits), ocapi.KMAP()
        sig, _ = synth.snp_signature_rows(panel, K)
        rows = np.zeros((sig.shape[0], 40), dtype=np.uint8)
        rows[:, :K] = sig
        isr = np.zeros(rows.shape[0], dtype=np.uint8)
        isr[0::2] = 1
        ocapi.add_kmers(obf, omap, rows, isr)
        del rows, sig
        obf.switch_mode()
        ocapi.ref_scan(obf, octx, panel.genome.tobytes(), K, R)
        octx.switch_mode()
        assert np.array_equal(ctx.bf_export_sparse(BF_ALT)[2], obf.set_positions())       # index parity at full size
        assert np.array_equal(ctx.bf_export_sparse(BF_CTX)[2], octx.set_positions())
        hi, lo, cnt = tab.host(0, ns)
        ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, K, R)
        ctx.counters_reset(); _scan(ctx, tab, 0, ns); torch.cuda.synchronize()
        assert np.array_equal(ctx.bf_export(BF_ALT)[3], obf.counts())
        ovals = dict(omap.items())
        got = _u32(counters)[n_bf:]
        ch, cl = synth.canonical_m(*synth.pack_ascii(synth.snp_signature_rows(panel, K)[0][0::2]), K)
        keys = synth.unpack_ascii(ch, cl, K)[:, :K]
        assert all(int(got[j]) == (ovals[bytes(keys[j])] & 0xFFFFFFFF) for j in range(0, n_vars, 7))
        ctx.reference_upload(panel.genome)
        cov, g1, g2, gq, st = ctx.call_isolated(panel.pos[:nv].astype(np.uint64), panel.var_allele_off[:nv + 1], panel.allele_off[:2 * nv + 1],
                                                panel.pool[:2 * nv], panel.freq[:2 * nv], panel.present_mask[:nv], panel.flags[:nv], 0.001, 200, False)
        ocov, og1, og2, ogq = ocapi.call_isolated(obf, omap, panel.genome, panel.pos[:nv], panel.allele_off[:2 * nv + 1], panel.var_allele_off[:nv + 1],
                                                  panel.pool[:2 * nv], panel.freq[:2 * nv], panel.present_mask[:nv], panel.flags[:nv], K, 0.001, 200, False)
        assert np.array_equal(cov, ocov) and np.array_equal(g1, og1) and np.array_equal(g2, og2) and np.array_equal(gq, ogq)
        assert (g1 + g2 > 0).sum() > 1000


@pytest.mark.parametrize("n_vars,expect", [(10_000_000, "subs"), (10_000_000, "tickets"), (10_000_000, "two-level"), (80_000_000, "subs-compact"), (80_000_000, "saturated")])
def test_c4_one_gpu_share_3p75e8_rows_b16(n_vars, expect):
    """one GPU's eighth of config C4 (3e9 k-mers / 8) against the replicated index; 38-40 nt spacing as SURVEY 8(d).
    `subs` (tickets filed under LDS-sized pieces of the gate) is what the library picks by itself for gates of 32 MiB and more;
    the forms it replaced stay as options."""
    n_rows, bits, plant = 375_000_000, 16 << 33, 1_000_000
    panel = synth.snp_panel(n_vars, 4242, spacing=40)
    tab = DeviceTable(panel, n_rows, K, R, 9, plant_variants=plant)
    with Context(K, R, bits) as ctx:
        if not expect.startswith("subs"):
            ctx.set_option("use_sub", 0)
        if expect != "tickets":
            ctx.set_option("use_tickets", 0)
        if expect == "two-level":
            ctx.set_option("use_partition", 1)
        build_device_index(ctx, panel, K)
        counters, n_bf, n_map = counters_tensor(ctx)
        assert n_map == n_vars
        d_rows = None
        if expect == "subs-compact":  # the table resident as 12-byte rows: what bench.py --strong scans at this size
            d_rows = torch.zeros(ctx.kmc_rows_bytes(n_rows) // 4, dtype=torch.int32, device="cuda:0")
            torch.cuda.synchronize()
            ctx.kmc_pack_rows_device(*tab.ptrs(), d_rows.data_ptr())

        def scan(a=0, b=None):
            if d_rows is None:
                return _scan(ctx, tab, a, b)
            b = n_rows if b is None else b
            assert a % 4 == 0                                                             # packed rows start on whole quads
            torch.cuda.synchronize()
            ctx.kmc_scan_rows_device(d_rows[3 * a:].data_ptr(), b - a)
            ctx.synchronize()
        scan()
        whole = _snap(counters)
        if expect == "two-level":        # 32 MiB fine gate -> 4 MiB coarse gate in front, survivors partitioned by fine-gate slice
            assert ctx.get_option("pregate_k") >= 1 and ctx.get_option("gate_log2") == 28 and ctx.get_option("scan_bins") == 16
        elif expect == "tickets":        # one 8-byte ticket per row filed under its 2 MiB gate slice
            assert ctx.get_option("gate_log2") == 28 and ctx.get_option("scan_tickets") == 16 and ctx.get_option("scan_bins") == 0 and ctx.get_option("scan_spilled") == 0
        elif expect == "subs":           # ... or under its 128 KiB piece of the gate, which is then answered out of LDS
            assert ctx.get_option("gate_log2") == 28 and ctx.get_option("scan_subs") == 256 and ctx.get_option("scan_tickets") == 0 and ctx.get_option("scan_spilled") == 0
        elif expect == "subs-compact":   # the gate of a whole-genome index is held to 1,024 pieces: 128 MiB
            assert ctx.get_option("gate_log2") == 30 and ctx.get_option("scan_subs") == 1024 and ctx.get_option("scan_spilled") == 0
        else:                            # 1.6e8 entries saturate a 4 MiB coarse gate: decided at finalize, scans go straight to the 256 MiB gate
            assert ctx.get_option("pregate_k") == 0 and ctx.get_option("gate_log2") == 31 and ctx.get_option("scan_bins") == 0 and ctx.get_option("scan_tickets") == 0
        _check_expected(ctx, panel, tab, whole, n_bf, n_check=plant)
        scan()
        assert torch.equal(counters, whole * 2)                                           # linearity
        ctx.counters_reset(); scan(0, 150_000_000); first = _snap(counters)
        ctx.counters_reset(); scan(150_000_000, n_rows)
        assert torch.equal(first + counters, whole)                                       # shard sum (uneven shards, chunk seams inside both)


def test_c1_sars_cov2_full_output_equals_the_oracle(tmp_path, golden_dir):
    want_path = os.path.join(golden_dir, "sars_cov2.oracle.malva.verbose.vcf.gz")
    if not os.path.exists(want_path):
        pytest.skip("tests/golden/sars_cov2.oracle.malva.verbose.vcf.gz not generated yet (tests/gen_c1_golden.py -v: hours of pure Python); "
                    "test_gpu_cli.py::test_sars_cov2_panel_config_c1 holds the record count and the reference's two calls meanwhile")
    want = gzip.open(want_path, "rt").read()
    fa = os.path.join(golden_dir, "reference_sarsCov2.fasta")
    vcf = str(tmp_path / "sars_cov2.vcf.gz")
    shutil.copy(os.path.join(golden_dir, "sars_cov2.vcf.gz"), vcf)
    prefix = str(tmp_path / "sample.kmercount")
    with open(prefix + ".txt", "w") as fh:
        for km, c in kmc_standin.count_fastq(os.path.join(golden_dir, "haploid.fq"), 43):
            fh.write("%s\t%d\n" % (km.decode(), c))
    common = ["-1", "-v", "-k", "35", "-r", "43", "-b", "1", "-f", "AF", fa, vcf, prefix]
    run_cli(["index"] + common)
    got = run_cli(["call"] + common)
    assert got.count("\n") == want.count("\n") == 15154 + got.split("#CHROM")[0].count("\n") + 1
    strip = lambda s: "\n".join(";".join(p for p in l.split(";") if not p.startswith("GTS=")) if "GTS=" in l else l for l in s.split("\n"))
    assert strip(got) == strip(want)                # header, 15,154 records: COVS, GT, GQ identical
    for a, b in zip(got.split("\n"), want.split("\n")):
        if a != b:                                  # GTS: printf("%f") of doubles that may differ in the last bit of exp()
            fa_ = [float(x.split(":")[1]) for x in a.split("GTS=")[1].split("\t")[0].split(",")]
            fb_ = [float(x.split(":")[1]) for x in b.split("GTS=")[1].split("\t")[0].split(",")]
            assert all(abs(x - y) <= 1.000001e-6 or (x != x and y != y) for x, y in zip(fa_, fb_)), (a, b)
    recs = [l.split("\t") for l in got.split("\n") if l and not l.startswith("#")]
    assert [(r[1], r[3], r[4], r[9]) for r in recs if not r[9].startswith("0:")] == [("17747", "C", "T", "1:94"), ("17858", "A", "G", "1:100")]
    plain = run_cli(["call"] + [c for c in common if c != "-v"])     # and the default (non-verbose) form: INFO is "."
    precs = [l.split("\t") for l in plain.split("\n") if l and not l.startswith("#")]
    assert [r[:7] + r[8:] for r in precs] == [r[:7] + r[8:] for r in recs] and all(r[7] == "." for r in precs)


@pytest.mark.parametrize("haploid", [True, False])
def test_c5_indel_mnp_panel_r63_b8(tmp_path, haploid):
    seed = 51 if haploid else 52
    prefix = str(tmp_path / "c5")
    contigs, records = vcf_synth.make_case(prefix, seed, haploid=haploid, k=35, n_clusters=700, vcf_strip_chr=True)
    table = str(tmp_path / "donor.kmers")
    vcf_synth.donor_table(contigs, records, 63, seed, table + ".txt")
    opt = pipeline.Options(haploid=haploid, verbose=True, k=35, ref_k=63, bf_size=8 << 33, strip_chr=True)
    idx = pipeline.index(prefix + ".fa", prefix + ".vcf", opt)
    kmers = [(l.split()[0].encode(), int(l.split()[1])) for l in open(table + ".txt")]
    want = pipeline.call(prefix + ".fa", prefix + ".vcf", idx, kmers, opt)
    args = ["-k", "35", "-r", "63", "-b", "8", "-p", "-v"] + (["-1"] if haploid else []) + [prefix + ".fa", prefix + ".vcf", table]
    # the compact index container here: the reference's holds two 8-GiB bit vectors, whose zstd passes are exercised at b=1
    env = dict(os.environ, MALVA_GENO_INDEX_FORMAT="hipz")
    run_cli(["index"] + args, env=env)
    got = run_cli(["call"] + args, env=env)
    assert got == want and got.count("\n") > 1500
    assert sum(1 for l in got.split("\n") if l and not l.startswith("#") and not l.endswith(":0")) > 200


def test_bench_one_gpu_line_replays_a_captured_step():
    """`python bench.py` on one GPU at a reduced C3: ONE JSON line; its timed steps are replays of one step captured into a HIP graph
    (after the warm-up steps a step allocates nothing and launches on one stream), the same steps launched kernel by kernel leave the
    same coverages, GT and GQ, the CPU oracle's sample agrees, and --no-graph gives the plain launches"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    common = [sys.executable, os.path.join(root, "bench.py"), "--kmers", "4e6", "--variants", "5e4", "--steps", "3", "--warmup", "2", "--no-c4-leg", "--no-c5-leg",
              "--sustained-s", "0", "--cpu-sample", "2e5", "--cpu-variants", "2e4"]
    for extra, launch in (([], "hipGraph"), (["--no-graph"], "stream")):
        r = subprocess.run(common + extra, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.split("\n") if l.strip()]
        assert len(lines) == 1, r.stdout[-2000:]
        d = json.loads(lines[0])
        assert d["launch"].startswith(launch), d["launch"]
        assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["overflow_records"] == 0
        assert all(v is True for k, v in d["parity_sample"].items() if k.endswith("_equal")), d["parity_sample"]
        if launch == "hipGraph":
            assert d["replay_equals_launched"] is True and d["ms_per_step_stream"] > 0
