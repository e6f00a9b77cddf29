"""The record loop on a RESIDENT panel (mg_cut_blocks_device -> mg_cover_blocks_device -> mg_genotype_device, and
mg_index_blocks_device) against the C oracle's block path (oracle/malva_oracle.c: mo_cut_blocks, mo_index_blocks,
mo_cover_blocks, mo_genotype_panel -- itself pinned to the Python model in tests/test_oracle_blocks_cpu.py), on the two
recipes BASELINE.json names for the general-block path: config C4's clustered SNP panel as SURVEY 8(d) draws it (38 nt mean
spacing, a tenth of the records in clusters of <= 4 within 17 nt, positions beyond 2^25 where are_near runs in float) and
config C5's indel / MNP clusters (k35 r63, haploid and diploid).  Every array stays in HBM between the calls; bits, keys,
counters, cuts, coverages, GT and GQ must equal the oracle's, likelihoods within 1e-6 (they are bit-equal)."""
import numpy as np
import pytest

from gpu_util import map_values_by_key
from malva_amd import BF_ALT, BF_CTX, Context, synth
from malva_amd.resident import ResidentPanel
from oracle import capi as ocapi

pytestmark = pytest.mark.gpu
TOL = 1e-6        # north_star's tolerance on normalised likelihoods


def oracle_blocks(panel, k):
    off, bc = ocapi.cut_blocks(panel.pos, panel.ref_size, panel.min_size, panel.contig_id, k)
    return dict(blk_ref_base=panel.contig_base[bc], blk_ref_len=panel.contig_len[bc], blk_var_off=off, pos=panel.pos, ref_size=panel.ref_size,
                min_size=panel.min_size, present=panel.present, var_allele_off=panel.var_allele_off, allele_off=panel.allele_off, pool=panel.pool,
                canon=panel.canon, gt=panel.gt, n_samples=panel.n_samples)


def run_recipe(panel, k, ref_k, haploid, bits, n_rows, plant, min_general, sparse=False, options=()):
    args = oracle_blocks(panel, k)
    sizes = np.diff(args["blk_var_off"].astype(np.int64))
    assert (sizes > 1).sum() >= min_general, "the recipe drew too few general blocks: %d" % (sizes > 1).sum()
    # ---- index: the oracle's, then the device's from the resident panel ----
    obf, octx, omap = ocapi.BF(bits), ocapi.BF(bits), ocapi.KMAP()
    ocapi.index_blocks(obf, omap, panel.genome, **args, haploid=haploid, k=k)
    obf.switch_mode()
    for b, l in zip(panel.contig_base, panel.contig_len):
        ocapi.ref_scan(obf, octx, panel.genome[int(b):int(b) + int(l)].tobytes(), k, ref_k)
    octx.switch_mode()
    with Context(k, ref_k, bits) as ctx:
        ctx.set_option("blocks_round_log2", 14)             # several rounds of tier 2 (and a round seam inside the general list) on panels this size
        ctx.set_option("use_record_counters", 2)            # the tiers' lookups read the records' own counter copies, as they do by themselves at whole-genome size
        for name, value in options:
            ctx.set_option(name, value)
        ctx.reference_upload(panel.genome)
        rp = ResidentPanel(panel, 0, haploid=haploid, sparse=sparse)
        ovf = rp.index(ctx)
        assert ovf.sum() == 0, "%d records handed back at index time" % int(ovf.sum())
        ctx.bf_finalize(BF_ALT)
        for b, l in zip(panel.contig_base, panel.contig_len):
            ctx.ref_scan_resident(int(b), int(l))         # main.cpp:383-401 on the reference already in HBM
        ctx.bf_finalize(BF_CTX)
        assert np.array_equal(ctx.bf_export_sparse(BF_ALT)[2], obf.set_positions())
        assert np.array_equal(ctx.bf_export_sparse(BF_CTX)[2], octx.set_positions())
        keys, _ = ctx.map_export()
        assert sorted(keys) == sorted(k_ for k_, _ in omap.items())
        # ---- scan: the donor's ref_k-mers around the first `plant` records + random rows ----
        hi, lo, cnt = synth.flat_kmer_table(panel, n_rows, k, ref_k, seed=7, max_records=plant)
        ocapi.kmc_scan_packed(octx, obf, omap, hi, lo, cnt, k, ref_k)
        ctx.kmc_scan(hi, lo, cnt)
        assert np.array_equal(ctx.bf_export(BF_ALT)[3], obf.counts())
        assert map_values_by_key(ctx) == dict(omap.items())
        # ---- call: cut, cover, genotype, everything resident ----
        assert ctx.get_option("record_counters_live") == 1
        rp.call_step(ctx)
        got = rp.results()
        assert np.array_equal(got["blk_var_off"], args["blk_var_off"])
        assert got["overflow"].sum() == 0, "%d records handed back at call time" % int(got["overflow"].sum())
        stats = {}
        want_cov = ocapi.cover_blocks(obf, omap, panel.genome, **args, haploid=haploid, k=k, stats=stats)
        assert np.array_equal(got["cov"], want_cov)
        assert (want_cov > 0).sum() > plant // 2
        g1, g2, gq = ocapi.genotype_panel(want_cov, panel.freq, panel.var_allele_off, 0.001, 200, haploid)
        assert np.array_equal(got["g1"], g1) and np.array_equal(got["g2"], g2) and np.array_equal(got["gq"], gq)
        assert ((g1 > 0) | (g2 > 0)).sum() > plant // 20
        # likelihood lists of a sample of records, through the oracle's per-variant form
        rng = np.random.default_rng(1)
        goff = rp.t["gt_off"].cpu().numpy().view(np.uint64)
        worst = 0.0
        for v in rng.choice(panel.n, size=min(panel.n, 3000), replace=False):
            a0, a1 = int(panel.var_allele_off[v]), int(panel.var_allele_off[v + 1])
            gts = ocapi.genotype(want_cov[a0:a1], panel.freq[a0:a1], 0.001, 200, haploid)
            if got["status"][v] != 0:
                continue
            _, _, norm = ocapi.select_gt([g[2] for g in gts])
            mine = got["probs"][int(goff[v]):int(goff[v + 1])]
            assert len(mine) == len(norm)
            both = ~(np.isnan(norm) & np.isnan(mine))
            worst = max(worst, float(np.max(np.abs(norm[both] - mine[both]), initial=0.0)))
        assert worst <= TOL
        # the same call step again: the resident arrays were not consumed
        rp.call_step(ctx)
        again = rp.results()
        assert np.array_equal(again["cov"], got["cov"]) and np.array_equal(again["gq"], got["gq"])
        stats["device_general_kmers"] = ctx.blocks_stats()[5]
    return stats


def test_c4_recipe_clustered_snps_beyond_2_to_the_25():
    """~38 nt mean spacing, 10 % of the SNPs in clusters of <= 4 within 17 nt; ONE sequence so that most
    positions lie beyond 2^24 (float are_near) -- at 1.2e6 records the last sit near 4.4e7."""
    k, ref_k = 35, 43
    panel = synth.clustered_snp_panel(1_200_000, seed=41, n_contigs=1)
    assert panel.pos.max() > (1 << 25)
    stats = run_recipe(panel, k, ref_k, False, 1 << 30, n_rows=3_000_000, plant=20_000, min_general=30_000)
    assert stats["signatures"] > 2 * panel.n


def test_index_on_the_device_is_repeatable():
    """`index` runs the tiers twice (count the exact map's rows, then insert): both passes must deal every record to the same
    tier whatever order the waves run in.  Five indexes of the same C5 panel on fresh contexts: same bits, same keys."""
    panel = synth.indel_panel(20_000, seed=77)
    k, bits = 35, 1 << 26
    args = oracle_blocks(panel, k)
    obf, omap = ocapi.BF(bits), ocapi.KMAP()
    ocapi.index_blocks(obf, omap, panel.genome, **args, haploid=False, k=k)
    obf.switch_mode()
    want_bits, want_keys = obf.set_positions(), sorted(k_ for k_, _ in omap.items())
    for _ in range(5):
        with Context(k, 63, bits) as ctx:
            ctx.reference_upload(panel.genome)
            rp = ResidentPanel(panel, 0, haploid=False)
            assert rp.index(ctx).sum() == 0
            ctx.bf_finalize(BF_ALT)
            assert np.array_equal(ctx.bf_export_sparse(BF_ALT)[2], want_bits)
            assert sorted(ctx.map_export()[0]) == want_keys


@pytest.mark.parametrize("haploid", [False, True])
def test_c5_recipe_indel_mnp_clusters_r63(haploid):
    """clusters of <= 6 mixing SNPs, MNPs, deletions, insertions (some >= k), <= 3 ALTs, 8 samples half
    unphased; k35 r63"""
    panel = synth.indel_panel(60_000, seed=52 + int(haploid))
    run_recipe(panel, 35, 63, haploid, 1 << 28, n_rows=2_000_000, plant=15_000, min_general=30_000)


@pytest.mark.parametrize("haploid", [False, True])
def test_large_panel_sparse_genotypes_through_the_workgroup_tier(haploid):
    """700 samples (beyond the flat tier's 512: every general record takes the workgroup kernel), 97 % of the genotypes 0|0
    phased, handed over SPARSE (mg_panel_dev.sp_*: 3 % of the dense matrix): the lone tier gathers its presence masks from the
    entries, the workgroup kernel walks the entries of a chain's members plus the one all-reference sample.  Index, counters,
    cuts, coverages, GT and GQ equal the oracle's (which reads the dense matrix)."""
    panel = synth.indel_panel(4_000, seed=61 + int(haploid), n_samples=700, hom_ref=0.97, unphased_frac=0.02)
    from malva_amd.capi import sparse_genotypes
    off, _, _ = sparse_genotypes(panel.gt, panel.n_samples)
    assert off[-1] < 0.06 * panel.gt.size
    run_recipe(panel, 35, 63, haploid, 1 << 26, n_rows=400_000, plant=3_000, min_general=2_000, sparse=True)


@pytest.mark.parametrize("variant", ["diploid", "haploid", "unphased", "sparse", "eight-samples"])
def test_snp_chains_in_one_kernel_and_through_picks_and_eval(variant):
    """chains of SNPs take fw_snp_kernel (lanes = the panel's haplotypes along the chain) when every haplotype fits its fixed
    geometry, and the picks + eval kernels otherwise: the C4 recipe's clustered SNPs as they come (two phased samples), haploid,
    with a third of the genotypes unphased (chains of two or more members then need every mix: left to the picks kernel), with
    sparse genotypes, and with eight samples (sixteen lanes per chain).  Each against the oracle with the kernel on and off; both
    ways enumerate the same number of signature k-mers."""
    n_samples = 8 if variant == "eight-samples" else 2
    panel = synth.clustered_snp_panel(150_000, seed=61, n_contigs=3, n_samples=n_samples)
    if variant == "unphased":
        rng = np.random.default_rng(5)
        drop = rng.random(panel.gt.shape) < 0.33
        panel.gt[drop] &= np.uint16(~(1 << 14) & 0xFFFF)
    counts = []
    for on in (1, 0):
        st = run_recipe(panel, 35, 43, variant == "haploid", 1 << 28, n_rows=600_000, plant=8_000, min_general=3_000, sparse=variant == "sparse",
                        options=[("use_snp_kernel", on)])
        counts.append(st["device_general_kmers"])
    assert counts[0] == counts[1] > 10_000


@pytest.mark.parametrize("variant", ["eight-samples", "haploid", "two-samples", "forty-samples"])
def test_picks_evaluated_by_the_chain_kernel_and_through_items(variant):
    """tier 2's picks of a chain are evaluated by the wave that holds them (fw_chain_kernel: geometry staged in LDS, chains in
    order of their length) or written out as items for fw_eval_kernel (use_chain_kernel = 0), with and without the ordering
    pass.  C5's indel / MNP clusters on panels of 8 samples (eight chains per wave), 2 samples (thirty-two chains per wave: the
    staging area's share per chain is small, chains of many alleles take the item path) and 40 samples (a wave per chain).
    Each way against the oracle; all ways enumerate the same number of signature k-mers."""
    n_samples = {"two-samples": 2, "forty-samples": 40}.get(variant, 8)
    panel = synth.indel_panel(12_000, seed=83, n_samples=n_samples)
    counts = []
    for chain, order in ((1, 1), (1, 0), (0, 1)):
        st = run_recipe(panel, 35, 63, variant == "haploid", 1 << 27, n_rows=500_000, plant=5_000, min_general=5_000,
                        options=[("use_chain_kernel", chain), ("use_chain_order", order)])
        counts.append(st["device_general_kmers"])
    assert counts[0] == counts[1] == counts[2] > 50_000


@pytest.mark.parametrize("haploid", [False, True])
def test_chains_with_bases_outside_acgt_in_reach(haploid):
    """a base outside ACGT near or inside a cluster: fw_chain_kernel looks at every base a window of the chain can hold when it
    stages the chain, and a chain with such a base takes the item path, where -- as before -- the record goes on to the workgroup
    kernel if one of its windows holds the base (that kernel assembles byte by byte and leaves the k-mer out, as the reference
    does).  Index from the oracle with made-up weights; coverages of every record the device kept equal the oracle's, and
    coverages and flags are the same whichever way tier 2 runs."""
    k, ref_k, bits = 35, 63, 1 << 26
    panel = synth.indel_panel(12_000, seed=84)
    rng = np.random.default_rng(9)
    gpos = panel.gpos()
    for v in rng.choice(panel.n, size=1_500, replace=False):
        panel.genome[int(gpos[v]) + int(rng.integers(-30, 31))] = ord("N")
    args = oracle_blocks(panel, k)
    obf, omap = ocapi.BF(bits), ocapi.KMAP()
    ocapi.index_blocks(obf, omap, panel.genome, **args, haploid=haploid, k=k)
    obf.switch_mode()
    cnts = obf.counts()
    cnts[:] = (1 + (np.arange(cnts.size, dtype=np.uint64) * np.uint64(2654435761)) % np.uint64(97)).astype(np.uint16)
    acgt = set(b"ACGT")
    for key, _ in list(omap.items()):
        if set(key) <= acgt:       # (a KMC table holds 2-bit k-mers: no scan can count a key with another letter in it)
            omap.increment(key, 1 + ocapi.xxh3_64(key) % 97)
    want_cov = ocapi.cover_blocks(obf, omap, panel.genome, **args, haploid=haploid, k=k)
    items = [(k_, v_) for k_, v_ in omap.items() if set(k_) <= acgt]
    seen = []
    for chain, order in ((1, 1), (1, 0), (0, 1)):
        with Context(k, ref_k, bits) as ctx:
            ctx.set_option("use_chain_kernel", chain)
            ctx.set_option("use_chain_order", order)
            ctx.bf_import_sparse(BF_ALT, 1, bits, obf.set_positions(), obf.counts())
            ctx.bf_import_sparse(BF_CTX, 1, bits, np.zeros(0, np.uint64), np.zeros(0, np.uint16))
            ctx.map_import([k_ for k_, _ in items], np.array([v for _, v in items], dtype=np.int32))
            ctx.reference_upload(panel.genome)
            rp = ResidentPanel(panel, 0, haploid=haploid)
            rp.call_step(ctx)
            got = rp.results()
            general_kmers, tier3_records = ctx.blocks_stats()[5:7]
        kept = np.repeat(got["overflow"] == 0, np.diff(panel.var_allele_off.astype(np.int64)))
        assert (got["overflow"] != 0).mean() < 0.08 and 100 < tier3_records < 0.2 * panel.n
        assert np.array_equal(got["cov"][kept], want_cov[kept])
        assert (want_cov[kept] > 0).sum() > 20_000
        seen.append((got["overflow"].copy(), got["cov"].copy(), general_kmers))
    for other in seen[1:]:
        assert np.array_equal(other[0], seen[0][0]) and np.array_equal(other[1], seen[0][1])
