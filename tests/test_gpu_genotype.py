"""V1, G1-G3 and the fused isolated-variant path on the device vs the oracle.
GT and GQ must be identical; likelihoods within 1e-6 (BASELINE north_star), and the
share that is bit-identical is reported."""
import numpy as np
import pytest

from gpu_util import build_index_pair, pad_rows
from malva_amd import BF_ALT, Context, synth
from malva_amd.capi import GT_NORMAL, GT_OVERCOV, GT_NOCOV
from oracle import capi as ocapi

pytestmark = pytest.mark.gpu
EPS = 0.001
TOL = 1e-6   # absolute, on normalised likelihoods


def _oracle_variant(cov, freq, haploid, max_cov=200):
    gts = ocapi.genotype(cov, freq, EPS, max_cov, haploid)
    bi, gq, norm = ocapi.select_gt([g[2] for g in gts])
    g1, g2 = (gts[bi][0], gts[bi][1]) if bi >= 0 else (0, -1 if haploid else 0)
    return g1, g2, gq, norm, gts


def test_known_answer_vectors():
    """SURVEY Appendix B rows, through mg_genotype"""
    cases = [([12, 9], [0.7, 0.3], (0, 1, 100)), ([30, 0], [0.95, 0.05], (0, 0, 100)), ([0, 25], [0.95, 0.05], (1, 1, 100)),
             ([10, 7, 3], [0.6, 0.3, 0.1], (0, 1, 100)), ([201, 3], [0.5, 0.5], (0, 0, 0)), ([5, 5], [1.0, 0.0], (0, 0, 100)),
             ([15, 14, 0, 1], [0.25] * 4, (0, 1, 100)), ([0, 0], [0.5, 0.5], (0, 0, 0))]
    cov = np.concatenate([c for c, _, _ in cases]).astype(np.uint32)
    freq = np.concatenate([f for _, f, _ in cases]).astype(np.float32)
    off = np.cumsum([0] + [len(c) for c, _, _ in cases]).astype(np.uint32)
    ctx = Context(35, 43, 1 << 16)
    g1, g2, gq, st, probs, goff = ctx.genotype(cov, freq, off, EPS, 200, False, want_probs=True)
    for i, (_, _, want) in enumerate(cases):
        assert (int(g1[i]), int(g2[i]), int(gq[i])) == want
    assert list(st) == [0, 0, 0, 0, GT_OVERCOV, 0, 0, GT_NOCOV]
    # third case: GTS 0.000001 / 0.999999
    p = probs[int(goff[2]):int(goff[3])]
    assert "%f" % p[1] == "0.000001" and "%f" % p[2] == "0.999999"
    h1, h2, hq, _, _, _ = ctx.genotype(np.array([10, 7, 3], np.uint32), np.array([0.6, 0.3, 0.1], np.float32),
                                       np.array([0, 3], np.uint32), EPS, 200, True)
    assert (int(h1[0]), int(h2[0]), int(hq[0])) == (0, -1, 100)
    ctx.close()


@pytest.mark.parametrize("haploid", [False, True])
def test_random_coverages_against_oracle(haploid):
    rng = np.random.default_rng(42 + haploid)
    n = 20000
    A = rng.integers(2, 7, size=n)
    off = np.zeros(n + 1, dtype=np.uint32)
    off[1:] = np.cumsum(A)
    cov = np.zeros(off[-1], dtype=np.uint32)
    freq = np.zeros(off[-1], dtype=np.float32)
    for v in range(n):
        a = int(A[v])
        mode = rng.integers(0, 10)
        c = rng.integers(0, 60, size=a)
        if mode == 0:
            c[:] = 0
        elif mode == 1:
            c[rng.integers(0, a)] = 201 + rng.integers(0, 100)
        elif mode == 2:
            c = rng.integers(0, 201, size=a)
        elif mode >= 6:
            c[rng.integers(0, a):] = 0
        f = rng.dirichlet(np.ones(a)).astype(np.float32)
        if mode == 3:
            f[rng.integers(0, a)] = 0
        cov[off[v]:off[v + 1]] = c
        freq[off[v]:off[v + 1]] = f
    ctx = Context(35, 43, 1 << 16)
    g1, g2, gq, st, probs, goff = ctx.genotype(cov, freq, off, EPS, 200, haploid, want_probs=True)
    exact = total = 0
    for v in range(n):
        o1, o2, oq, norm, gts = _oracle_variant(cov[off[v]:off[v + 1]], freq[off[v]:off[v + 1]], haploid)
        assert (int(g1[v]), int(g2[v])) == (o1, o2), v
        assert int(gq[v]) == oq, v
        if st[v] == GT_NORMAL:
            p = probs[int(goff[v]):int(goff[v + 1])]
            assert len(p) == len(norm)
            both_nan = np.isnan(p) & np.isnan(norm)     # every raw value underflowed: 0/0 on both sides
            assert np.all(both_nan | (np.abs(p - norm) <= TOL)), v
            exact += int(np.sum((p == norm) | both_nan)); total += len(p)
    print("normalised likelihoods bit-identical: %d / %d" % (exact, total))
    assert exact == total             # logf, ln and exp restate the host libm's algorithms: bit-identical
    ctx.close()


def test_lookup_cover_flat_descriptors():
    """signatures with several k-mers (truncating running mean), several signatures per allele (max),
    zero weights skipped, empty alleles"""
    rng = np.random.default_rng(8)
    k, bits = 35, 1 << 20
    kms = [bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=k)) for _ in range(400)]
    ctx = Context(k, 43, bits)
    obf, omap = ocapi.BF(bits), ocapi.KMAP()
    for i, km in enumerate(kms):
        (omap if i % 2 == 0 else obf).add_key(km)
    ctx.map_insert(pad_rows(np.array([list(k_) for k_ in kms[0::2]], dtype=np.uint8)))
    ctx.bf_insert(BF_ALT, pad_rows(np.array([list(k_) for k_ in kms[1::2]], dtype=np.uint8)))
    obf.switch_mode(); ctx.bf_finalize(BF_ALT)
    w = rng.integers(0, 80, size=len(kms))
    for i, km in enumerate(kms):
        if w[i]:
            (omap.increment if i % 2 == 0 else obf.increment)(km, int(w[i]))
    ctx.map_increment(pad_rows(np.array([list(k_) for k_ in kms[0::2]], dtype=np.uint8)), w[0::2].astype(np.int32))
    ctx.bf_increment(BF_ALT, pad_rows(np.array([list(k_) for k_ in kms[1::2]], dtype=np.uint8)), w[1::2].astype(np.uint32))
    # descriptors: 60 allele slots, 0..3 signatures each, 1..6 k-mers per signature
    rows, is_ref, so, ao = [], [], [0], [0]
    for a in range(60):
        ref = a % 3 == 0
        for _ in range(rng.integers(0, 4)):
            for _ in range(rng.integers(1, 7)):
                i = int(rng.integers(0, 200)) * 2 + (0 if ref else 1)
                km = kms[i] if rng.random() < 0.8 else bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), size=k))
                rows.append(km); is_ref.append(1 if ref else 0)
            so.append(len(rows))
        ao.append(len(so) - 1)
    got = ctx.lookup_cover(rows, is_ref, so, ao)
    orows, _ = ocapi.rows_from_kmers(rows)
    want = ocapi.set_coverages(ocapi.lookup_weights(obf, omap, orows, np.array(is_ref, np.uint8)), so, ao)
    assert np.array_equal(got, want)
    ctx.close()


@pytest.mark.parametrize("haploid", [False, True])
def test_call_isolated_mixed_variants(haploid):
    """fused device path vs the oracle's loop-B restatement on SNPs, MNPs, indels, 2-3 alleles"""
    k, ref_k, bits = 35, 43, 1 << 24
    panel = synth.mixed_panel(3000, 31 + haploid, k=k)
    # every 7th variant gets a non-ACGT base inside its window: those take the byte-wise path on the device
    for v in range(0, panel.n, 7):
        panel.genome[panel.pos[v] - 1 - (v % 15)] = ord("N") if v % 2 else ord("R")
    ctx = Context(k, ref_k, bits)
    obf, octx, omap = build_index_pair(ctx, panel, k, ref_k, bits)
    # weights: every signature k-mer gets a random count, through the ASCII increment API on both sides
    rows, valid = synth.signature_rows(panel, k)
    is_ref = np.zeros(rows.shape[0], dtype=np.uint8)
    is_ref[panel.var_allele_off[:-1]] = 1
    rng = np.random.default_rng(77)
    w = rng.integers(0, 70, size=rows.shape[0]).astype(np.uint32)
    w[rng.random(rows.shape[0]) < 0.02] = 250        # some over-covered alleles
    # REF signatures holding a non-ACGT byte are exact-map keys no KMC k-mer can ever match, so a scan
    # leaves them at 0; the fused device path relies on that (they live in the host-side overflow list)
    w[(is_ref == 1) & ~(synth.CODE[rows] <= 3).all(axis=1)] = 0
    sel = valid & (w > 0)
    pr = pad_rows(rows[sel])
    for r, isr, c in zip(pr, is_ref[sel], w[sel]):
        km = bytes(r).split(b"\0", 1)[0]
        (omap.increment if isr else obf.increment)(km, int(c))
    ctx.map_increment(pr[is_ref[sel] == 1], w[sel][is_ref[sel] == 1].astype(np.int32))
    ctx.bf_increment(BF_ALT, pr[is_ref[sel] == 0], w[sel][is_ref[sel] == 0])
    ocov, og1, og2, ogq = ocapi.call_isolated(obf, omap, panel.genome.tobytes(), panel.pos, panel.allele_off,
                                              panel.var_allele_off, panel.pool, panel.freq, panel.present_mask,
                                              panel.flags & 1, k, 0.001, 200, haploid)
    ctx.reference_upload(panel.genome)
    cov, g1, g2, gq, st, probs, goff = ctx.call_isolated(panel.pos.astype(np.uint64), panel.var_allele_off, panel.allele_off,
                                                         panel.pool, panel.freq, panel.present_mask, panel.flags, 0.001, 200,
                                                         haploid, want_probs=True)
    assert np.array_equal(cov, ocov)
    assert np.array_equal(g1, og1) and np.array_equal(g2, og2) and np.array_equal(gq, ogq)
    for v in range(0, panel.n, 5):
        if st[v] != GT_NORMAL:
            continue
        a0, a1 = int(panel.var_allele_off[v]), int(panel.var_allele_off[v + 1])
        _, _, _, norm, _ = _oracle_variant(ocov[a0:a1], panel.freq[a0:a1], haploid)
        pv = probs[int(goff[v]):int(goff[v + 1])]
        both_nan = np.isnan(pv) & np.isnan(norm)
        assert np.all(both_nan | (np.abs(pv - norm) <= TOL)), v
    assert (st == GT_OVERCOV).any() and (st == GT_NOCOV).any() and (st == GT_NORMAL).any()
    ctx.close()
