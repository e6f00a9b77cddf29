"""Synthetic inputs for the tests and bench.py: a random genome, an isolated
variant panel, a donor and a KMC-style k-mer table.

Data generation only -- nothing here computes any part of the hot path's
results.  Everything is numpy, seeded, and vectorised so the 1e8-row table of
BASELINE.json's config C3 takes about a minute to build.

Packed k-mer layout (the C ABI's table format): an n-base string as a 128-bit
value hi:lo, base i at bits 2(n-1-i)+1..2(n-1-i)  ("M-form": MSB first, right
aligned), A=0 C=1 G=2 T=3.
"""
from dataclasses import dataclass

import numpy as np

U = np.uint64
_M2 = U(0x3333333333333333)
_M4 = U(0x0F0F0F0F0F0F0F0F)
CODE = np.full(256, 255, dtype=np.uint8)
for _i, _c in enumerate(b"ACGT"):
    CODE[_c] = _i
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)


def pairrev64(x):
    """reverse the order of the 32 two-bit groups of each uint64"""
    x = np.asarray(x, dtype=U)
    x = ((x >> U(2)) & _M2) | ((x & _M2) << U(2))
    x = ((x >> U(4)) & _M4) | ((x & _M4) << U(4))
    return x.byteswap()


def _shr128(hi, lo, s):
    if s == 0:
        return hi, lo
    if s < 64:
        return hi >> U(s), (lo >> U(s)) | (hi << U(64 - s))
    return np.zeros_like(hi), hi >> U(s - 64)


def _mask(n):
    bits = 2 * n
    lo = U(0xFFFFFFFFFFFFFFFF) if bits >= 64 else U((1 << bits) - 1)
    hi = U(0) if bits <= 64 else (U(0xFFFFFFFFFFFFFFFF) if bits >= 128 else U((1 << (bits - 64)) - 1))
    return hi, lo


def revcomp_m(hi, lo, n):
    """M-form of the reverse complement"""
    rh, rl = _shr128(pairrev64(lo), pairrev64(hi), 2 * (64 - n))
    mh, ml = _mask(n)
    return (~rh) & mh, (~rl) & ml


def canonical_m(hi, lo, n):
    """min(kmer, revcomp) in strcmp order == integer order of the M-forms"""
    rh, rl = revcomp_m(hi, lo, n)
    take_rc = (rh < hi) | ((rh == hi) & (rl < lo))
    return np.where(take_rc, rh, hi), np.where(take_rc, rl, lo)


def pack_codes(codes):
    """uint8 [n, L] of 0..3 -> (hi, lo) M-form"""
    codes = np.asarray(codes, dtype=np.uint8)
    n, L = codes.shape
    hi = np.zeros(n, dtype=U)
    lo = np.zeros(n, dtype=U)
    for i in range(L):
        sh = 2 * (L - 1 - i)
        c = codes[:, i].astype(U)
        if sh >= 64:
            hi |= c << U(sh - 64)
        else:
            lo |= c << U(sh)
    return hi, lo


def pack_ascii(rows):
    """uint8 [n, L] of upper-case ACGT bytes -> (hi, lo)"""
    codes = CODE[np.asarray(rows, dtype=np.uint8)]
    if (codes > 3).any():
        raise ValueError("non-ACGT byte in a k-mer to pack")
    return pack_codes(codes)


def unpack_ascii(hi, lo, n, stride=None):
    """(hi, lo) M-form -> uint8 [rows, stride] NUL-padded ASCII"""
    hi = np.asarray(hi, dtype=U)
    lo = np.asarray(lo, dtype=U)
    stride = stride or (n + 1 + 7) // 8 * 8
    out = np.zeros((hi.shape[0], stride), dtype=np.uint8)
    for i in range(n):
        sh = 2 * (n - 1 - i)
        c = ((hi >> U(sh - 64)) if sh >= 64 else (lo >> U(sh))) & U(3)
        out[:, i] = BASES[c.astype(np.int64)]
    return out


def random_genome(length, seed):
    rng = np.random.default_rng(seed)
    return BASES[rng.integers(0, 4, size=length, dtype=np.uint8)]


def windows(genome, starts, width):
    """uint8 [len(starts), width] windows of a byte array"""
    idx = np.asarray(starts, dtype=np.int64)[:, None] + np.arange(width, dtype=np.int64)[None, :]
    return genome[idx]


@dataclass
class Panel:
    """Flat description of an isolated-variant panel over one contig (what a
    host VCF reader would hand to mg_call_isolated)."""
    genome: np.ndarray            # uint8 ASCII
    pos: np.ndarray               # int64, 0-based
    var_allele_off: np.ndarray    # uint32 [n+1]
    allele_off: np.ndarray        # uint32 [n_alleles+1] offsets into pool
    pool: np.ndarray              # uint8 allele bytes
    freq: np.ndarray              # float32 per allele slot (slot 0 = REF)
    present_mask: np.ndarray      # uint64 per variant
    flags: np.ndarray             # uint8 per variant (bit0 eligible)
    donor_gt: np.ndarray          # int8 [n, 2] allele index per donor haplotype

    @property
    def n(self):
        return len(self.pos)

    def allele(self, v, a):
        s = int(self.var_allele_off[v]) + a
        return self.pool[self.allele_off[s]:self.allele_off[s + 1]]

    def n_alleles(self, v):
        return int(self.var_allele_off[v + 1] - self.var_allele_off[v])


def snp_panel(n_vars, seed, spacing=100, first=1000, tail=1000):
    """SURVEY 8(d) C3 recipe: isolated biallelic SNPs every `spacing` bases on a
    random genome; AF = (1 + draw % 4999) / 10000; both alleles present in the panel."""
    rng = np.random.default_rng(seed)
    length = first + spacing * n_vars + tail
    genome = random_genome(length, seed + 1)
    pos = first + spacing * np.arange(n_vars, dtype=np.int64)
    ref_code = CODE[genome[pos]]
    alt_code = (ref_code + 1 + rng.integers(0, 3, size=n_vars, dtype=np.uint8)) % 4
    pool = np.empty(2 * n_vars, dtype=np.uint8)
    pool[0::2] = genome[pos]
    pool[1::2] = BASES[alt_code]
    af = ((1 + rng.integers(0, 4999, size=n_vars)) / 10000.0).astype(np.float32)
    freq = np.empty(2 * n_vars, dtype=np.float32)
    freq[1::2] = af
    # frequencies[0] = (float)(1.0 - (double)sum)  (variant.hpp:143)
    freq[0::2] = (1.0 - af.astype(np.float64)).astype(np.float32)
    donor = rng.integers(0, 2, size=(n_vars, 2), dtype=np.int8)
    return Panel(genome=genome, pos=pos,
                 var_allele_off=(2 * np.arange(n_vars + 1)).astype(np.uint32),
                 allele_off=np.arange(2 * n_vars + 1, dtype=np.uint32), pool=pool, freq=freq,
                 present_mask=np.full(n_vars, 3, dtype=np.uint64), flags=np.ones(n_vars, dtype=np.uint8),
                 donor_gt=donor)


def mixed_panel(n_vars, seed, k=35, spacing=120, first=1000, tail=1000, max_alleles=3, max_len=12):
    """Isolated variants of mixed type: SNPs, MNPs, insertions, deletions, 2..max_alleles
    alleles (all shorter than k), some alleles absent from the panel, some variants not present."""
    rng = np.random.default_rng(seed)
    length = first + spacing * n_vars + tail
    genome = random_genome(length, seed + 1)
    pos = first + spacing * np.arange(n_vars, dtype=np.int64)
    vo, ao, pool, freq, pm, flags, donor = [0], [0], [], [], [], [], []
    for v in range(n_vars):
        A = int(rng.integers(2, max_alleles + 1))
        ref_len = int(rng.integers(1, max_len + 1)) if rng.random() < 0.4 else 1
        alleles = [bytes(genome[pos[v]:pos[v] + ref_len])]
        while len(alleles) < A:
            alen = int(rng.integers(1, max_len + 1)) if rng.random() < 0.5 else 1
            cand = bytes(BASES[rng.integers(0, 4, size=alen)])
            if cand not in alleles:
                alleles.append(cand)
        fr = rng.dirichlet(np.ones(A)).astype(np.float32)
        if rng.random() < 0.05:
            fr[1:] = 0  # -> f[0] == 1 -> not present
        f0 = np.float32(1.0 - float(np.sum(fr[1:].astype(np.float64))))
        fr[0] = max(f0, np.float32(0))
        present = fr[0] != np.float32(1.0)
        mask = 0
        for a in range(A):
            if a < 2 or rng.random() < 0.7:
                mask |= 1 << a
        for a, al in enumerate(alleles):
            pool.extend(al)
            ao.append(len(pool))
            freq.append(fr[a])
        vo.append(vo[-1] + A)
        pm.append(mask)
        flags.append(1 if present else 0)
        donor.append([int(rng.integers(0, A)), int(rng.integers(0, A))])
    return Panel(genome=genome, pos=pos, var_allele_off=np.array(vo, dtype=np.uint32),
                 allele_off=np.array(ao, dtype=np.uint32), pool=np.array(pool, dtype=np.uint8),
                 freq=np.array(freq, dtype=np.float32), present_mask=np.array(pm, dtype=np.uint64),
                 flags=np.array(flags, dtype=np.uint8), donor_gt=np.array(donor, dtype=np.int8))


def signature_rows(panel: Panel, k):
    """The single signature k-mer of every (variant, allele) of an isolated panel:
    ref[pos-mp, pos) + allele + ref[pos+ref_size, +ms).  Returns (rows uint8 [n_alleles, k],
    valid mask).  Input construction for index building in the tests/bench; the device
    path rebuilds these itself inside call_isolated."""
    na = int(panel.var_allele_off[-1])
    rows = np.zeros((na, k), dtype=np.uint8)
    valid = np.zeros(na, dtype=bool)
    alen = np.diff(panel.allele_off.astype(np.int64))
    for v in range(panel.n):
        a0, a1 = int(panel.var_allele_off[v]), int(panel.var_allele_off[v + 1])
        if not (panel.flags[v] & 1):
            continue
        rs = int(alen[a0])
        p = int(panel.pos[v])
        for s in range(a0, a1):
            if not ((int(panel.present_mask[v]) >> (s - a0)) & 1):
                continue
            L = int(alen[s])
            mp, ms = k // 2 - L // 2, (k + 1) // 2 - (L - L // 2)
            rows[s, :mp] = panel.genome[p - mp:p]
            rows[s, mp:mp + L] = panel.pool[panel.allele_off[s]:panel.allele_off[s + 1]]
            rows[s, mp + L:] = panel.genome[p + rs:p + rs + ms]
            valid[s] = True
    return rows, valid


def snp_signature_rows(panel: Panel, k):
    """vectorised signature_rows for snp_panel panels (all alleles one base)"""
    h = k // 2
    n = panel.n
    w = windows(panel.genome, panel.pos - h, k)
    rows = np.repeat(w, 2, axis=0)
    rows[:, h] = panel.pool
    return rows, np.ones(2 * n, dtype=bool)


def site_rows(panel: Panel, k, ref_k, offsets=(-2, -1, 0, 1, 2)):
    """ref_k-mers of the donor's two haplotypes around every variant site of an snp_panel (window offsets around the
    centred position) -> (hi, lo, variant index, haplotype, offset) per row, M-form, not canonicalised.  A homozygous
    donor contributes each window once, as KMC lists distinct k-mers; windows holding a non-ACGT symbol are dropped."""
    off = (ref_k - k) // 2
    centre = off + k // 2                      # column of the variant base in a centred ref_k window
    his, los, vs, hs, ds = [], [], [], [], []
    idx = np.arange(panel.n, dtype=np.int64)
    for h in range(2):
        allele_slot = panel.var_allele_off[:-1].astype(np.int64) + panel.donor_gt[:, h].astype(np.int64)
        base = panel.pool[panel.allele_off[allele_slot]]
        for d in offsets:
            w = windows(panel.genome, panel.pos - centre + d, ref_k).copy()
            w[:, centre - d] = base
            keep = np.ones(panel.n, dtype=bool)
            if h == 1:
                keep &= panel.donor_gt[:, 0] != panel.donor_gt[:, 1]
            keep &= (CODE[w] <= 3).all(axis=1)       # KMC drops windows that hold a non-ACGT symbol
            a, b = pack_ascii(w[keep])
            his.append(a); los.append(b)
            vs.append(idx[keep]); hs.append(np.full(int(keep.sum()), h, dtype=np.int8)); ds.append(np.full(int(keep.sum()), d, dtype=np.int8))
    return np.concatenate(his), np.concatenate(los), np.concatenate(vs), np.concatenate(hs), np.concatenate(ds)


def kmer_table(panel: Panel, n_rows, k, ref_k, seed, offsets=(-2, -1, 0, 1, 2), snp_only=True):
    """KMC-style table: ref_k-mers of the donor's two haplotypes around every variant site
    (window offsets around the centred position) topped up to n_rows with uniform random
    ref_k-mers, all canonical, counts in [2, 63], rows shuffled.  -> (hi, lo, cnt)"""
    assert snp_only, "donor windows are generated for one-base alleles"
    rng = np.random.default_rng(seed)
    hi, lo, _, _, _ = site_rows(panel, k, ref_k, offsets)
    n_site = hi.shape[0]
    if n_site > n_rows:
        sel = rng.permutation(n_site)[:n_rows]
        hi, lo = hi[sel], lo[sel]
        n_site = n_rows
    n_rand = n_rows - n_site
    mh, ml = _mask(ref_k)
    rh = rng.integers(0, 1 << 63, size=n_rand, dtype=np.uint64) * U(2) + rng.integers(0, 2, size=n_rand, dtype=np.uint64)
    rl = rng.integers(0, 1 << 63, size=n_rand, dtype=np.uint64) * U(2) + rng.integers(0, 2, size=n_rand, dtype=np.uint64)
    hi = np.concatenate([hi, rh & mh]); lo = np.concatenate([lo, rl & ml])
    hi, lo = canonical_m(hi, lo, ref_k)
    perm = rng.permutation(n_rows)
    cnt = (2 + rng.integers(0, 62, size=n_rows)).astype(np.uint32)
    return hi[perm], lo[perm], cnt


def head(panel: Panel, n):
    """the first n variants of an snp_panel (biallelic, one base per allele) as a Panel of their own"""
    return Panel(genome=panel.genome, pos=panel.pos[:n], var_allele_off=panel.var_allele_off[:n + 1], allele_off=panel.allele_off[:2 * n + 1],
                 pool=panel.pool[:2 * n], freq=panel.freq[:2 * n], present_mask=panel.present_mask[:n], flags=panel.flags[:n],
                 donor_gt=panel.donor_gt[:n])


def device_table(panel: Panel, n_rows, k, ref_k, seed, device, plant_variants=None):
    """kmer_table for tables of 1e8..4e9 rows: the uniform random rows are drawn ON THE GPU (torch: random bits and a
    scatter; numpy needs minutes per 1e8 rows), the windows around the first `plant_variants` variant sites come from
    site_rows() and are scattered to distinct random places.  Rows are not canonicalised (the scan does that itself,
    main.cpp:491-499 via BF/KMAP).  -> dict: d_hi, d_lo (int64 bit patterns), d_cnt (int32), and for the planted rows
    their place, variant, haplotype, window offset and count (numpy), so a test knows what every counter must hold."""
    import torch
    dev = torch.device("cuda", device) if isinstance(device, int) else device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sub = panel if plant_variants is None else head(panel, plant_variants)
    hi, lo, var, hap, off = site_rows(sub, k, ref_k)
    n_site = int(hi.size)
    if n_site > n_rows:
        keep = np.random.default_rng(seed).permutation(n_site)[:n_rows]
        hi, lo, var, hap, off = hi[keep], lo[keep], var[keep], hap[keep], off[keep]
        n_site = n_rows

    def bits32():
        return torch.randint(0, 1 << 32, (n_rows,), dtype=torch.int64, device=dev, generator=g)
    top = 2 * ref_k - 64                                      # bits of the 2-bit string that live in `hi`
    d_lo = (bits32() << 32) | bits32()                        # 64 random bits as an int64 bit pattern
    if top >= 64:
        d_hi = (bits32() << 32) | bits32()
    elif top > 0:
        d_hi = ((bits32() << 32) | bits32()) & ((1 << top) - 1)
    else:
        d_hi = torch.zeros(n_rows, dtype=torch.int64, device=dev)
        if top < 0:
            d_lo = d_lo & ((1 << (2 * ref_k)) - 1)
    d_cnt = torch.randint(2, 64, (n_rows,), dtype=torch.int32, device=dev, generator=g)
    # distinct places for the planted rows: one per stride of n_rows / n_site rows, at a random offset inside it
    stride = n_rows // max(n_site, 1)
    where = torch.arange(n_site, dtype=torch.int64, device=dev) * stride
    if stride > 1:
        where += torch.randint(0, stride, (n_site,), dtype=torch.int64, device=dev, generator=g)
    if 1 < n_site < (1 << 31):          # site_rows() lists haplotype by haplotype, offset by offset: deal them over the places
        where = where[torch.randperm(n_site, device=dev, generator=g)]
    d_hi[where] = torch.from_numpy(hi.view(np.int64)).to(dev)
    d_lo[where] = torch.from_numpy(lo.view(np.int64)).to(dev)
    out = {"d_hi": d_hi, "d_lo": d_lo, "d_cnt": d_cnt, "n": n_rows, "n_site": n_site, "site_where": where.cpu().numpy(),
           "site_cnt": d_cnt[where].cpu().numpy().astype(np.int64), "site_var": var, "site_hap": hap, "site_off": off}
    torch.cuda.synchronize()
    return out


# ---- flat panels with CLUSTERS (the general-block path): BASELINE config C4 as SURVEY 8(d) draws it, and config C5 -----------

@dataclass
class FlatPanel:
    """The kept records of a panel in file order as the flat arrays of include/malva_hip.h's mg_panel_dev (+ frequencies
    and a donor).  Sequences are concatenated in `genome`; record v sits on sequence contig_id[v] at 0-based `pos[v]`."""
    genome: np.ndarray            # uint8 ASCII, all sequences concatenated
    contig_names: list
    contig_base: np.ndarray       # uint64 [n_contigs]
    contig_len: np.ndarray        # uint32 [n_contigs]
    contig_id: np.ndarray         # uint32 [n]
    pos: np.ndarray               # int32 [n]
    ref_size: np.ndarray          # uint32 [n]
    min_size: np.ndarray          # uint32 [n]  (Variant::min_size over REF and ALTs)
    present: np.ndarray           # uint8 [n]   (Variant::is_present)
    var_allele_off: np.ndarray    # uint32 [n + 1]
    allele_off: np.ndarray        # uint32 [slots + 1]
    pool: np.ndarray              # uint8 allele bytes
    canon: np.ndarray             # uint8 [slots] first allele of the variant with the same text
    freq: np.ndarray              # float32 [slots]
    gt: np.ndarray                # uint16 [n, n_samples]: a1 | a2 << 7 | phased << 14
    n_samples: int
    donor_gt: np.ndarray          # int8 [n, 2]

    @property
    def n(self):
        return len(self.pos)

    def gpos(self):
        """offset of every record in the concatenated genome"""
        return self.contig_base[self.contig_id].astype(np.int64) + self.pos.astype(np.int64)

    def allele(self, v, a):
        s = int(self.var_allele_off[v]) + a
        return bytes(self.pool[self.allele_off[s]:self.allele_off[s + 1]])

    def head(self, n):
        """the first n records as a panel of their own (same genome)"""
        return self.slice(0, n)

    def slice(self, a, b):
        """records [a, b) as a panel of their own (same genome and sequences)"""
        s0, s1 = int(self.var_allele_off[a]), int(self.var_allele_off[b])
        p0, p1 = int(self.allele_off[s0]), int(self.allele_off[s1])
        return FlatPanel(genome=self.genome, contig_names=self.contig_names, contig_base=self.contig_base, contig_len=self.contig_len,
                         contig_id=self.contig_id[a:b], pos=self.pos[a:b], ref_size=self.ref_size[a:b], min_size=self.min_size[a:b], present=self.present[a:b],
                         var_allele_off=(self.var_allele_off[a:b + 1] - self.var_allele_off[a]).astype(np.uint32),
                         allele_off=(self.allele_off[s0:s1 + 1] - self.allele_off[s0]).astype(np.uint32), pool=self.pool[p0:p1],
                         canon=self.canon[s0:s1], freq=self.freq[s0:s1], gt=self.gt[a:b], n_samples=self.n_samples, donor_gt=self.donor_gt[a:b])

    def split_points(self, parts, min_gap=64):
        """record indices 0 = c_0 <= c_1 <= ... <= c_parts = n that cut the panel into `parts` nearly equal runs WITHOUT cutting a
        block: every cut sits where the next record is on another sequence or more than `min_gap` nt behind the previous one
        (beyond are_near's reach for any k <= 64, float rounding included)"""
        n = self.n
        ok = np.ones(n + 1, dtype=bool)
        if n > 1:
            far = (np.diff(self.pos.astype(np.int64)) > min_gap + self.ref_size[:-1].astype(np.int64)) | (np.diff(self.contig_id.astype(np.int64)) != 0)
            ok[1:n] = far
        cand = np.nonzero(ok)[0]
        cuts = [0]
        for r in range(1, parts):
            want = n * r // parts
            j = int(np.searchsorted(cand, want))
            cuts.append(int(cand[min(j, cand.size - 1)]))
        cuts.append(n)
        return cuts


def flat_from_snp_panel(panel: "Panel") -> FlatPanel:
    """An snp_panel (isolated biallelic SNPs on one sequence, what mg_call_isolated takes) as the FlatPanel of the general record
    loop: one sequence, ref_size = min_size = 1, every record present, two phased samples 0|1 and 0|0 (both alleles present, as
    the Panel's present_mask = 3 says).  Same records, same frequencies: the general path must give what the lone path gives."""
    n = panel.n
    gt = np.empty((n, 2), dtype=np.uint16)
    gt[:, 0] = (1 << 7) | (1 << 14)       # 0|1
    gt[:, 1] = 1 << 14                    # 0|0
    vo = panel.var_allele_off.astype(np.uint32)
    ao = panel.allele_off.astype(np.uint32)
    canon = np.tile(np.array([0, 1], dtype=np.uint8), n)
    return FlatPanel(genome=panel.genome, contig_names=["1"], contig_base=np.zeros(1, dtype=np.uint64), contig_len=np.array([panel.genome.size], dtype=np.uint32),
                     contig_id=np.zeros(n, dtype=np.uint32), pos=panel.pos.astype(np.int32), ref_size=np.ones(n, dtype=np.uint32), min_size=np.ones(n, dtype=np.uint32),
                     present=np.ones(n, dtype=np.uint8), var_allele_off=vo, allele_off=ao, pool=panel.pool, canon=canon, freq=panel.freq, gt=gt, n_samples=2,
                     donor_gt=panel.donor_gt)


def _canon_of(var_allele_off, allele_off, pool):
    """first allele of the variant with the same text, per slot (Variant::get_allele_index, variant.hpp:228-240); numpy, vectorised
    over the common case (all alleles of a variant differ in their first 8 bytes or length) with a loop for the rest"""
    na = int(var_allele_off[-1])
    canon = np.zeros(na, dtype=np.uint8)
    A = np.diff(var_allele_off.astype(np.int64))
    local = np.arange(na, dtype=np.int64) - np.repeat(var_allele_off[:-1].astype(np.int64), A)
    canon[:] = local
    alen = np.diff(allele_off.astype(np.int64))
    multi = np.nonzero(A > 2)[0]
    for v in multi:                      # duplicates can only occur among ALTs (or an ALT spelled like REF): rare, checked by text
        a0 = int(var_allele_off[v])
        seen = {}
        for a in range(int(A[v])):
            t = bytes(pool[allele_off[a0 + a]:allele_off[a0 + a + 1]])
            canon[a0 + a] = seen.setdefault(t, a)
    two = np.nonzero(A == 2)[0]
    if two.size:                         # biallelic: ALT == REF text?
        s0 = var_allele_off[two].astype(np.int64)
        same_len = alen[s0] == alen[s0 + 1]
        for v in two[same_len]:
            a0 = int(var_allele_off[v])
            if bytes(pool[allele_off[a0]:allele_off[a0 + 1]]) == bytes(pool[allele_off[a0 + 1]:allele_off[a0 + 2]]):
                canon[a0 + 1] = 0
    return canon


def clustered_snp_panel(n_vars, seed, n_contigs=24, mean_spacing=38, cluster_frac=0.10, max_cluster=4, cluster_span=17, n_samples=2, k=35):
    """BASELINE config C4's panel as SURVEY 8(d) draws it: biallelic SNPs at ~`mean_spacing` nt mean spacing, `cluster_frac` of
    them in clusters of 2..max_cluster within `cluster_span` nt, on `n_contigs` sequences (a 3.1e9-nt genome at 8e7 SNPs: 24
    sequences keep positions inside int32, as a VCF's do).  Gaps between units are drawn from [k/2 + 3, 2 mean - k/2 - 3], so
    the units are apart by more than are_near's reach at low positions; beyond 2^24 the reference's float rule (var_block.hpp
    :417-423) joins a few more neighbours, which is the reference's semantics, not the generator's.  n_samples diploid phased
    samples, GT bits random, forced so that every ALT is carried at least once (else is_present filtering drops it)."""
    rng = np.random.default_rng(seed)
    n_clustered = int(round(n_vars * cluster_frac))
    sizes = []
    left = n_clustered
    csz = rng.integers(2, max_cluster + 1, size=max(1, n_clustered // 2 + 1))
    cs = np.cumsum(csz)
    n_cl = int(np.searchsorted(cs, left, side="right"))
    csz = csz[:n_cl]
    left -= int(csz.sum())
    n_single = n_vars - int(csz.sum())
    unit = np.concatenate([np.ones(n_single, dtype=np.int64), csz.astype(np.int64)])
    rng.shuffle(unit)
    U = unit.size
    # offsets inside clusters: distinct values of 1..cluster_span, sorted (first member at 0)
    is_cl = unit > 1
    ncl = int(is_cl.sum())
    offs = np.zeros((ncl, max_cluster), dtype=np.int64)
    if ncl:
        perm = rng.permuted(np.tile(np.arange(1, cluster_span + 1, dtype=np.int8), (ncl, 1)), axis=1)[:, :max_cluster - 1].astype(np.int64)
        take = unit[is_cl][:, None] - 1 > np.arange(max_cluster - 1)[None, :]
        perm = np.where(take, perm, 1 << 20)
        perm.sort(axis=1)
        offs[:, 1:] = perm
    half = (k + 1) // 2
    lo_gap, hi_gap = half + 3, max(half + 4, 2 * mean_spacing - half - 3)
    gap = rng.integers(lo_gap, hi_gap + 1, size=U).astype(np.int64)         # from the END of the previous unit to this unit's first record
    span = np.zeros(U, dtype=np.int64)
    if ncl:
        last = np.take_along_axis(offs, (unit[is_cl] - 1)[:, None], axis=1)[:, 0]
        span[is_cl] = last
    # deal the units over the sequences in order: unit u -> sequence u * n_contigs // U
    cid_u = (np.arange(U, dtype=np.int64) * n_contigs) // U
    first = 1000
    step = gap + np.concatenate([[0], span[:-1]])
    run = np.cumsum(step)
    firsts = np.searchsorted(cid_u, np.arange(n_contigs))
    base_run = run[firsts] - gap[firsts]                                        # what the run held before each sequence's first unit
    start = first + run - np.repeat(base_run, np.diff(np.append(firsts, U)))
    # records
    rec_unit = np.repeat(np.arange(U), unit)
    within = np.arange(n_vars, dtype=np.int64) - np.repeat(np.cumsum(unit) - unit, unit)
    pos = start[rec_unit].copy()
    if ncl:
        cl_index = np.cumsum(is_cl) - 1
        m = is_cl[rec_unit]
        pos[m] += offs[cl_index[rec_unit[m]], within[m]]
    contig_id = cid_u[rec_unit].astype(np.uint32)
    ends = np.zeros(n_contigs, dtype=np.int64)
    np.maximum.at(ends, contig_id, pos)
    contig_len = (ends + 1000).astype(np.uint32)
    contig_base = np.zeros(n_contigs, dtype=np.uint64)
    contig_base[1:] = np.cumsum(contig_len.astype(np.uint64))[:-1]
    genome = random_genome(int(contig_len.astype(np.int64).sum()), seed + 1)
    gpos = contig_base[contig_id].astype(np.int64) + pos
    ref_code = CODE[genome[gpos]]
    alt_code = (ref_code + 1 + rng.integers(0, 3, size=n_vars, dtype=np.uint8)) % 4
    pool = np.empty(2 * n_vars, dtype=np.uint8)
    pool[0::2] = genome[gpos]
    pool[1::2] = BASES[alt_code]
    af = ((1 + rng.integers(0, 4999, size=n_vars)) / 10000.0).astype(np.float32)
    freq = np.empty(2 * n_vars, dtype=np.float32)
    freq[1::2] = af
    freq[0::2] = (1.0 - af.astype(np.float64)).astype(np.float32)
    bits = rng.integers(0, 2, size=(n_vars, n_samples, 2), dtype=np.uint8)
    none = ~bits.reshape(n_vars, -1).any(axis=1)
    bits[none, 0, 0] = 1
    gt = (bits[:, :, 0].astype(np.uint16) | (bits[:, :, 1].astype(np.uint16) << 7) | np.uint16(1 << 14))
    donor = rng.integers(0, 2, size=(n_vars, 2), dtype=np.int8)
    ones = np.ones(n_vars, dtype=np.uint32)
    return FlatPanel(genome=genome, contig_names=[str(i + 1) for i in range(n_contigs)], contig_base=contig_base, contig_len=contig_len,
                     contig_id=contig_id, pos=pos.astype(np.int32), ref_size=ones, min_size=ones.copy(), present=np.ones(n_vars, dtype=np.uint8),
                     var_allele_off=(2 * np.arange(n_vars + 1)).astype(np.uint32), allele_off=np.arange(2 * n_vars + 1, dtype=np.uint32), pool=pool,
                     canon=np.tile(np.array([0, 1], dtype=np.uint8), n_vars), freq=freq, gt=gt, n_samples=n_samples, donor_gt=donor)


def indel_panel(n_clusters, seed, k=35, n_samples=8, n_contigs=2, unphased_frac=0.5, max_cluster=6, cluster_gap=400, hom_ref=0.45):
    """BASELINE config C5's panel: clusters of 1..max_cluster records mixing SNPs, MNPs, 1-30 nt deletions and
    1-60 nt insertions (some of k bases and more: the sliding-signature path), up to 3 ALTs, a few records overlapping the
    deletion before them, `n_samples` samples of which `unphased_frac` of the genotypes are unphased; clusters `cluster_gap`
    nt apart on average.  Runs haploid (first allele of every genotype) and diploid from the same arrays."""
    rng = np.random.default_rng(seed)
    csz = rng.integers(1, max_cluster + 1, size=n_clusters).astype(np.int64)
    n = int(csz.sum())
    rec_cl = np.repeat(np.arange(n_clusters), csz)
    within = np.arange(n, dtype=np.int64) - np.repeat(np.cumsum(csz) - csz, csz)
    kind = rng.random(n)
    # REF length: SNP / insertion anchor 1; MNP 2..6; deletion 2..31
    is_mnp = (kind >= 0.40) & (kind < 0.55)
    is_del = (kind >= 0.55) & (kind < 0.75)
    is_ins = kind >= 0.75
    ref_len = np.ones(n, dtype=np.int64)
    ref_len[is_mnp] = rng.integers(2, 7, size=int(is_mnp.sum()))
    ref_len[is_del] = rng.integers(2, 32, size=int(is_del.sum()))
    n_alt = np.where(rng.random(n) < 0.85, 1, rng.integers(2, 4, size=n)).astype(np.int64)
    A = n_alt + 1
    vo = np.zeros(n + 1, dtype=np.int64)
    vo[1:] = np.cumsum(A)
    na = int(vo[-1])
    slot_var = np.repeat(np.arange(n), A)
    slot_a = np.arange(na, dtype=np.int64) - vo[slot_var]
    alen = np.ones(na, dtype=np.int64)
    ref_slot = slot_a == 0
    alen[ref_slot] = ref_len
    v_of = slot_var
    alt = ~ref_slot
    r = rng.random(na)
    mnp_alt = alt & is_mnp[v_of]
    alen[mnp_alt] = ref_len[v_of[mnp_alt]]                                        # MNP: same length
    del_alt = alt & is_del[v_of]
    alen[del_alt] = np.where(r[del_alt] < 0.8, 1, rng.integers(1, 6, size=int(del_alt.sum())))
    ins_alt = alt & is_ins[v_of]
    long_ins = ins_alt & (r < 0.06)
    alen[ins_alt] = rng.integers(2, 61, size=int(ins_alt.sum()))
    alen[long_ins] = rng.integers(k, k + 26, size=int(long_ins.sum()))           # alleles of k bases and more
    ao = np.zeros(na + 1, dtype=np.int64)
    ao[1:] = np.cumsum(alen)
    # positions: inside a cluster the next record starts 1..20 nt behind the previous one's REF span -- or, one time in
    # ten, inside it (overlapping records: the walks' "shorten and retry" branch)
    adv = ref_len + rng.integers(0, 20, size=n)
    inside = (rng.random(n) < 0.10) & (ref_len > 1)
    adv[inside] = rng.integers(1, np.maximum(ref_len[inside], 2))
    cl_step = rng.integers(cluster_gap // 2, cluster_gap * 3 // 2 + 1, size=n_clusters).astype(np.int64) + 2 * k + 140
    prev_adv = np.concatenate([[0], adv[:-1]])
    prev_adv[within == 0] = 0
    cl_of = rec_cl
    cid_cl = (np.arange(n_clusters, dtype=np.int64) * n_contigs) // n_clusters
    firsts = np.searchsorted(cid_cl, np.arange(n_contigs))
    # cluster start = running sum of (cluster step + the previous cluster's extent) inside its sequence
    ext = np.zeros(n_clusters, dtype=np.int64)
    np.add.at(ext, cl_of, prev_adv)
    last_ref = np.zeros(n_clusters, dtype=np.int64)
    np.maximum.at(last_ref, cl_of, ref_len)
    step = cl_step + np.concatenate([[0], (ext + last_ref)[:-1]])
    run = np.cumsum(step)
    base_run = run[firsts] - cl_step[firsts]
    cl_start = 500 + run - np.repeat(base_run, np.diff(np.append(firsts, n_clusters)))
    off_in = np.cumsum(prev_adv) - np.repeat((np.cumsum(prev_adv) - prev_adv)[np.cumsum(csz) - csz], csz)
    pos = cl_start[cl_of] + off_in
    contig_id = cid_cl[cl_of].astype(np.uint32)
    ends = np.zeros(n_contigs, dtype=np.int64)
    np.maximum.at(ends, contig_id, pos + ref_len)
    contig_len = (ends + 500 + 2 * k).astype(np.uint32)
    contig_base = np.zeros(n_contigs, dtype=np.uint64)
    contig_base[1:] = np.cumsum(contig_len.astype(np.uint64))[:-1]
    genome = random_genome(int(contig_len.astype(np.int64).sum()), seed + 1)
    gpos = contig_base[contig_id].astype(np.int64) + pos
    pool = BASES[rng.integers(0, 4, size=int(ao[-1]), dtype=np.uint8)]
    # REF alleles spell the genome
    rs = np.nonzero(ref_slot)[0]
    idx = np.repeat(ao[rs], ref_len) + (np.arange(int(ref_len.sum()), dtype=np.int64) - np.repeat(np.cumsum(ref_len) - ref_len, ref_len))
    src = np.repeat(gpos, ref_len) + (np.arange(int(ref_len.sum()), dtype=np.int64) - np.repeat(np.cumsum(ref_len) - ref_len, ref_len))
    pool[idx] = genome[src]
    # one-base ALTs of one-base REFs differ from REF
    snp_alt = alt & (alen == 1) & (ref_len[v_of] == 1)
    sidx = ao[:-1][snp_alt]
    pool[sidx] = BASES[(CODE[genome[gpos[v_of[snp_alt]]]] + 1 + rng.integers(0, 3, size=sidx.size, dtype=np.uint8)) % 4]
    min_size = np.full(n, 1 << 30, dtype=np.int64)
    np.minimum.at(min_size, v_of, alen)
    # frequencies: AF per ALT = (1 + draw % 2999) / 10000 (sums stay below 1), f[0] = (float)(1 - sum)  (variant.hpp:134-146)
    freq = np.zeros(na, dtype=np.float32)
    freq[alt] = ((1 + rng.integers(0, 2999, size=int(alt.sum()))) / 10000.0).astype(np.float32)
    acc = np.zeros(n, dtype=np.float64)
    np.add.at(acc, v_of, freq.astype(np.float64))
    freq[ref_slot] = (1.0 - acc).astype(np.float32)
    # genotypes
    a1 = (rng.random((n, n_samples)) * A[:, None]).astype(np.uint16)
    a2 = (rng.random((n, n_samples)) * A[:, None]).astype(np.uint16)
    is_hom_ref = rng.random((n, n_samples)) < hom_ref          # (hom_ref near 1: a large panel, nearly all 0|0)
    a1[is_hom_ref] = 0
    a2[is_hom_ref & (rng.random((n, n_samples)) < (0.8 if hom_ref < 0.9 else 1.0))] = 0
    carried = (a1 > 0).any(axis=1)
    a1[~carried, 0] = 1                                                           # every record carries an ALT in some first haplotype
    phased = rng.random((n, n_samples)) >= unphased_frac
    gt = a1 | (a2 << 7) | (phased.astype(np.uint16) << 14)
    donor = np.stack([(rng.random(n) * A).astype(np.int8), (rng.random(n) * A).astype(np.int8)], axis=1)
    vo32, ao32 = vo.astype(np.uint32), ao.astype(np.uint32)
    return FlatPanel(genome=genome, contig_names=[str(i + 1) for i in range(n_contigs)], contig_base=contig_base, contig_len=contig_len,
                     contig_id=contig_id, pos=pos.astype(np.int32), ref_size=ref_len.astype(np.uint32), min_size=min_size.astype(np.uint32),
                     present=np.ones(n, dtype=np.uint8), var_allele_off=vo32, allele_off=ao32, pool=pool, canon=_canon_of(vo32, ao32, pool), freq=freq,
                     gt=gt.astype(np.uint16), n_samples=n_samples, donor_gt=donor)


def donor_rows(panel: FlatPanel, ref_k, max_records=None, margin=None):
    """ref_k-mers of the two donor haplotypes around the panel's records -> (hi, lo) M-form, not canonicalised, duplicates
    removed.  The donor carries donor_gt[v][h] at every record that does not overlap the previous applied one (else REF).
    Built cluster by cluster: records closer than 2 ref_k share a stretch of the haplotype.  `max_records` bounds the work
    (the first so many records)."""
    n = panel.n if max_records is None else min(panel.n, int(max_records))
    margin = ref_k if margin is None else margin
    gpos = panel.gpos()
    his, los = [], []
    for h in range(2):
        pieces = []                                   # the haplotype around the records, stretches separated by 'N'
        v = 0
        while v < n:
            cid = panel.contig_id[v]
            cb, cl = int(panel.contig_base[cid]), int(panel.contig_len[cid])
            s0 = max(cb, int(gpos[v]) - margin)
            cur = s0
            out = []
            w = v
            while w < n and panel.contig_id[w] == cid and (w == v or int(gpos[w]) <= end_reach):
                p = int(gpos[w])
                if p >= cur:                          # not overlapping what was applied before
                    out.append(panel.genome[cur:p])
                    a = int(panel.donor_gt[w, h])
                    s = int(panel.var_allele_off[w]) + a
                    out.append(panel.pool[panel.allele_off[s]:panel.allele_off[s + 1]])
                    cur = p + int(panel.ref_size[w])
                end_reach = max(cur, p + int(panel.ref_size[w])) + 2 * margin
                w += 1
            out.append(panel.genome[cur:min(cb + cl, cur + margin)])
            pieces.append(np.concatenate(out))
            pieces.append(np.frombuffer(b"N", dtype=np.uint8))
            v = w
        hap = np.concatenate(pieces)
        if hap.size >= ref_k:
            codes = CODE[hap]
            bad = np.concatenate([[0], np.cumsum(codes > 3)])
            ok = (bad[ref_k:] - bad[:-ref_k]) == 0
            starts = np.nonzero(ok)[0]
            for a in range(0, starts.size, 1 << 20):
                st = starts[a:a + (1 << 20)]
                hi, lo = pack_codes(codes[st[:, None] + np.arange(ref_k)[None, :]])
                his.append(hi); los.append(lo)
    hi, lo = np.concatenate(his), np.concatenate(los)
    key = np.unique(np.stack([hi, lo], axis=1), axis=0)
    return key[:, 0].copy(), key[:, 1].copy()


def flat_kmer_table(panel: FlatPanel, n_rows, k, ref_k, seed, max_records=None):
    """KMC-style table for a FlatPanel: the donor's ref_k-mers around the records (donor_rows) topped up to n_rows with
    uniform random ref_k-mers, canonical, counts in [2, 63], shuffled -> (hi, lo, cnt)"""
    rng = np.random.default_rng(seed)
    hi, lo = donor_rows(panel, ref_k, max_records)
    if hi.size > n_rows:
        sel = rng.permutation(hi.size)[:n_rows]
        hi, lo = hi[sel], lo[sel]
    n_rand = n_rows - hi.size
    mh, ml = _mask(ref_k)
    rh = rng.integers(0, 1 << 63, size=n_rand, dtype=np.uint64) * U(2) + rng.integers(0, 2, size=n_rand, dtype=np.uint64)
    rl = rng.integers(0, 1 << 63, size=n_rand, dtype=np.uint64) * U(2) + rng.integers(0, 2, size=n_rand, dtype=np.uint64)
    hi = np.concatenate([hi, rh & mh]); lo = np.concatenate([lo, rl & ml])
    hi, lo = canonical_m(hi, lo, ref_k)
    perm = rng.permutation(n_rows)
    cnt = (2 + rng.integers(0, 62, size=n_rows)).astype(np.uint32)
    return hi[perm], lo[perm], cnt


def write_vcf_fasta(panel: FlatPanel, prefix, freq_key="AF"):
    """the panel as <prefix>.fa + <prefix>.vcf (diploid GT columns as held), so the same records can go through the CLI
    and through the oracle's VCF model"""
    with open(prefix + ".fa", "w") as fh:
        for name, b, l in zip(panel.contig_names, panel.contig_base, panel.contig_len):
            fh.write(">%s\n" % name)
            seq = panel.genome[int(b):int(b) + int(l)].tobytes().decode()
            for a in range(0, len(seq), 1 << 16):
                fh.write(seq[a:a + (1 << 16)] + "\n")
    samples = ["S%d" % i for i in range(panel.n_samples)]
    with open(prefix + ".vcf", "w") as fh:
        fh.write("##fileformat=VCFv4.2\n##INFO=<ID=%s,Number=A,Type=Float,Description=\"af\">\n"
                 "##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n" % freq_key)
        for name, l in zip(panel.contig_names, panel.contig_len):
            fh.write("##contig=<ID=%s,length=%d>\n" % (name, int(l)))
        fh.write("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(samples) + "\n")
        for v in range(panel.n):
            a0, a1 = int(panel.var_allele_off[v]), int(panel.var_allele_off[v + 1])
            alleles = [panel.allele(v, a) .decode() for a in range(a1 - a0)]
            g = panel.gt[v]
            gts = ["%d%s%d" % (int(x) & 127, "|" if (int(x) >> 14) & 1 else "/", (int(x) >> 7) & 127) for x in g]
            af = ",".join(repr(float(np.float32(f))) for f in panel.freq[a0 + 1:a1])
            fh.write("%s\t%d\t.\t%s\t%s\t.\t.\t%s=%s\tGT\t%s\n" % (panel.contig_names[int(panel.contig_id[v])], int(panel.pos[v]) + 1, alleles[0],
                                                               ",".join(alleles[1:]), freq_key, af, "\t".join(gts)))


def device_table_flat(panel: FlatPanel, n_rows, k, ref_k, seed, device, plant_records=None, offsets=(-2, -1, 0, 1, 2), chunk=2_000_000):
    """kmer table for a FlatPanel drawn ON THE GPU (torch: random bits, gathers and a scatter): uniform random ref_k-mers with
    the donor's windows around the first `plant_records` records planted at distinct random places.  SNP-only panels
    (clustered_snp_panel) get their windows from the two donor haplotypes held on the device -- centred windows at `offsets`
    around every record, neighbours' donor alleles included; panels with indels take donor_rows() on the host.  Rows are
    not canonicalised (the scan does that).  -> dict: d_hi, d_lo (int64 bit patterns), d_cnt (int32), n, n_site"""
    import torch
    dev = torch.device("cuda", device) if isinstance(device, int) else device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    n_plant = panel.n if plant_records is None else min(panel.n, int(plant_records))

    def bits32(n):
        return torch.randint(0, 1 << 32, (n,), dtype=torch.int64, device=dev, generator=g)
    top = 2 * ref_k - 64
    d_lo = (bits32(n_rows) << 32) | bits32(n_rows)
    if top >= 64:
        d_hi = (bits32(n_rows) << 32) | bits32(n_rows)
    elif top > 0:
        d_hi = ((bits32(n_rows) << 32) | bits32(n_rows)) & ((1 << top) - 1)
    else:
        d_hi = torch.zeros(n_rows, dtype=torch.int64, device=dev)
        if top < 0:
            d_lo = d_lo & ((1 << (2 * ref_k)) - 1)
    d_cnt = torch.randint(2, 64, (n_rows,), dtype=torch.int32, device=dev, generator=g)
    snp_only = bool((panel.ref_size[:n_plant] == 1).all()) and bool((np.diff(panel.allele_off[:int(panel.var_allele_off[n_plant]) + 1].astype(np.int64)) == 1).all())
    his, los = [], []
    if n_plant and snp_only:
        code_t = torch.from_numpy(CODE.astype(np.int64)).to(dev)
        gpos = panel.gpos()[:n_plant]
        centre = (ref_k - k) // 2 + k // 2
        genome_t = torch.from_numpy(panel.genome).to(dev)
        cb = panel.contig_base[panel.contig_id[:n_plant]].astype(np.int64)
        ce = cb + panel.contig_len[panel.contig_id[:n_plant]].astype(np.int64)
        gpos_t = torch.from_numpy(gpos).to(dev)
        haps = []
        for h in range(2):
            hap = genome_t.clone()
            slot = panel.var_allele_off[:n_plant].astype(np.int64) + panel.donor_gt[:n_plant, h].astype(np.int64)
            hap[gpos_t] = torch.from_numpy(panel.pool[panel.allele_off[slot]]).to(dev)
            haps.append(hap)
        del genome_t

        def pack(w):
            hi = torch.zeros(w.shape[0], dtype=torch.int64, device=dev)
            lo = torch.zeros(w.shape[0], dtype=torch.int64, device=dev)
            for i in range(ref_k):
                sh = 2 * (ref_k - 1 - i)
                if sh >= 64:
                    hi |= w[:, i] << (sh - 64)
                else:
                    lo |= w[:, i] << sh
            return hi, lo
        for d in offsets:
            start = gpos - centre + d
            keep = (start >= cb) & (start + ref_k <= ce)
            st = torch.from_numpy(start[keep]).to(dev)
            for a in range(0, st.numel(), chunk):
                idx = st[a:a + chunk][:, None] + torch.arange(ref_k, device=dev)[None, :]
                w0 = code_t[haps[0][idx].to(torch.int64)]                  # [m, ref_k] codes 0..3 (255: non-ACGT)
                w1 = code_t[haps[1][idx].to(torch.int64)]
                g0, g1 = (w0 <= 3).all(dim=1), (w1 <= 3).all(dim=1)
                hi0, lo0 = pack(w0)
                hi1, lo1 = pack(w1)
                g1 &= (hi1 != hi0) | (lo1 != lo0) | ~g0                      # the second haplotype's window only where it differs (KMC lists distinct
                his += [hi0[g0], hi1[g1]]                                    # k-mers; windows shared by neighbouring records of a cluster stay doubled:
                los += [lo0[g0], lo1[g1]]                                    # torch.unique(dim=0) returns garbage at 1e8 rows on this ROCm build)
        del haps
        hi_t, lo_t = torch.cat(his), torch.cat(los)
    elif n_plant:
        hi, lo = donor_rows(panel, ref_k, n_plant)
        hi_t, lo_t = torch.from_numpy(hi.view(np.int64)).to(dev), torch.from_numpy(lo.view(np.int64)).to(dev)
    else:
        hi_t = lo_t = torch.zeros(0, dtype=torch.int64, device=dev)
    n_site = int(hi_t.numel())
    if n_site > n_rows:
        sel = torch.randperm(n_site, device=dev, generator=g)[:n_rows]
        hi_t, lo_t, n_site = hi_t[sel], lo_t[sel], n_rows
    if n_site:
        stride = n_rows // n_site
        where = torch.arange(n_site, dtype=torch.int64, device=dev) * stride
        if stride > 1:
            where += torch.randint(0, stride, (n_site,), dtype=torch.int64, device=dev, generator=g)
        where = where[torch.randperm(n_site, device=dev, generator=g)] if n_site < (1 << 31) else where
        d_hi[where] = hi_t
        d_lo[where] = lo_t
    torch.cuda.synchronize()
    return {"d_hi": d_hi, "d_lo": d_lo, "d_cnt": d_cnt, "n": n_rows, "n_site": n_site}
