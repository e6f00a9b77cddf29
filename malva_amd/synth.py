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
