"""Python restatement of the reference's record model and variant-block logic.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  Pure-Python loops: meant for
the small parity cases and for the reference's haploid example, not for speed.

Restates (paths relative to the reference checkout):
  Variant                 variant.hpp:43-253 (fields, frequencies, genotypes)
  VCF text decode         what variant.hpp:66-211 asks of htslib, for text VCF
  VB.are_overlapping/near var_block.hpp:408-423
  VB.chains_right/left    var_block.hpp:436-525, 534-624
  VB.combine              var_block.hpp:630-677
  VB.ref_subs             var_block.hpp:682-702
  VB.allele_combs         var_block.hpp:709-786
  VB.extract_kmers        var_block.hpp:95-219
"""
import gzip
import itertools
import math
import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple


def _F32(x) -> float:
    """x rounded to IEEE binary32 (round to nearest even), as a C cast to float rounds an int or a double"""
    return struct.unpack("f", struct.pack("f", float(x)))[0]


def f32(x: float) -> float:
    """round a Python float to IEEE binary32 and back (the reference stores `float`)."""
    return struct.unpack("<f", struct.pack("<f", x))[0]


def substr(s: str, pos: int, n: int) -> str:
    """std::string(s, pos, n): clips at the end, throws if pos > size."""
    if pos < 0 or pos > len(s):
        raise IndexError("std::out_of_range: substr pos %d size %d" % (pos, len(s)))
    return s[pos:pos + max(n, 0)] if n >= 0 else s[pos:]


@dataclass
class Variant:
    seq_name: str = ""
    ref_pos: int = 0                      # 0-based
    idx: str = "."
    ref_sub: str = ""
    alts: List[str] = field(default_factory=list)
    quality: float = float("nan")
    filter: str = "PASS"
    info: str = "."
    genotypes: List[Tuple[int, int]] = field(default_factory=list)
    phasing: List[bool] = field(default_factory=list)
    ref_size: int = 0
    min_size: int = 0
    max_size: int = 0
    has_alts: bool = True
    is_present: bool = True
    frequencies: List[float] = field(default_factory=list)   # float32 values
    coverages: List[int] = field(default_factory=list)
    computed_gts: List[Tuple[str, float]] = field(default_factory=list)

    def set_sizes(self):  # variant.hpp:108-124
        if not self.alts:
            self.has_alts = False
        else:
            sizes = [self.ref_size] + [len(a) for a in self.alts]
            self.min_size, self.max_size = min(sizes), max(sizes)

    def get_allele(self, i: int) -> str:  # variant.hpp:216-222
        if i == 0:
            return self.ref_sub
        if i - 1 >= len(self.alts):
            raise IndexError("GT allele %d beyond the kept ALT list (the reference reads out of bounds here)" % i)
        return self.alts[i - 1]

    def get_allele_index(self, a: str) -> int:  # variant.hpp:228-240
        if self.ref_sub == a:
            return 0
        for i, alt in enumerate(self.alts, 1):
            if alt == a:
                return i
        return -1


def set_frequencies(v: Variant, raw_freqs: Optional[List[float]], uniform: bool):
    """variant.hpp:126-156.  raw_freqs = INFO values of the frequency key as floats."""
    if not uniform:
        if raw_freqs is None:
            raise KeyError("frequency key missing from INFO (the reference dereferences NULL here)")
        if len(raw_freqs) < len(v.alts):
            raise IndexError("fewer frequency values than ALT alleles (the reference reads out of bounds here)")
        fr = [0.0] + [f32(raw_freqs[i]) for i in range(len(v.alts))]
        acc = 0.0
        for x in fr:           # accumulate(..., 0.0): double
            acc += x
        fr[0] = f32(1.0 - acc)
        if fr[0] < 0:
            fr[0] = 0.0
        v.frequencies = fr
    else:
        u = f32(1.0 / (len(v.alts) + 1))
        v.frequencies = [u] * (len(v.alts) + 1)
    if v.frequencies[0] == 1.0:
        v.is_present = False


# ---------------------------------------------------------------------------
# Text VCF / FASTA readers: only what variant.hpp:66-211 and main.cpp:190-219,
# 283-295 observe through htslib / kseq.
# ---------------------------------------------------------------------------

def _open_text(path):
    with open(path, "rb") as fh:
        magic = fh.read(2)
    if magic == b"\x1f\x8b":
        return gzip.open(path, "rt")
    return open(path, "rt")


def read_fasta(path, strip_chr=False) -> Dict[str, str]:
    """main.cpp:283-295: id = first word of the header, sequence upper-cased."""
    refs, name, chunks = {}, None, []
    with _open_text(path) as fh:
        for line in fh:
            line = line.rstrip("\n").rstrip("\r")
            if line.startswith(">"):
                if name is not None:
                    refs[name] = "".join(chunks).upper()
                name = line[1:].split()[0] if len(line) > 1 else ""
                if strip_chr and name.startswith("chr"):
                    name = name[3:]
                chunks = []
            elif name is not None:
                chunks.append(line.strip())
        if name is not None:
            refs[name] = "".join(chunks).upper()
    return refs


class VCFReader:
    """Header lines, sample subset, and a record iterator yielding Variant objects."""

    def __init__(self, path, samples="-"):
        self.path = path
        self.header_lines = []
        self.sample_names = []
        with _open_text(path) as fh:
            for line in fh:
                if line.startswith("##"):
                    self.header_lines.append(line.rstrip("\n"))
                elif line.startswith("#"):
                    cols = line.rstrip("\n").split("\t")
                    self.sample_names = cols[9:]
                    break
        if samples == "-":
            self.keep = list(range(len(self.sample_names)))
        else:
            with open(samples) as fh:
                wanted = {l.strip() for l in fh if l.strip()}
            missing = wanted - set(self.sample_names)
            if missing:
                raise ValueError("ERROR: VCF samples subset")
            self.keep = [i for i, s in enumerate(self.sample_names) if s in wanted]  # VCF order

    def records(self, freq_key="AF", uniform=False):
        with _open_text(self.path) as fh:
            for line in fh:
                if line.startswith("#"):
                    continue
                line = line.rstrip("\n")
                if not line:
                    continue
                yield self._variant(line.split("\t"), freq_key, uniform)

    def _variant(self, c, freq_key, uniform) -> Variant:
        v = Variant()
        v.seq_name = c[0]
        v.ref_pos = int(c[1]) - 1
        v.idx = c[2]
        v.ref_sub = c[3].upper()
        v.ref_size = len(v.ref_sub)
        alts = [] if c[4] == "." else c[4].split(",")
        v.alts = [a.upper() for a in alts if not a.startswith("<")]   # variant.hpp:79-88
        v.coverages = [0] * (len(v.alts) + 1)
        v.quality = float("nan") if c[5] == "." else f32(float(c[5]))
        v.set_sizes()
        if v.has_alts:
            raw = None
            if c[7] != ".":
                for kv in c[7].split(";"):
                    if kv.startswith(freq_key + "="):
                        raw = [float("nan") if x == "." else float(x) for x in kv[len(freq_key) + 1:].split(",")]
                        break
            set_frequencies(v, raw, uniform)
            if v.is_present:
                self._genotypes(v, c)
        return v

    def _genotypes(self, v: Variant, c):
        """variant.hpp:158-211 over what bcf_get_genotypes returns for text GT fields."""
        fmt = c[8].split(":") if len(c) > 8 else []
        if "GT" not in fmt or not self.keep:
            v.has_alts = False
            return
        gi = fmt.index("GT")
        parsed = []
        for s in self.keep:
            fields = c[9 + s].split(":")
            gt = fields[gi] if gi < len(fields) else "."
            hit = _GT_CACHE.get(gt)      # a panel repeats a handful of GT strings tens of thousands of times per record
            if hit is None:
                alleles, phased, cur, ph = [], [], "", False
                for ch in gt:
                    if ch in "/|":
                        alleles.append(cur); phased.append(ph)
                        cur, ph = "", (ch == "|")
                    else:
                        cur += ch
                alleles.append(cur); phased.append(ph)
                hit = _GT_CACHE[gt] = ([-1 if a in (".", "") else int(a) for a in alleles], phased)
            parsed.append(hit)
        ploidy = max(len(a) for a, _ in parsed)
        flat = []   # ploidy values per sample: (allele or None for vector_end, phased bit)
        for al, ph in parsed:
            for j in range(ploidy):
                flat.append((al[j], ph[j]) if j < len(al) else (None, False))
        n = len(parsed)
        for i in range(n):
            first = flat[i * ploidy]
            # curr_gt[1]: with ploidy 1 this is the NEXT sample's value (variant.hpp:184,203-205);
            # past the last sample the reference reads beyond the array -- restated as "end".
            nxt = flat[i * ploidy + 1] if i * ploidy + 1 < len(flat) else (None, False)
            if ploidy == 1 and i * ploidy + 1 >= len(flat):
                nxt = (None, False)
            if nxt[0] is None:
                a1 = a2 = first[0]
                is_ph = True
            else:
                a1, a2 = first[0], nxt[0]
                is_ph = bool(nxt[1])
            a1 = 0 if a1 is None or a1 < 0 else a1
            a2 = 0 if a2 is None or a2 < 0 else a2
            v.genotypes.append(_PAIRS.setdefault((a1, a2), (a1, a2)))   # one tuple object per distinct pair (27,934 samples a record)
            v.phasing.append(is_ph)


_GT_CACHE: Dict[str, tuple] = {}     # memoisation only: GT string -> (alleles, phased flags), never mutated
_PAIRS: Dict[tuple, tuple] = {}


# ---------------------------------------------------------------------------
# Variant block
# ---------------------------------------------------------------------------

VK_GROUP = Dict[int, Dict[int, List[List[str]]]]


class VB:
    def __init__(self, k: int, error_rate: float):
        self.k = k
        self.error_rate = f32(error_rate)
        self.variants: List[Variant] = []

    # var_block.hpp:408-412
    @staticmethod
    def are_overlapping(v1: Variant, v2: Variant) -> bool:
        return v1.ref_pos <= v2.ref_pos < v1.ref_pos + v1.ref_size

    # var_block.hpp:417-423.  `int + ... + ceil((float)k / 2) >= int` under `using namespace std`: ceil is the FLOAT
    # overload, so the int sum is converted to float, the addition rounds to float and v2.ref_pos is compared as a
    # float.  Exact below 2^24; beyond that positions are rounded to multiples of 2, 4, 8, 16 (most of a human
    # chromosome) and the answer differs from the exact one now and then, in both directions.
    def are_near(self, v1: Variant, v2: Variant, extra: int = 0) -> bool:
        lhs = _F32(_F32(v1.ref_pos + v1.ref_size - v1.min_size - 1 + extra) + _F32(math.ceil(self.k / 2)))
        return bool(lhs >= _F32(v2.ref_pos))

    def is_near_to_last(self, v: Variant) -> bool:  # var_block.hpp:77-80
        return self.are_near(self.variants[-1], v)

    def add_variant(self, v):
        self.variants.append(v)

    def empty(self):
        return not self.variants

    def clear(self):
        self.variants = []

    def _chains(self, i: int, step: int) -> List[List[int]]:
        """get_combs_on_the_right (step=+1) / _left (step=-1).

        The two reference functions are mirror images; (a, b) below is always
        ordered left-to-right on the genome, as the reference's argument order is.
        """
        V = self.variants
        mid = V[i]

        def ordered(x, y):           # (left, right) on the genome
            return (x, y) if step > 0 else (y, x)

        chains: List[List[int]] = []
        sums: List[int] = []
        j = i + step
        halt = False
        while 0 <= j < len(V) and not halt:
            cur = V[j]
            j_now, j = j, j + step
            if not cur.is_present:
                continue
            if self.are_overlapping(*ordered(mid, cur)):
                continue
            gain = cur.ref_size - cur.min_size
            if not chains:
                if self.are_near(*ordered(mid, cur)):
                    chains.append([j_now]); sums.append(gain)
                continue
            added = False
            for c in range(len(chains)):
                last = V[chains[c][-1]]
                if not self.are_overlapping(*ordered(last, cur)):
                    added = True
                    if self.are_near(*ordered(mid, cur), sums[c]):
                        chains[c].append(j_now); sums[c] += gain
            if not added:
                new_chains, new_sums = [], []
                for c in range(len(chains)):
                    nc, ns = list(chains[c]), sums[c]
                    # pop members that overlap cur; the reference indexes back() of an
                    # emptied vector here (UB) -- restated as "stop when empty".
                    while nc and self.are_overlapping(*ordered(V[nc[-1]], cur)):
                        m = V[nc.pop()]
                        ns -= m.ref_size - m.min_size
                    nc.append(j_now)
                    if self.are_near(*ordered(mid, cur), ns):
                        added = True
                        new_chains.append(nc); new_sums.append(ns + gain)
                chains.extend(new_chains); sums.extend(new_sums)
                if not added:
                    halt = True
        return chains

    def combine(self, left: List[List[int]], right: List[List[int]], i: int) -> List[List[int]]:
        """var_block.hpp:630-677"""
        if not left and not right:
            return [[i]]
        if not left:
            return [[i] + r for r in right]
        out = []
        for l in left:
            base = list(reversed(l)) + [i]
            if not right:
                out.append(base)
            else:
                out.extend(base + r for r in right)
        return out

    def ref_subs(self, comb: List[int], reference: str) -> List[str]:
        """var_block.hpp:682-702"""
        subs, last_end = [], -1
        for index in comb:
            v = self.variants[index]
            if last_end != -1:
                subs.append(substr(reference, last_end, v.ref_pos - last_end))
            last_end = v.ref_pos + v.ref_size
        return subs

    def allele_combs(self, comb: List[int], central: int, haploid: bool):
        """build_alleles_combs + combine_haplotypes, var_block.hpp:709-786 (a set: order-free)."""
        V = self.variants
        out = set()
        for gt_i in range(len(V[central].genotypes)):
            if haploid:
                out.add(tuple(V[j].get_allele(V[j].genotypes[gt_i][0]) for j in comb))
                continue
            phased = all(V[j].phasing[gt_i] for j in comb)
            hap1 = tuple(V[j].get_allele(V[j].genotypes[gt_i][0]) for j in comb)
            hap2 = tuple(V[j].get_allele(V[j].genotypes[gt_i][1]) for j in comb)
            if phased:
                out.add(hap1); out.add(hap2)
            else:
                # every pick of hap1[l] / hap2[l] per level (the 2N rows of combine_haplotypes)
                for pick in itertools.product(*zip(hap1, hap2)):
                    out.add(tuple(pick))
        return out

    def extract_kmers(self, reference: str, haploid: bool) -> VK_GROUP:
        """var_block.hpp:95-219"""
        return {vi: self.extract_one(vi, reference, haploid) for vi in range(len(self.variants))}

    def extract_one(self, vi: int, reference: str, haploid: bool) -> Dict[int, List[List[str]]]:
        """the body of extract_kmers' loop over the block's variants (var_block.hpp:100-216): the signatures of variant vi.
        Variants are enumerated independently of one another, which is what lets tests/gen_c1_golden.py spread a block of
        thousands of variants x tens of thousands of samples over processes without touching the restated logic."""
        k = self.k
        v = self.variants[vi]
        per_allele: Dict[int, List[List[str]]] = {}
        if (not v.is_present) or v.ref_pos < k or v.ref_pos > len(reference) - k:
            return per_allele
        combs = self.combine(self._chains(vi, -1), self._chains(vi, +1), vi)
        for comb in combs:
            rsubs = self.ref_subs(comb, reference)
            for aac in self.allele_combs(comb, vi, haploid):
                sig: List[str] = []
                if len(aac) == 1 and len(aac[0]) >= k:
                    mid_allele = aac[0]
                    sig = [mid_allele[p:p + k] for p in range(len(mid_allele) - k + 1)]
                else:
                    kmer, mid_pos, mid_allele = "", 0, ""
                    for j, allele in enumerate(aac):
                        if comb[j] == vi:
                            mid_pos, mid_allele = len(kmer), allele
                        kmer += allele + (rsubs[j] if j < len(rsubs) else "")
                    first_part = mid_pos + len(mid_allele) // 2
                    second_part = len(kmer) - first_part
                    missing_prefix = k // 2 - first_part
                    missing_suffix = math.ceil(k / 2) - second_part
                    if missing_prefix >= 0:
                        fv = self.variants[comb[0]]
                        kmer = substr(reference, fv.ref_pos - missing_prefix, missing_prefix) + kmer
                    else:
                        kmer = kmer[-missing_prefix:]
                    if missing_suffix >= 0:
                        lv = self.variants[comb[-1]]
                        kmer += substr(reference, lv.ref_pos + lv.ref_size, missing_suffix)
                    else:
                        if -missing_suffix > len(kmer):
                            raise IndexError("std::out_of_range in erase")
                        kmer = kmer[:len(kmer) + missing_suffix]
                    sig = [kmer]
                ai = v.get_allele_index(mid_allele)
                per_allele.setdefault(ai, []).append(sig)
        return per_allele


def flatten_vk(kmers: VK_GROUP, var_n_alleles: List[int]):
    """VK_GROUP of one block -> flat descriptors for the batched lookup/coverage
    functions: (kmer list, is_ref list, sig_kmer_off, allele_sig_off) with one
    allele slot per (variant, allele) in order."""
    ks, is_ref, sig_off, al_off = [], [], [0], [0]
    for vi, A in enumerate(var_n_alleles):
        per = kmers.get(vi, {})
        for a in range(A):
            for sig in per.get(a, []):
                for km in sig:
                    ks.append(km.encode()); is_ref.append(1 if a == 0 else 0)
                sig_off.append(len(ks))
            al_off.append(len(sig_off) - 1)
    return ks, is_ref, sig_off, al_off
