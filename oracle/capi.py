"""ctypes binding of the C oracle (oracle/malva_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under malva_amd/ may import this.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmalva_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libmalva_oracle.so missing: run `make oracle`")
        L = C.CDLL(path)
        vp, cp, u64, u32, i32, sz = C.c_void_p, C.c_char_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_size_t
        sig = {
            "mo_xxh3_64": (u64, [vp, sz]),
            "mo_canonical": (None, [cp, C.c_int, vp]),
            "mo_are_near": (C.c_int, [C.c_int] * 6),
            "mo_bf_new": (vp, [u64]),
            "mo_bf_free": (None, [vp]),
            "mo_bf_hash": (u64, [cp]),
            "mo_bf_add_key": (None, [vp, cp]),
            "mo_bf_test_key": (C.c_int, [vp, cp]),
            "mo_bf_switch_mode": (None, [vp]),
            "mo_bf_increment": (C.c_int, [vp, cp, u32]),
            "mo_bf_get_count": (C.c_uint16, [vp, cp]),
            "mo_bf_size": (u64, [vp]),
            "mo_bf_nwords": (u64, [vp]),
            "mo_bf_words": (vp, [vp]),
            "mo_bf_nset": (u64, [vp]),
            "mo_bf_counts": (vp, [vp]),
            "mo_bf_popcount": (u64, [vp]),
            "mo_bf_set_positions": (u64, [vp, vp, u64]),
            "mo_bf_load_words": (None, [vp, vp]),
            "mo_kmap_new": (vp, []),
            "mo_kmap_free": (None, [vp]),
            "mo_kmap_add_key": (None, [vp, cp]),
            "mo_kmap_test_key": (C.c_int, [vp, cp]),
            "mo_kmap_increment": (None, [vp, cp, C.c_int]),
            "mo_kmap_get_count": (C.c_int, [vp, cp]),
            "mo_kmap_size": (u64, [vp]),
            "mo_kmap_entry": (vp, [vp, u64, vp, vp]),
            "mo_add_kmers": (None, [vp, vp, vp, sz, sz, vp]),
            "mo_kmc_scan": (None, [vp, vp, vp, vp, sz, vp, sz, C.c_int, C.c_int]),
            "mo_kmc_scan_packed": (None, [vp, vp, vp, vp, vp, vp, sz, C.c_int, C.c_int]),
            "mo_kmc_scan_packed_mt": (C.c_int, [vp, vp, vp, vp, vp, vp, sz, C.c_int, C.c_int, C.c_int]),
            "mo_ref_scan": (C.c_int, [vp, vp, vp, sz, C.c_int, C.c_int]),
            "mo_lookup_weights": (None, [vp, vp, vp, sz, sz, vp, vp]),
            "mo_set_coverages": (None, [vp, vp, vp, u64, vp]),
            "mo_genotype": (C.c_int, [vp, vp, C.c_int, C.c_float, C.c_int, C.c_int, vp, vp, vp, C.c_int]),
            "mo_select_gt": (C.c_int, [vp, C.c_int, vp, vp]),
            "mo_call_isolated": (None, [vp, vp, vp, sz, sz, vp, vp, vp, vp, vp, vp, vp, C.c_int, C.c_float,
                                        C.c_int, C.c_int, vp, vp, vp, vp]),
            "mo_allele_canon": (None, [sz, vp, vp, vp, vp]),
            "mo_cut_blocks": (sz, [sz, vp, vp, vp, vp, C.c_int, vp, vp]),
            "mo_cover_blocks": (C.c_int64, [vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, C.c_int, C.c_int, vp, vp]),
            "mo_index_blocks": (C.c_int64, [vp, vp, vp, sz, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, u32, C.c_int, C.c_int, vp]),
            "mo_genotype_panel": (None, [vp, vp, vp, sz, C.c_float, C.c_int, C.c_int, vp, vp, vp]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def xxh3_64(data: bytes) -> int:
    return lib().mo_xxh3_64(data, len(data))


def canonical(kmer: bytes) -> bytes:
    """k bytes of BF::_canonical's buffer (may contain NULs)."""
    buf = C.create_string_buffer(len(kmer) + 1)
    lib().mo_canonical(kmer, len(kmer), buf)
    return buf.raw[: len(kmer)]


def are_near(v1_ref_pos: int, v1_ref_size: int, v1_min_size: int, sum_to_add: int, k: int, v2_ref_pos: int) -> bool:
    """VB::are_near in the reference's float arithmetic (var_block.hpp:417-423)"""
    return bool(lib().mo_are_near(v1_ref_pos, v1_ref_size, v1_min_size, sum_to_add, k, v2_ref_pos))


def rows_from_kmers(kmers, stride=None):
    """list of bytes -> (uint8 array [n, stride] NUL padded, stride)."""
    n = len(kmers)
    if stride is None:
        stride = (max((len(k) for k in kmers), default=0) + 1 + 7) // 8 * 8
    arr = np.zeros((n, stride), dtype=np.uint8)
    for i, k in enumerate(kmers):
        if len(k) >= stride:
            raise ValueError("k-mer longer than row stride")
        arr[i, : len(k)] = np.frombuffer(k, dtype=np.uint8)
    return arr, stride


class BF:
    """bloom_filter.hpp:52-157"""

    def __init__(self, size_bits):
        self.h = lib().mo_bf_new(size_bits)
        if not self.h:
            raise MemoryError("mo_bf_new")

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().mo_bf_free(self.h)
            except TypeError:          # interpreter shutdown: the module globals are already gone
                pass
            self.h = None

    def add_key(self, kmer: bytes):
        lib().mo_bf_add_key(self.h, kmer)

    def test_key(self, kmer: bytes) -> bool:
        return bool(lib().mo_bf_test_key(self.h, kmer))

    def switch_mode(self):
        lib().mo_bf_switch_mode(self.h)

    def increment(self, kmer: bytes, counter: int) -> bool:
        return bool(lib().mo_bf_increment(self.h, kmer, counter))

    def get_count(self, kmer: bytes) -> int:
        return lib().mo_bf_get_count(self.h, kmer)

    @property
    def size(self):
        return lib().mo_bf_size(self.h)

    @property
    def nset(self):
        return lib().mo_bf_nset(self.h)

    def popcount(self):
        return lib().mo_bf_popcount(self.h)

    def words(self):
        n = lib().mo_bf_nwords(self.h)
        return np.ctypeslib.as_array(C.cast(lib().mo_bf_words(self.h), C.POINTER(C.c_uint64)), shape=(n,))

    def counts(self):
        n = self.nset
        if n == 0:
            return np.zeros(0, dtype=np.uint16)
        return np.ctypeslib.as_array(C.cast(lib().mo_bf_counts(self.h), C.POINTER(C.c_uint16)), shape=(n,))

    def load_words(self, words):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        assert words.shape[0] == lib().mo_bf_nwords(self.h)
        lib().mo_bf_load_words(self.h, _p(words))

    def set_positions(self):
        n = self.popcount()
        out = np.zeros(n, dtype=np.uint64)
        lib().mo_bf_set_positions(self.h, _p(out), n)
        return out


class KMAP:
    """kmap.hpp:46-132"""

    def __init__(self):
        self.h = lib().mo_kmap_new()

    def __del__(self):
        if getattr(self, "h", None):
            try:
                lib().mo_kmap_free(self.h)
            except TypeError:          # interpreter shutdown
                pass
            self.h = None

    def add_key(self, kmer: bytes):
        lib().mo_kmap_add_key(self.h, kmer)

    def test_key(self, kmer: bytes) -> bool:
        return bool(lib().mo_kmap_test_key(self.h, kmer))

    def increment(self, kmer: bytes, counter: int):
        lib().mo_kmap_increment(self.h, kmer, counter)

    def get_count(self, kmer: bytes) -> int:
        return lib().mo_kmap_get_count(self.h, kmer)

    def __len__(self):
        return lib().mo_kmap_size(self.h)

    def items(self):
        ln, val = C.c_uint32(), C.c_int32()
        for e in range(len(self)):
            p = lib().mo_kmap_entry(self.h, e, C.byref(ln), C.byref(val))
            yield C.string_at(p, ln.value), val.value


def add_kmers(bf: BF, ref_bf: KMAP, rows, is_ref):
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    is_ref = np.ascontiguousarray(is_ref, dtype=np.uint8)
    lib().mo_add_kmers(bf.h, ref_bf.h, _p(rows), rows.shape[1], rows.shape[0], _p(is_ref))


def kmc_scan(context_bf: BF, bf: BF, ref_bf: KMAP, rows, counts, k, ref_k):
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    lib().mo_kmc_scan(context_bf.h, bf.h, ref_bf.h, _p(rows), rows.shape[1], _p(counts), rows.shape[0], k, ref_k)


def kmc_scan_packed(context_bf: BF, bf: BF, ref_bf: KMAP, hi, lo, counts, k, ref_k):
    hi = np.ascontiguousarray(hi, dtype=np.uint64)
    lo = np.ascontiguousarray(lo, dtype=np.uint64)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    lib().mo_kmc_scan_packed(context_bf.h, bf.h, ref_bf.h, _p(hi), _p(lo), _p(counts), hi.shape[0], k, ref_k)


def kmc_scan_packed_mt(context_bf: BF, bf: BF, ref_bf: KMAP, hi, lo, counts, k, ref_k, n_threads):
    """the same scan by n_threads threads with atomic (commuting) counter updates: identical results"""
    hi = np.ascontiguousarray(hi, dtype=np.uint64)
    lo = np.ascontiguousarray(lo, dtype=np.uint64)
    counts = np.ascontiguousarray(counts, dtype=np.uint32)
    return lib().mo_kmc_scan_packed_mt(context_bf.h, bf.h, ref_bf.h, _p(hi), _p(lo), _p(counts), hi.shape[0], k, ref_k, n_threads)


def ref_scan(bf: BF, context_bf: BF, reference: bytes, k, ref_k):
    rc = lib().mo_ref_scan(bf.h, context_bf.h, reference, len(reference), k, ref_k)
    if rc != 0:
        raise ValueError("contig shorter than (ref_k-k)/2: the reference would throw here")


def lookup_weights(bf: BF, ref_bf: KMAP, rows, is_ref):
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    is_ref = np.ascontiguousarray(is_ref, dtype=np.uint8)
    w = np.zeros(rows.shape[0], dtype=np.int32)
    lib().mo_lookup_weights(bf.h, ref_bf.h, _p(rows), rows.shape[1], rows.shape[0], _p(is_ref), _p(w))
    return w


def set_coverages(w, sig_kmer_off, allele_sig_off):
    w = np.ascontiguousarray(w, dtype=np.int32)
    so = np.ascontiguousarray(sig_kmer_off, dtype=np.uint64)
    ao = np.ascontiguousarray(allele_sig_off, dtype=np.uint64)
    cov = np.zeros(len(ao) - 1, dtype=np.uint32)
    lib().mo_set_coverages(_p(w), _p(so), _p(ao), len(ao) - 1, _p(cov))
    return cov


def genotype(cov, freq, error_rate, max_cov, haploid):
    """-> list of (g1, g2, value); g2 == -1 in haploid mode."""
    cov = np.ascontiguousarray(cov, dtype=np.uint32)
    freq = np.ascontiguousarray(freq, dtype=np.float32)
    A = len(cov)
    cap = A * (A + 1) // 2 + A + 2
    g1 = np.zeros(cap, dtype=np.int32)
    g2 = np.zeros(cap, dtype=np.int32)
    vals = np.zeros(cap, dtype=np.float64)
    n = lib().mo_genotype(_p(cov), _p(freq), A, C.c_float(error_rate), max_cov, int(haploid), _p(g1), _p(g2),
                          _p(vals), cap)
    assert n >= 0
    return [(int(g1[i]), int(g2[i]), float(vals[i])) for i in range(n)]


def select_gt(vals):
    """-> (index or -1, GQ, normalised list)"""
    vals = np.ascontiguousarray(vals, dtype=np.float64)
    norm = np.zeros(len(vals), dtype=np.float64)
    gq = C.c_int()
    bi = lib().mo_select_gt(_p(vals), len(vals), _p(norm), C.byref(gq))
    return bi, gq.value, norm


def call_isolated(bf: BF, ref_bf: KMAP, reference: bytes, pos, allele_off, var_allele_off, allele_pool, freq,
                  present_mask, is_present, k, error_rate, max_cov, haploid):
    pos = np.ascontiguousarray(pos, dtype=np.int64)
    allele_off = np.ascontiguousarray(allele_off, dtype=np.uint32)
    var_allele_off = np.ascontiguousarray(var_allele_off, dtype=np.uint32)
    freq = np.ascontiguousarray(freq, dtype=np.float32)
    present_mask = np.ascontiguousarray(present_mask, dtype=np.uint64)
    is_present = np.ascontiguousarray(is_present, dtype=np.uint8)
    n = len(pos)
    na = int(var_allele_off[-1])
    cov = np.zeros(na, dtype=np.uint32)
    g1 = np.zeros(n, dtype=np.int32)
    g2 = np.zeros(n, dtype=np.int32)
    gq = np.zeros(n, dtype=np.int32)
    pool = np.frombuffer(allele_pool, dtype=np.uint8) if isinstance(allele_pool, (bytes, bytearray)) else allele_pool
    pool = np.ascontiguousarray(pool, dtype=np.uint8)
    refarr = np.frombuffer(reference, dtype=np.uint8) if isinstance(reference, (bytes, bytearray)) else reference
    lib().mo_call_isolated(bf.h, ref_bf.h, _p(refarr), len(refarr), n, _p(pos), _p(allele_off), _p(var_allele_off),
                           _p(pool), _p(freq), _p(present_mask), _p(is_present), k, C.c_float(error_rate), max_cov,
                           int(haploid), _p(cov), _p(g1), _p(g2), _p(gq))
    return cov, g1, g2, gq


# ---- variant blocks over a flat panel (the arrays mg_cover_blocks / mg_index_blocks take) ---------------------------

def _arr(a, dt):
    return np.ascontiguousarray(a, dtype=dt)


def allele_canon(var_allele_off, allele_off, pool):
    """first allele of the variant with the same text, per allele slot (Variant::get_allele_index, variant.hpp:228-240)"""
    vo, ao, pool = _arr(var_allele_off, np.uint32), _arr(allele_off, np.uint32), _arr(pool, np.uint8)
    canon = np.zeros(int(vo[-1]), dtype=np.uint8)
    lib().mo_allele_canon(len(vo) - 1, _p(vo), _p(ao), _p(pool), _p(canon))
    return canon


def cut_blocks(pos, ref_size, min_size, contig_id, k):
    """the record loops' block cut (main.cpp:341, 547) -> (blk_var_off [n_blocks + 1], blk_contig [n_blocks])"""
    pos, rs, ms, cid = _arr(pos, np.int32), _arr(ref_size, np.uint32), _arr(min_size, np.uint32), _arr(contig_id, np.uint32)
    n = len(pos)
    off = np.zeros(n + 1, dtype=np.uint32)
    bc = np.zeros(max(n, 1), dtype=np.uint32)
    nb = lib().mo_cut_blocks(n, _p(pos), _p(rs), _p(ms), _p(cid), k, _p(off), _p(bc))
    return off[:nb + 1].copy(), bc[:nb].copy()


def _block_args(reference, blk_ref_base, blk_ref_len, blk_var_off, pos, ref_size, min_size, present, var_allele_off, allele_off, pool, canon, gt,
                n_samples):
    ref = np.frombuffer(reference, dtype=np.uint8) if isinstance(reference, (bytes, bytearray)) else _arr(reference, np.uint8)
    a = dict(ref=ref, bb=_arr(blk_ref_base, np.uint64), bl=_arr(blk_ref_len, np.uint32), bo=_arr(blk_var_off, np.uint32), pos=_arr(pos, np.int32),
             rs=_arr(ref_size, np.uint32), ms=_arr(min_size, np.uint32), pr=_arr(present, np.uint8), vo=_arr(var_allele_off, np.uint32),
             ao=_arr(allele_off, np.uint32), pool=_arr(pool, np.uint8), canon=_arr(canon, np.uint8), gt=_arr(gt, np.uint16).reshape(-1))
    assert a["gt"].size >= len(a["pos"]) * n_samples
    return a


def cover_blocks(bf: BF, ref_bf: KMAP, reference, blk_ref_base, blk_ref_len, blk_var_off, pos, ref_size, min_size, present, var_allele_off,
                 allele_off, pool, canon, gt, n_samples, haploid, k, stats=None):
    """extract_kmers + set_coverages (main.cpp:556-557) for a batch of blocks -> coverage per allele slot"""
    a = _block_args(reference, blk_ref_base, blk_ref_len, blk_var_off, pos, ref_size, min_size, present, var_allele_off, allele_off, pool, canon, gt,
                    n_samples)
    cov = np.zeros(int(a["vo"][-1]), dtype=np.uint32)
    st = np.zeros(2, dtype=np.uint64)
    rc = lib().mo_cover_blocks(bf.h, ref_bf.h, _p(a["ref"]), len(a["bb"]), _p(a["bb"]), _p(a["bl"]), _p(a["bo"]), _p(a["pos"]), _p(a["rs"]), _p(a["ms"]),
                               _p(a["pr"]), _p(a["vo"]), _p(a["ao"]), _p(a["pool"]), _p(a["canon"]), _p(a["gt"]), n_samples, int(haploid), k, _p(cov), _p(st))
    if rc != 0:
        raise IndexError("std::out_of_range in block %d (the reference throws here)" % (-rc - 1))
    if stats is not None:
        stats["kmers"], stats["signatures"] = int(st[0]), int(st[1])
    return cov


def index_blocks(bf: BF, ref_bf: KMAP, reference, blk_ref_base, blk_ref_len, blk_var_off, pos, ref_size, min_size, present, var_allele_off,
                 allele_off, pool, canon, gt, n_samples, haploid, k):
    """extract_kmers + add_kmers_to_bf (main.cpp:349-350) for a batch of blocks; returns the k-mers added"""
    a = _block_args(reference, blk_ref_base, blk_ref_len, blk_var_off, pos, ref_size, min_size, present, var_allele_off, allele_off, pool, canon, gt,
                    n_samples)
    st = np.zeros(2, dtype=np.uint64)
    rc = lib().mo_index_blocks(bf.h, ref_bf.h, _p(a["ref"]), len(a["bb"]), _p(a["bb"]), _p(a["bl"]), _p(a["bo"]), _p(a["pos"]), _p(a["rs"]), _p(a["ms"]),
                               _p(a["pr"]), _p(a["vo"]), _p(a["ao"]), _p(a["pool"]), _p(a["canon"]), _p(a["gt"]), n_samples, int(haploid), k, _p(st))
    if rc != 0:
        raise IndexError("std::out_of_range in block %d (the reference throws here)" % (-rc - 1))
    return int(st[0])


def genotype_panel(cov, freq, var_allele_off, error_rate, max_cov, haploid):
    """VB::genotype + normalise / first-strict-max / GQ for every variant -> (gt1, gt2, gq); gt2 = -1 in haploid mode"""
    cov, freq, vo = _arr(cov, np.uint32), _arr(freq, np.float32), _arr(var_allele_off, np.uint32)
    n = len(vo) - 1
    g1, g2, gq = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros(n, np.int32)
    lib().mo_genotype_panel(_p(cov), _p(freq), _p(vo), n, C.c_float(error_rate), max_cov, int(haploid), _p(g1), _p(g2), _p(gq))
    return g1, g2, gq
