"""KMC database files (<db>.kmc_pre / <db>.kmc_suf), restated from KMC's published layout (KMC >= 2, the
"0x200" format; KMC API kmc_file.cpp: CKMCFile::ReadParamsFrom_prefix_file_buf / ReadNextKmer).  TEST INFRASTRUCTURE:
the reference reads these through the third-party KMC API (main.cpp:39, 444-449, 482-490; `-lkmc`, not vendored and
absent from this image) and its checkout holds no database fixture, so this restatement is PARITY UNPINNED: it pins
the product's reader (malva_amd/host/kmc_db.hpp + kmc_decode_kernel) against an independent reading of the same
published layout, not against a file KMC itself wrote.

<db>.kmc_pre
    "KMCP"
    u64 lut[n_bins * 4^lut_prefix_len]   per bin and prefix value: index of the first record carrying it; records of
                                         all bins are numbered through in file order (CKMCFile keeps ONE running
                                         record number while listing and compares it with consecutive table entries)
    u32 signature_map[4^signature_len + 1]   signature -> bin (only used for random access; ignored when listing)
    header, `header_offset` bytes:
        u32 kmer_length, u32 mode (0 = counts), u32 counter_size, u32 lut_prefix_length, u32 signature_len,
        u32 min_count, u32 max_count (low half), u64 total_kmers, u8 !both_strands, 3 x u8 0, u32 max_count (high
        half), zero padding, u32 kmc_version = 0x200          <- the version field ends the header
    u32 header_offset
    "KMCP"
<db>.kmc_suf
    "KMCS", total_kmers records of (kmer_length - lut_prefix_length) / 4 suffix bytes (first symbol in the top two bits
    of the first byte, A=0 C=1 G=2 T=3) + counter_size counter bytes (little-endian), "KMCS"

Listing order = file order: bin after bin, inside a bin ascending by (prefix, suffix)."""
import struct

import numpy as np

CODE = {ord("A"): 0, ord("C"): 1, ord("G"): 2, ord("T"): 3}
SYM = b"ACGT"
HEADER_BYTES = 68      # 7 x u32 + u64 + flag + 27 bytes kept for later use (max_count's high half sits in them) + u32 version


def _kmer_int(kmer: bytes) -> int:
    v = 0
    for ch in kmer:
        v = (v << 2) | CODE[ch]
    return v


def write_db(prefix_path, kmers, k, lut_prefix_len=None, n_bins=3, signature_len=5, counter_size=1, min_count=2, max_count=255,
             both_strands=True):
    """kmers: iterable of (bytes of length k over ACGT, count).  K-mers are dealt to `n_bins` bins by a hash of the k-mer
    (KMC deals them by minimiser signature: any partition gives a valid database for listing) and sorted inside a bin."""
    if lut_prefix_len is None:
        lut_prefix_len = next(p for p in (7, 6, 5, 4, 3, 2, 1) if (k - p) % 4 == 0)
    assert (k - lut_prefix_len) % 4 == 0 and 1 <= lut_prefix_len < k
    suffix_syms = k - lut_prefix_len
    suffix_bytes = suffix_syms // 4
    single = 1 << (2 * lut_prefix_len)
    bins = [[] for _ in range(n_bins)]
    for kmer, count in kmers:
        assert len(kmer) == k
        v = _kmer_int(kmer)
        bins[(v * 0x9E3779B97F4A7C15 >> 40) % n_bins].append((v, int(count)))
    lut = np.zeros(n_bins * single, dtype=np.uint64)
    body = bytearray()
    rec_no = 0
    for b, items in enumerate(bins):
        items.sort()
        starts = np.zeros(single + 1, dtype=np.int64)
        for v, _ in items:
            starts[(v >> (2 * suffix_syms)) + 1] += 1
        lut[b * single:(b + 1) * single] = rec_no + np.cumsum(starts)[:-1]
        for v, count in items:
            body += (v & ((1 << (2 * suffix_syms)) - 1)).to_bytes(suffix_bytes, "big")
            body += min(int(count), 256 ** counter_size - 1).to_bytes(counter_size, "little")   # a counter saturates at its width
        rec_no += len(items)
    with open(prefix_path + ".kmc_suf", "wb") as fh:
        fh.write(b"KMCS" + bytes(body) + b"KMCS")
    header = struct.pack("<7IQB3xI", k, 0, counter_size, lut_prefix_len, signature_len, min_count, max_count & 0xFFFFFFFF, rec_no,
                         0 if both_strands else 1, max_count >> 32)
    header += b"\0" * (HEADER_BYTES - 4 - len(header)) + struct.pack("<I", 0x200)
    with open(prefix_path + ".kmc_pre", "wb") as fh:
        fh.write(b"KMCP" + lut.tobytes() + np.zeros((1 << (2 * signature_len)) + 1, dtype=np.uint32).tobytes() + header +
                 struct.pack("<I", len(header)) + b"KMCP")
    return rec_no


class KmcDb:
    """Listing-mode reader: params, the prefix table and the raw records (numpy, no per-k-mer Python)."""

    def __init__(self, prefix_path):
        pre = open(prefix_path + ".kmc_pre", "rb").read()
        if pre[:4] != b"KMCP" or pre[-4:] != b"KMCP":
            raise ValueError("not a KMC prefix file")
        (version,) = struct.unpack_from("<I", pre, len(pre) - 12)
        if version != 0x200:
            raise ValueError("KMC database version 0x%x (only the KMC 2/3 format 0x200 is read)" % version)
        header_offset = pre[len(pre) - 8]                      # CKMCFile reads ONE byte here (fgetc)
        h0 = len(pre) - 8 - header_offset
        (self.k, self.mode, self.counter_size, self.lut_prefix_len, self.signature_len, self.min_count, max_lo, self.total,
         not_both, max_hi) = struct.unpack_from("<7IQB3xI", pre, h0)
        self.max_count = max_lo | (max_hi << 32)
        self.both_strands = not not_both
        if self.mode != 0:
            raise ValueError("KMC database in Quake mode (float counters)")
        sig_map_bytes = 4 * ((1 << (2 * self.signature_len)) + 1)
        lut_bytes = h0 - 4 - sig_map_bytes
        single = 1 << (2 * self.lut_prefix_len)
        if lut_bytes <= 0 or lut_bytes % (8 * single):
            raise ValueError("prefix table of %d bytes is not a whole number of 4^%d-entry bins" % (lut_bytes, self.lut_prefix_len))
        self.lut = np.frombuffer(pre, dtype="<u8", count=lut_bytes // 8, offset=4).copy()
        self.suffix_bytes = (self.k - self.lut_prefix_len) // 4
        if self.lut_prefix_len + 4 * self.suffix_bytes != self.k:
            raise ValueError("k - lut_prefix_len is not a multiple of 4")
        self.rec = self.suffix_bytes + self.counter_size
        suf = np.fromfile(prefix_path + ".kmc_suf", dtype=np.uint8)
        if bytes(suf[:4]) != b"KMCS" or bytes(suf[-4:]) != b"KMCS" or suf.size != 8 + self.total * self.rec:
            raise ValueError("suffix file does not hold %d records of %d bytes" % (self.total, self.rec))
        self.records = suf[4:-4].reshape(self.total, self.rec)

    def table(self):
        """-> (hi, lo, cnt): the listing as the scan's SoA table (M-form, MSB-first, right aligned); records outside
        [min_count, max_count] are what ReadNextKmer skips: they come back with count 0, which adds nothing"""
        n = self.total
        idx = np.arange(n, dtype=np.uint64)
        j = np.searchsorted(self.lut, idx, side="right") - 1           # last table entry <= record index
        prefix = (j & ((1 << (2 * self.lut_prefix_len)) - 1)).astype(object)
        val = np.zeros(n, dtype=object)
        for s in range(self.suffix_bytes):
            val = (val << 8) | self.records[:, s].astype(object)
        val = val | (prefix << (8 * self.suffix_bytes))
        cnt = np.zeros(n, dtype=np.uint64)
        for s in range(self.counter_size):
            cnt |= self.records[:, self.suffix_bytes + s].astype(np.uint64) << np.uint64(8 * s)
        cnt[(cnt < self.min_count) | (cnt > self.max_count)] = 0
        hi = np.array([int(v) >> 64 for v in val], dtype=np.uint64)
        lo = np.array([int(v) & 0xFFFFFFFFFFFFFFFF for v in val], dtype=np.uint64)
        return hi, lo, cnt.astype(np.uint32)

    def kmers(self):
        """-> list of (bytes, count) in listing order, skipped records left out: what CKMCFile::ReadNextKmer +
        CKmerAPI::to_string hand to main.cpp:488-490"""
        hi, lo, cnt = self.table()
        out = []
        for h, l, c in zip(hi, lo, cnt):
            if c == 0:
                continue
            v = (int(h) << 64) | int(l)
            out.append((bytes(SYM[(v >> (2 * (self.k - 1 - i))) & 3] for i in range(self.k)), int(c)))
        return out
