"""The reference's index file <vcf>.c<ref_k>.k<k>.malvax.zst restated: main.cpp:406-412 (write) / :455-461 (read),
BF::operator>> / << (bloom_filter.hpp:127-146), KMAP::operator>> / << (kmap.hpp:52-82), around sdsl-lite v2.1.1's
serialisation of bit_vector and int_vector<16> (u64 length in bits, then the data as whole 64-bit words; the
fixed-width int_vector writes no width byte) and one zstd stream (zstdstream.h:52: level 5; zstd::ifstream also accepts
an uncompressed file, zstdstream.cpp:156-158).

TEST INFRASTRUCTURE.  sdsl-lite is a third-party library absent from the image and the reference checkout holds no
index fixture: FORMAT UNPINNED -- this pins the product's writer/reader (malva_amd/host/index_file.hpp) against an
independent restatement of the same published layouts, not against a file the reference binary wrote.
zstd itself comes from the system's libzstd.so.1 through ctypes."""
import ctypes as C
import struct

import numpy as np

from . import capi

_Z = None


def _zstd():
    global _Z
    if _Z is None:
        _Z = C.CDLL("libzstd.so.1")
        _Z.ZSTD_compressBound.restype = C.c_size_t
        _Z.ZSTD_compressBound.argtypes = [C.c_size_t]
        _Z.ZSTD_compress.restype = C.c_size_t
        _Z.ZSTD_compress.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        _Z.ZSTD_isError.restype = C.c_uint
        _Z.ZSTD_isError.argtypes = [C.c_size_t]
        _Z.ZSTD_createDStream.restype = C.c_void_p
        _Z.ZSTD_freeDStream.argtypes = [C.c_void_p]
        _Z.ZSTD_decompressStream.restype = C.c_size_t
        _Z.ZSTD_decompressStream.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return _Z


class _Buf(C.Structure):
    _fields_ = [("p", C.c_void_p), ("size", C.c_size_t), ("pos", C.c_size_t)]


def zstd_compress_frames(chunks, level=5):
    """each chunk (bytes-like / uint8 array) becomes one zstd frame; concatenated frames are one valid zstd stream"""
    z = _zstd()
    out = []
    for ch in chunks:
        a = np.frombuffer(ch, dtype=np.uint8) if not isinstance(ch, np.ndarray) else np.ascontiguousarray(ch).view(np.uint8).reshape(-1)
        cap = z.ZSTD_compressBound(a.size)
        dst = np.empty(cap, dtype=np.uint8)
        n = z.ZSTD_compress(dst.ctypes.data, cap, a.ctypes.data, a.size, level)
        if z.ZSTD_isError(n):
            raise RuntimeError("ZSTD_compress failed")
        out.append(dst[:n].tobytes())
    return b"".join(out)


def zstd_stream_reader(path, out_chunk=1 << 24):
    """generator of decompressed uint8 arrays; a file that does not start with a zstd frame is passed through"""
    data = np.fromfile(path, dtype=np.uint8)
    if data.size < 4 or bytes(data[:4]) != b"\x28\xb5\x2f\xfd":
        for a in range(0, data.size, out_chunk):
            yield data[a:a + out_chunk]
        return
    z = _zstd()
    ds = z.ZSTD_createDStream()
    try:
        inb = _Buf(data.ctypes.data, data.size, 0)
        while inb.pos < inb.size:
            dst = np.empty(out_chunk, dtype=np.uint8)
            outb = _Buf(dst.ctypes.data, dst.size, 0)
            rc = z.ZSTD_decompressStream(ds, C.byref(outb), C.byref(inb))
            if z.ZSTD_isError(rc):
                raise RuntimeError("ZSTD_decompressStream failed")
            if outb.pos:
                yield dst[:outb.pos]
    finally:
        z.ZSTD_freeDStream(ds)


class _Pull:
    def __init__(self, gen):
        self.gen, self.cur, self.pos = gen, np.zeros(0, dtype=np.uint8), 0

    def take(self, n):
        parts = []
        while n:
            if self.pos == self.cur.size:
                self.cur, self.pos = next(self.gen), 0
            t = min(n, self.cur.size - self.pos)
            parts.append(self.cur[self.pos:self.pos + t])
            self.pos += t
            n -= t
        return parts[0] if len(parts) == 1 else np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)

    def value(self, fmt):
        return struct.unpack("<" + fmt, self.take(struct.calcsize("<" + fmt)).tobytes())[0]


def _bf_chunks(bf: capi.BF, mode: bool):
    """BF::operator>> : bool, size_t, bit_vector, int_vector<16>"""
    size = bf.size
    yield struct.pack("<BQQ", 1 if mode else 0, size, size)
    words = bf.words()                                   # ceil(size / 64) words, bit i = word i>>6 bit i&63 (sdsl's order)
    step = 1 << 24
    for a in range(0, words.size, step):
        yield words[a:a + step]
    counts = bf.counts() if mode else np.zeros(0, dtype=np.uint16)
    yield struct.pack("<Q", counts.size * 16)
    if counts.size:
        yield np.ascontiguousarray(counts).view(np.uint8)
    tail = (8 - (counts.size * 2) % 8) % 8
    if tail:
        yield b"\0" * tail


def write_index(path, context_bf: capi.BF, bf: capi.BF, ref_bf: capi.KMAP, compress=True):
    """main.cpp:406-412: context_bf, bf, ref_bf into one stream (both filters in read mode, main.cpp:378,404)"""
    def chunks():
        yield from _bf_chunks(context_bf, True)
        yield from _bf_chunks(bf, True)
        items = list(ref_bf.items())
        yield struct.pack("<Q", len(items))
        yield b"".join(struct.pack("<Q", len(k)) + k + struct.pack("<i", v) for k, v in items)
    with open(path, "wb") as fh:
        if compress:
            fh.write(zstd_compress_frames(chunks()))
        else:
            for ch in chunks():
                fh.write(ch if isinstance(ch, (bytes, bytearray)) else np.ascontiguousarray(ch).tobytes())


def read_index(path):
    """-> [(mode, size, set positions (ascending u64), counts u16)] x 2 (context_bf, bf), {key: value}"""
    pull = _Pull(zstd_stream_reader(path))
    filters = []
    for _ in range(2):
        mode, size, bits = pull.value("B"), pull.value("Q"), pull.value("Q")
        assert bits == size and mode in (0, 1)
        n_words = (size + 63) // 64
        pos = []
        step = 1 << 21
        for a in range(0, n_words, step):
            w = pull.take(8 * min(step, n_words - a)).view(np.uint64)
            nz = np.nonzero(w)[0]
            for j in nz:
                x = int(w[j])
                while x:
                    b = (x & -x).bit_length() - 1
                    pos.append((a + int(j)) * 64 + b)
                    x &= x - 1
        cbits = pull.value("Q")
        assert cbits % 16 == 0
        counts = pull.take(cbits // 8).view(np.uint16).copy()
        pull.take((8 - (cbits // 8) % 8) % 8)
        filters.append((mode, size, np.array(pos, dtype=np.uint64), counts))
    n = pull.value("Q")
    kmap = {}
    for _ in range(n):
        ln = pull.value("Q")
        key = pull.take(ln).tobytes()
        kmap[key] = pull.value("i")
    return filters, kmap
