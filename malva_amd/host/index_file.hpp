// The index file between `malva-geno index` and `malva-geno call` (main.cpp:406-412 writes, :455-461 reads).
//
// Two containers, one payload (context_bf, bf, ref_bf in that order):
//
//   <vcf>.c<ref_k>.k<k>.malvax.zst   the reference's own: ONE zstd stream (the reference writes level 5, zstdstream.h:52) holding, for each
//       Bloom filter, what BF::operator>> writes (bloom_filter.hpp:127-136): bool _mode, size_t _size, then the
//       sdsl-lite v2.1.1 serialisation of bit_vector and of int_vector<16> -- each a u64 length IN BITS followed by
//       the data as whole 64-bit words (sdsl int_vector<t_width>::serialize with t_width > 0 writes no width byte) --
//       and then what KMAP::operator>> writes (kmap.hpp:52-64): size_t n, n x (size_t length, bytes, int value).
//       The rank directory is not stored; both sides rebuild it on load (bloom_filter.hpp:143).
//       sdsl-lite is a third-party library absent from the reference checkout and no index fixture exists there:
//       FORMAT UNPINNED (restated from sdsl's published serialisation), see DESIGN.md.
//   <vcf>.c<ref_k>.k<k>.malvax.hipz  this build's compact form: the filters as sorted bit positions (a filter is a few million
//       set bits in 2^33..2^37: the reference's form costs two passes over gigabytes of zeros), the keys without padding, all of
//       it in independently compressed chunks (zstd level 1 + CRC-32) that every host core writes and reads side by side.
//
// `index` writes the reference's container unless MALVA_GENO_INDEX_FORMAT=hipz; `call` reads whichever exists
// (.zst first).  The payload in memory is sparse either way: that is what mg_bf_import_sparse / export_sparse take.
#pragma once
#include <sys/stat.h>
#include <zlib.h>
#include <zstd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace malva {

struct IndexPayload {
    struct Filter {
        uint64_t mode = 0;
        std::vector<uint64_t> pos;  // ascending positions of the set bits == counter order
        std::vector<uint16_t> cnt;  // one per set bit
    } filt[2];                      // [0] context_bf, [1] bf  (main.cpp:409-410)
    size_t stride = 0;
    std::vector<char> rows;         // ref_bf keys, NUL-terminated inside `stride` bytes each
    std::vector<int32_t> vals;
};

// ---- reference container -----------------------------------------------------------------------------------------
// The image carries libzstd's runtime system-wide and its header only under /opt/conda (Makefile): the two may be different
// releases.  The streaming calls used here (ZSTD_compressStream2, ZSTD_CCtx_setParameter, ZSTD_decompressStream) are stable
// since 1.4.0 with an unchanged ABI inside major version 1; anything else is refused at the first use instead of
// misbehaving somewhere inside a frame.
inline void zstd_check()
{
    const unsigned rt = ZSTD_versionNumber();
    if (rt / 10000 != ZSTD_VERSION_MAJOR || rt < 10400)
        throw std::runtime_error("libzstd " + std::string(ZSTD_versionString()) + " at run time, header " + std::to_string(ZSTD_VERSION_MAJOR) + "." +
                                 std::to_string(ZSTD_VERSION_MINOR) + "." + std::to_string(ZSTD_VERSION_RELEASE) + ": need the same major version, 1.4.0 or later");
}
class ZstdWriter {
  public:
    explicit ZstdWriter(const std::string &path) : out_(ZSTD_CStreamOutSize())
    {
        zstd_check();
        f_ = fopen(path.c_str(), "wb");
        if (!f_) throw std::runtime_error("cannot write " + path);
        c_ = ZSTD_createCCtx();
        if (!c_) throw std::runtime_error("zstd: no context");
        // Level 1, not the reference's 5 (zstdstream.h:52): any level decodes the same way, the file is no larger on this data (a few
        // million isolated non-zero bytes in gigabytes of zeros), and level 5's match search costs 3x the time per set bit
        // (2e7 set bits in 8 GB: 10.6 s against 3.6 s).
        ZSTD_CCtx_setParameter(c_, ZSTD_c_compressionLevel, 1);
        ZSTD_CCtx_setParameter(c_, ZSTD_c_nbWorkers, 4);        // same frame format; ignored by a single-threaded libzstd
    }
    ~ZstdWriter()
    {
        if (c_) ZSTD_freeCCtx(c_);
        if (f_) fclose(f_);
    }
    void put(const void *p, size_t n) { feed(p, n, ZSTD_e_continue); }
    template <class T> void put_value(const T &v) { put(&v, sizeof v); }
    void finish()
    {
        feed(nullptr, 0, ZSTD_e_end);
        if (fclose(f_) != 0) {
            f_ = nullptr;
            throw std::runtime_error("index file: close failed");
        }
        f_ = nullptr;
    }

  private:
    void feed(const void *p, size_t n, ZSTD_EndDirective mode)
    {
        ZSTD_inBuffer in{p, n, 0};
        for (;;) {
            ZSTD_outBuffer o{out_.data(), out_.size(), 0};
            const size_t left = ZSTD_compressStream2(c_, &o, &in, mode);
            if (ZSTD_isError(left)) throw std::runtime_error(std::string("zstd: ") + ZSTD_getErrorName(left));
            if (o.pos && fwrite(out_.data(), 1, o.pos, f_) != o.pos) throw std::runtime_error("index file: write failed");
            if (mode == ZSTD_e_end ? left == 0 : in.pos == in.size) break;
        }
    }
    FILE *f_ = nullptr;
    ZSTD_CCtx *c_ = nullptr;
    std::vector<char> out_;
};

class ZstdReader { // like zstd::ifstream, a file that is not a zstd frame is read as it is (zstdstream.cpp:156-158)
  public:
    explicit ZstdReader(const std::string &path) : in_(ZSTD_DStreamInSize()), out_(ZSTD_DStreamOutSize() * 8)
    {
        zstd_check();
        f_ = fopen(path.c_str(), "rb");
        if (!f_) throw std::runtime_error("cannot open index " + path);
        d_ = ZSTD_createDCtx();
        in_n_ = fread(in_.data(), 1, in_.size(), f_);
        static const unsigned char magic[4] = {0x28, 0xB5, 0x2F, 0xFD}; // a zstd frame (ZSTD_MAGICNUMBER, little-endian)
        compressed_ = in_n_ >= 4 && memcmp(in_.data(), magic, 4) == 0;
    }
    ~ZstdReader()
    {
        if (d_) ZSTD_freeDCtx(d_);
        if (f_) fclose(f_);
    }
    void get(void *p, size_t n)
    {
        char *dst = (char *)p;
        while (n) {
            if (out_pos_ == out_n_) refill();
            const size_t take = std::min(n, out_n_ - out_pos_);
            memcpy(dst, out_.data() + out_pos_, take);
            out_pos_ += take;
            dst += take;
            n -= take;
        }
    }
    template <class T> T get_value()
    {
        T v;
        get(&v, sizeof v);
        return v;
    }

  private:
    void refill()
    {
        out_pos_ = out_n_ = 0;
        while (out_n_ == 0) {
            if (in_pos_ == in_n_) {
                in_n_ = fread(in_.data(), 1, in_.size(), f_);
                in_pos_ = 0;
                if (in_n_ == 0) throw std::runtime_error("index file: truncated");
            }
            if (!compressed_) {
                out_n_ = std::min(in_n_ - in_pos_, out_.size());
                memcpy(out_.data(), in_.data() + in_pos_, out_n_);
                in_pos_ += out_n_;
                return;
            }
            ZSTD_inBuffer in{in_.data(), in_n_, in_pos_};
            ZSTD_outBuffer o{out_.data(), out_.size(), 0};
            const size_t rc = ZSTD_decompressStream(d_, &o, &in);
            if (ZSTD_isError(rc)) throw std::runtime_error(std::string("index file: zstd: ") + ZSTD_getErrorName(rc));
            in_pos_ = in.pos;
            out_n_ = o.pos;
        }
    }
    FILE *f_ = nullptr;
    ZSTD_DCtx *d_ = nullptr;
    std::vector<char> in_, out_;
    size_t in_n_ = 0, in_pos_ = 0, out_n_ = 0, out_pos_ = 0;
    bool compressed_ = false;
};

// BF::operator>> (bloom_filter.hpp:127-136) from the sparse form
inline void write_bf_sdsl(ZstdWriter &w, const IndexPayload::Filter &f, uint64_t size_bits)
{
    w.put_value<uint8_t>(f.mode ? 1 : 0); // bool _mode
    w.put_value<uint64_t>(size_bits);     // size_t _size
    w.put_value<uint64_t>(size_bits);     // bit_vector: length in bits, then ceil(size/64) words
    const uint64_t n_words = (size_bits + 63) / 64, chunk = 1ULL << 23; // 64 MiB of words at a time
    std::vector<uint64_t> words(std::min(chunk, n_words), 0); // zeroed once; after each piece only the words that got a bit are cleared again
    size_t i = 0;
    for (uint64_t w0 = 0; w0 < n_words; w0 += chunk) {
        const uint64_t nw = std::min(chunk, n_words - w0);
        const size_t first = i;
        for (; i < f.pos.size() && f.pos[i] < (w0 + nw) * 64; ++i) words[(f.pos[i] >> 6) - w0] |= 1ULL << (f.pos[i] & 63);
        w.put(words.data(), nw * 8);
        for (size_t j = first; j < i; ++j) words[(f.pos[j] >> 6) - w0] = 0;
    }
    // int_vector<16>: length in bits, then whole words.  In write mode the reference's _counts is empty.
    const uint64_t n = f.mode ? f.cnt.size() : 0;
    w.put_value<uint64_t>(n * 16);
    if (n) w.put(f.cnt.data(), n * 2);
    const uint64_t tail = (8 - (n * 2) % 8) % 8;
    const char zeros[8] = {0};
    if (tail) w.put(zeros, tail);
}

// BF::operator<< (bloom_filter.hpp:138-146) into the sparse form
inline void read_bf_sdsl(ZstdReader &r, IndexPayload::Filter &f, uint64_t expect_bits, const std::string &path)
{
    f.mode = r.get_value<uint8_t>();
    const uint64_t size = r.get_value<uint64_t>(), bv_bits = r.get_value<uint64_t>();
    if (f.mode > 1 || size != bv_bits) throw std::runtime_error("index " + path + " is corrupt (filter header)");
    if (size != expect_bits) throw std::runtime_error("index " + path + " was built with another -b (filters of " + std::to_string(size) + " bits)");
    const uint64_t n_words = (size + 63) / 64, chunk = 1ULL << 23;
    std::vector<uint64_t> words;
    f.pos.clear();
    for (uint64_t w0 = 0; w0 < n_words; w0 += chunk) {
        const uint64_t nw = std::min(chunk, n_words - w0);
        words.resize(nw);
        r.get(words.data(), nw * 8);
        for (uint64_t j = 0; j < nw; ++j) {
            uint64_t x = words[j];
            while (x) {
                f.pos.push_back((w0 + j) * 64 + (uint64_t)__builtin_ctzll(x));
                x &= x - 1;
            }
        }
        if (f.pos.size() >= 0xFFFFFFFFULL) throw std::runtime_error("index " + path + ": a filter with 2^32 or more set bits");
    }
    const uint64_t cnt_bits = r.get_value<uint64_t>();
    if (cnt_bits % 16 || (f.mode && cnt_bits / 16 != f.pos.size()) || cnt_bits / 16 > f.pos.size())
        throw std::runtime_error("index " + path + " is corrupt (counter vector of " + std::to_string(cnt_bits) + " bits for " + std::to_string(f.pos.size()) + " set bits)");
    const uint64_t n = cnt_bits / 16;
    f.cnt.assign(f.pos.size(), 0);
    if (n) r.get(f.cnt.data(), n * 2);
    char pad[8];
    const uint64_t tail = (8 - (n * 2) % 8) % 8;
    if (tail) r.get(pad, tail);
}

inline void save_index_zst(const std::string &path, const IndexPayload &p, uint64_t bf_bits)
{
    ZstdWriter w(path);
    write_bf_sdsl(w, p.filt[0], bf_bits);
    write_bf_sdsl(w, p.filt[1], bf_bits);
    const uint64_t n = p.vals.size(); // KMAP::operator>>, kmap.hpp:52-64
    w.put_value<uint64_t>(n);
    std::vector<char> buf; // (the compressor is fed megabytes at a time: three calls per key cost 4 s per 1e7 keys)
    buf.reserve((8u << 20) + 4096);
    for (uint64_t i = 0; i < n; ++i) {
        const char *key = &p.rows[i * p.stride];
        const uint64_t len = strnlen(key, p.stride);
        const int32_t val = p.vals[i];
        const size_t at = buf.size();
        buf.resize(at + 8 + len + 4);
        memcpy(&buf[at], &len, 8);
        memcpy(&buf[at + 8], key, len);
        memcpy(&buf[at + 8 + len], &val, 4);
        if (buf.size() >= (8u << 20)) {
            w.put(buf.data(), buf.size());
            buf.clear();
        }
    }
    if (!buf.empty()) w.put(buf.data(), buf.size());
    w.finish();
}

inline void load_index_zst(const std::string &path, IndexPayload &p, uint64_t bf_bits, size_t stride)
{
    ZstdReader r(path);
    read_bf_sdsl(r, p.filt[0], bf_bits, path);
    read_bf_sdsl(r, p.filt[1], bf_bits, path);
    const uint64_t n = r.get_value<uint64_t>(); // KMAP::operator<<, kmap.hpp:66-82
    if (n > 0xFFFFFFFEULL) throw std::runtime_error("index " + path + " is corrupt (key count)");
    p.stride = stride;
    p.rows.clear();
    p.vals.clear();
    for (uint64_t i = 0; i < n; ++i) {
        const uint64_t len = r.get_value<uint64_t>();
        if (len >= stride) throw std::runtime_error("index " + path + ": key of " + std::to_string(len) + " bytes");
        p.rows.resize(p.rows.size() + stride, 0);
        r.get(&p.rows[i * stride], len);
        p.vals.push_back(r.get_value<int32_t>());
    }
}

// ---- this build's compact container ----------------------------------------------------------------------------------
namespace hipz {
template <class T> void gz_put(gzFile f, const T *p, size_t n)
{
    const char *b = (const char *)p;
    size_t left = n * sizeof(T);
    while (left) {
        const unsigned chunk = (unsigned)std::min<size_t>(left, 1u << 30);
        if (gzwrite(f, b, chunk) != (int)chunk) throw std::runtime_error("index file: write failed");
        b += chunk;
        left -= chunk;
    }
}
template <class T> void gz_get(gzFile f, T *p, size_t n)
{
    char *b = (char *)p;
    size_t left = n * sizeof(T);
    while (left) {
        const unsigned chunk = (unsigned)std::min<size_t>(left, 1u << 30);
        if (gzread(f, b, chunk) != (int)chunk) throw std::runtime_error("index file: truncated");
        b += chunk;
        left -= chunk;
    }
}
const char MAGIC[8] = {'M', 'G', 'H', 'I', 'P', 'X', '2', '\n'};
} // namespace hipz

// Version 3 of the compact container: the payload's arrays as SECTIONS of independently compressed chunks (zstd level 1, 8 MiB of
// raw bytes each), so that writing and reading run on all host cores.  (Version 2 pushed everything through one gzip stream: 5 s to
// write and 2.3 s to read the index of 1e7 SNPs, 40 s and 18 s at whole-genome size.)  Layout, all integers little-endian u64:
//   "MGHIPX3\n", k, ref_k, bf_bits
//   mode[0], n_set[0], mode[1], n_set[1], n_keys, key_bytes        (key_bytes: the keys as rows of this many bytes, NUL-padded)
//   six sections -- pos[0] (u64), cnt[0] (u16), pos[1], cnt[1], keys, vals (i32) -- each:
//       raw_bytes, n_chunks, then per chunk its compressed size and the CRC-32 of its raw bytes, then the chunks back to back
// Version 2 files are still read.
namespace hipz {
const char MAGIC3[8] = {'M', 'G', 'H', 'I', 'P', 'X', '3', '\n'};
// version 4 = version 3 + two header words tying the index to the panel it was built from: the VCF's size and its modification time
// (ns).  `call` takes a compact index whose tie does not match the VCF it is given for stale (a file copied or restored beside a
// reference container rebuilt for another panel carries a fresh mtime and the same -k/-r/-b) and reads the reference's container.
const char MAGIC4[8] = {'M', 'G', 'H', 'I', 'P', 'X', '4', '\n'};
struct PanelTie {
    uint64_t size = 0, mtime_ns = 0; // 0, 0: unknown (an index of version 3, or one converted from the reference's container)
    bool known() const { return size || mtime_ns; }
    bool operator==(const PanelTie &o) const { return size == o.size && mtime_ns == o.mtime_ns; }
};
inline PanelTie panel_tie_of(const std::string &vcf_path)
{
    struct stat st;
    PanelTie t;
    if (stat(vcf_path.c_str(), &st) == 0) {
        t.size = (uint64_t)st.st_size;
        t.mtime_ns = (uint64_t)st.st_mtim.tv_sec * 1000000000ULL + (uint64_t)st.st_mtim.tv_nsec;
    }
    return t;
}
constexpr size_t CHUNK = 8u << 20;
template <class F> void parallel_for(size_t n, F f)
{
    const unsigned nt = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), 16, n}));
    std::vector<std::exception_ptr> errs(nt);
    std::vector<std::thread> th;
    auto work = [&](unsigned t) {
        try {
            for (size_t i = t; i < n; i += nt) f(i);
        } catch (...) {
            errs[t] = std::current_exception();
        }
    };
    for (unsigned t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    for (auto &e : errs)
        if (e) std::rethrow_exception(e);
}
inline void put_section(FILE *f, const void *data, size_t bytes)
{
    const size_t n_chunks = (bytes + CHUNK - 1) / CHUNK;
    std::vector<std::vector<char>> out(n_chunks);
    std::vector<uint64_t> crcs(n_chunks);
    parallel_for(n_chunks, [&](size_t i) {
        const size_t at = i * CHUNK, n = std::min(CHUNK, bytes - at);
        crcs[i] = crc32(crc32(0L, Z_NULL, 0), (const Bytef *)data + at, (uInt)n);
        out[i].resize(ZSTD_compressBound(n));
        const size_t c = ZSTD_compress(out[i].data(), out[i].size(), (const char *)data + at, n, 1);
        if (ZSTD_isError(c)) throw std::runtime_error(std::string("index file: zstd: ") + ZSTD_getErrorName(c));
        out[i].resize(c);
    });
    std::vector<uint64_t> head{(uint64_t)bytes, (uint64_t)n_chunks};
    for (size_t i = 0; i < n_chunks; ++i) {
        head.push_back(out[i].size());
        head.push_back(crcs[i]);
    }
    if (fwrite(head.data(), 8, head.size(), f) != head.size()) throw std::runtime_error("index file: write failed");
    for (auto &o : out)
        if (!o.empty() && fwrite(o.data(), 1, o.size(), f) != o.size()) throw std::runtime_error("index file: write failed");
}
inline void get_section(FILE *f, void *data, size_t bytes, uint64_t file_bytes, const std::string &path)
{
    uint64_t head[2];
    if (fread(head, 8, 2, f) != 2) throw std::runtime_error("index file: truncated");
    const size_t n_chunks = (bytes + CHUNK - 1) / CHUNK;
    if (head[0] != bytes || head[1] != n_chunks) throw std::runtime_error("index " + path + " is corrupt (section header)");
    std::vector<uint64_t> pairs(2 * n_chunks), csize(n_chunks), crcs(n_chunks);
    if (n_chunks && fread(pairs.data(), 8, 2 * n_chunks, f) != 2 * n_chunks) throw std::runtime_error("index file: truncated");
    for (size_t i = 0; i < n_chunks; ++i) {
        csize[i] = pairs[2 * i];
        crcs[i] = pairs[2 * i + 1];
    }
    uint64_t total = 0;
    for (uint64_t c : csize) {
        if (c > file_bytes) throw std::runtime_error("index " + path + " is corrupt (chunk size)");
        total += c;
    }
    if (total > file_bytes) throw std::runtime_error("index " + path + " is corrupt (chunk sizes)");
    std::vector<char> comp(total);
    if (total && fread(comp.data(), 1, total, f) != total) throw std::runtime_error("index file: truncated");
    std::vector<uint64_t> at(n_chunks + 1, 0);
    for (size_t i = 0; i < n_chunks; ++i) at[i + 1] = at[i] + csize[i];
    parallel_for(n_chunks, [&](size_t i) {
        const size_t o = i * CHUNK, n = std::min(CHUNK, bytes - o);
        const size_t d = ZSTD_decompress((char *)data + o, n, comp.data() + at[i], csize[i]);
        if (ZSTD_isError(d) || d != n || crc32(crc32(0L, Z_NULL, 0), (const Bytef *)data + o, (uInt)n) != crcs[i])
            throw std::runtime_error("index " + path + " is corrupt (chunk " + std::to_string(i) + ")");
    });
}
} // namespace hipz

inline void save_index_hipz(const std::string &path, const IndexPayload &p, uint64_t k, uint64_t ref_k, uint64_t bf_bits, hipz::PanelTie tie = hipz::PanelTie())
{
    using namespace hipz;
    zstd_check();
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) throw std::runtime_error("cannot write " + path);
    try {
        const uint64_t nkeys = p.vals.size();
        uint64_t key_bytes = 1;
        for (uint64_t i = 0; i < nkeys; ++i) key_bytes = std::max<uint64_t>(key_bytes, strnlen(&p.rows[i * p.stride], p.stride) + 1);
        key_bytes = std::min<uint64_t>(key_bytes, p.stride);
        const uint64_t hdr[9] = {k, ref_k, bf_bits, p.filt[0].mode, p.filt[0].pos.size(), p.filt[1].mode, p.filt[1].pos.size(), nkeys, key_bytes};
        const uint64_t tie_words[2] = {tie.size, tie.mtime_ns};
        if (fwrite(MAGIC4, 1, 8, f) != 8 || fwrite(hdr, 8, 9, f) != 9 || fwrite(tie_words, 8, 2, f) != 2) throw std::runtime_error("index file: write failed");
        for (int i = 0; i < 2; ++i) {
            put_section(f, p.filt[i].pos.data(), p.filt[i].pos.size() * 8);
            put_section(f, p.filt[i].cnt.data(), p.filt[i].cnt.size() * 2);
        }
        std::vector<char> keys(nkeys * key_bytes, 0); // the rows without their padding up to the ABI's stride
        parallel_for((nkeys + 65535) / 65536, [&](size_t c) {
            for (uint64_t i = c * 65536, e = std::min<uint64_t>(nkeys, (c + 1) * 65536); i < e; ++i)
                memcpy(&keys[i * key_bytes], &p.rows[i * p.stride], std::min<size_t>(key_bytes, p.stride));
        });
        put_section(f, keys.data(), keys.size());
        put_section(f, p.vals.data(), p.vals.size() * 4);
    } catch (...) {
        fclose(f);
        throw;
    }
    if (fclose(f) != 0) throw std::runtime_error("index file: close failed");
}

inline void load_index_hipz3(const std::string &path, IndexPayload &p, uint64_t k, uint64_t ref_k, uint64_t bf_bits, size_t stride)
{
    using namespace hipz;
    zstd_check();
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open index " + path);
    struct stat st; // (of the descriptor that is open, not of whatever the name points at by now; an unknown size bounds nothing)
    const uint64_t file_bytes = fstat(fileno(f), &st) == 0 && st.st_size > 0 ? (uint64_t)st.st_size : (uint64_t)1 << 50;
    try {
        char magic[8];
        uint64_t hdr[9], tie_words[2];
        if (fread(magic, 1, 8, f) != 8 || fread(hdr, 8, 9, f) != 9) throw std::runtime_error("index file: truncated");
        const bool v4 = memcmp(magic, MAGIC4, 8) == 0;
        if (v4 && fread(tie_words, 8, 2, f) != 2) throw std::runtime_error("index file: truncated");
        if ((!v4 && memcmp(magic, MAGIC3, 8) != 0) || hdr[0] != k || hdr[1] != ref_k || hdr[2] != bf_bits)
            throw std::runtime_error("index " + path + " was built with other -k/-r/-b");
        // a corrupt length must not turn into a huge allocation: a filter holds fewer than 2^32 set bits (mg_bf_finalize), and no
        // field can promise more bytes than zstd can have packed into the file
        const uint64_t most = std::min<uint64_t>(bf_bits, 0xFFFFFFFEULL), inflated_cap = file_bytes * 4000 + (1u << 20);
        const uint64_t nkeys = hdr[7], key_bytes = hdr[8];
        for (int i = 0; i < 2; ++i)
            if (hdr[3 + 2 * i] > 1 || hdr[4 + 2 * i] > most || hdr[4 + 2 * i] * 10 > inflated_cap) throw std::runtime_error("index " + path + " is corrupt (filter header)");
        if (key_bytes == 0 || key_bytes > stride || nkeys > 0xFFFFFFFEULL || nkeys * (key_bytes + 4) > inflated_cap)
            throw std::runtime_error("index " + path + " is corrupt (key table)");
        for (int i = 0; i < 2; ++i) {
            p.filt[i].mode = hdr[3 + 2 * i];
            p.filt[i].pos.resize(hdr[4 + 2 * i]);
            p.filt[i].cnt.resize(hdr[4 + 2 * i]);
            get_section(f, p.filt[i].pos.data(), p.filt[i].pos.size() * 8, file_bytes, path);
            get_section(f, p.filt[i].cnt.data(), p.filt[i].cnt.size() * 2, file_bytes, path);
        }
        std::vector<char> keys(nkeys * key_bytes);
        get_section(f, keys.data(), keys.size(), file_bytes, path);
        p.stride = stride;
        p.rows.assign(nkeys * stride, 0);
        p.vals.resize(nkeys);
        parallel_for((nkeys + 65535) / 65536, [&](size_t c) {
            for (uint64_t i = c * 65536, e = std::min<uint64_t>(nkeys, (c + 1) * 65536); i < e; ++i) {
                memcpy(&p.rows[i * stride], &keys[i * key_bytes], key_bytes);
                p.rows[i * stride + std::min<uint64_t>(key_bytes, stride - 1)] = 0; // (whatever the file says, a key ends inside its row)
            }
        });
        get_section(f, p.vals.data(), p.vals.size() * 4, file_bytes, path);
    } catch (...) {
        fclose(f);
        throw;
    }
    fclose(f);
}

// the panel tie in a compact index's header (unknown for older versions or an unreadable file)
inline hipz::PanelTie index_hipz_tie(const std::string &path)
{
    using namespace hipz;
    PanelTie t;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return t;
    char magic[8];
    uint64_t hdr[11];
    if (fread(magic, 1, 8, f) == 8 && memcmp(magic, MAGIC4, 8) == 0 && fread(hdr, 8, 11, f) == 11) {
        t.size = hdr[9];
        t.mtime_ns = hdr[10];
    }
    fclose(f);
    return t;
}

inline void load_index_hipz(const std::string &path, IndexPayload &p, uint64_t k, uint64_t ref_k, uint64_t bf_bits, size_t stride)
{
    using namespace hipz;
    {
        char magic[8] = {0};
        FILE *probe = fopen(path.c_str(), "rb");
        const bool v3 = probe && fread(magic, 1, 8, probe) == 8 && (memcmp(magic, MAGIC3, 8) == 0 || memcmp(magic, MAGIC4, 8) == 0);
        if (probe) fclose(probe);
        if (v3) return load_index_hipz3(path, p, k, ref_k, bf_bits, stride);
    }
    gzFile f = gzopen(path.c_str(), "rb"); // version 2: one gzip stream
    if (!f) throw std::runtime_error("cannot open index " + path);
    struct stat st;
    const uint64_t file_bytes = stat(path.c_str(), &st) == 0 ? (uint64_t)st.st_size : 0;
    try {
        char magic[8];
        gz_get(f, magic, 8);
        uint64_t hdr[3];
        gz_get(f, hdr, 3);
        if (memcmp(magic, MAGIC, 8) != 0 || hdr[0] != k || hdr[1] != ref_k || hdr[2] != bf_bits)
            throw std::runtime_error("index " + path + " was built with other -k/-r/-b");
        // a corrupt length must not turn into a huge allocation: a filter holds fewer than 2^32 set bits (mg_bf_finalize),
        // and no field can promise more bytes than gzip can have packed into the file
        const uint64_t most = std::min<uint64_t>(bf_bits, 0xFFFFFFFEULL), inflated_cap = file_bytes * 1100 + (1u << 20);
        for (int i = 0; i < 2; ++i) {
            uint64_t h2[2];
            gz_get(f, h2, 2);
            if (h2[0] > 1 || h2[1] > most || h2[1] * 10 > inflated_cap) throw std::runtime_error("index " + path + " is corrupt (filter header)");
            p.filt[i].mode = h2[0];
            p.filt[i].pos.resize(h2[1]);
            p.filt[i].cnt.resize(h2[1]);
            gz_get(f, p.filt[i].pos.data(), p.filt[i].pos.size());
            gz_get(f, p.filt[i].cnt.data(), p.filt[i].cnt.size());
        }
        uint64_t nkeys = 0, file_stride = 0;
        gz_get(f, &nkeys, 1);
        gz_get(f, &file_stride, 1);
        if (file_stride != stride || nkeys > 0xFFFFFFFEULL || nkeys * (stride + 4) > inflated_cap) throw std::runtime_error("index " + path + " is corrupt (key table)");
        p.stride = stride;
        p.rows.resize(nkeys * stride);
        p.vals.resize(nkeys);
        gz_get(f, p.rows.data(), p.rows.size());
        gz_get(f, p.vals.data(), p.vals.size());
    } catch (...) {
        gzclose(f);
        throw;
    }
    gzclose(f);
}

} // namespace malva
