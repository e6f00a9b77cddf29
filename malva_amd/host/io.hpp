// Host I/O for the malva-geno driver: gz-transparent line reader, FASTA, and the
// VCF text decode the hot path's record model needs.  These stand where the
// reference uses kseq.h (main.cpp:117,283-295) and htslib (variant.hpp:66-211,
// main.cpp:190-219); only the behaviour the reference observes through those
// libraries is implemented (text VCF, plain or gzip/bgzip; no BCF).
#pragma once
#include <zlib.h>

#include <algorithm>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <string_view>
#include <condition_variable>
#include <deque>
#include <exception>
#include <memory>
#include <mutex>
#include <thread>
#include <utility>
#include <vector>

namespace malva {

// gzopen() reads plain files transparently, so one reader serves .vcf/.vcf.gz/.fa/.fa.gz -- and .bcf: a BCF2 file
// (BGZF = concatenated gzip members, which gzread walks through; magic "BCF\2\2") is translated record by record into
// the VCF text lines it encodes, so everything downstream reads one format.  The reference gets BCF through htslib's
// bcf_open / bcf_read (main.cpp:261-272, 309-314; variant.hpp:66-211), third party and absent here: the binary layout is
// restated from the published VCF/BCF specification (v4.3, section 6), "parity unpinned" -- tests/bcf_writer.py is the
// independent second reading the tests use.
class LineReader {
    gzFile f = nullptr;
    std::vector<char> buf;
    size_t pos = 0, end = 0;
    // ---- BCF ----
    bool bcf = false;
    std::vector<std::string> bcf_header; // header text, line by line, handed out first
    size_t bcf_hdr_at = 0;
    std::vector<std::string> dict, contigs; // string dictionary (FILTER / INFO / FORMAT ids) and contig names
    std::vector<char> rec;

    // ---- BGZF (what bgzip writes: .vcf.gz as tabix wants it, and every .bcf) ----
    // A BGZF file is a sequence of gzip members of at most 64 KiB of data each, whose header says how long the member is
    // (extra sub-field 'BC') and whose trailer says how much it inflates to: the members of a group are inflated side by side,
    // each straight to its place in the buffer.  zlib's gzread does the same work on one thread: 0.7 of the 1.2 s `call` spent on
    // the SARS-CoV-2 panel.
    bool bgzf = false;
    int bfd = -1;
    std::vector<unsigned char> craw; // compressed bytes read ahead
    size_t craw_n = 0;
    bool craw_eof = false;
    static bool bgzf_header(const unsigned char *h, size_t n, size_t *member_bytes) // h: at least 18 bytes of a member's start
    {
        if (n < 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) return false;
        const size_t xlen = h[10] | (size_t)h[11] << 8;
        if (xlen != 6 || h[12] != 'B' || h[13] != 'C' || h[14] != 2 || h[15] != 0) return false; // (bgzip writes exactly this; anything else goes through zlib)
        *member_bytes = (size_t)(h[16] | (size_t)h[17] << 8) + 1;
        return *member_bytes >= 26;
    }
    bool fill_bgzf()
    {
        struct Member {
            size_t at, bytes, out_at, out_bytes;
        };
        // inflated bytes per group (MALVA_GENO_BGZF_GROUP: tests put a group seam every few members)
        static const size_t group_bytes = getenv("MALVA_GENO_BGZF_GROUP") ? (size_t)std::max(1L, atol(getenv("MALVA_GENO_BGZF_GROUP"))) : (size_t)24 << 20;
        for (;;) {
            std::vector<Member> ms;
            size_t at = 0, total = 0;
            for (;;) { // members already read, then more of the file
                size_t mb = 0;
                const bool have_header = craw_n - at >= 18;
                if (have_header && !bgzf_header(craw.data() + at, craw_n - at, &mb)) throw std::runtime_error("BGZF: a block that is not a BGZF member");
                if (have_header && craw_n - at >= mb) {
                    const unsigned char *t = craw.data() + at + mb - 4;
                    const size_t isize = t[0] | (size_t)t[1] << 8 | (size_t)t[2] << 16 | (size_t)t[3] << 24;
                    if (isize > (1u << 16)) throw std::runtime_error("BGZF: a block larger than 64 KiB");
                    ms.push_back({at, mb, total, isize});
                    at += mb;
                    total += isize;
                    if (total >= group_bytes) break;
                    continue;
                }
                if (craw_eof) {
                    if (craw_n - at) throw std::runtime_error("BGZF: truncated file");
                    break;
                }
                if (!ms.empty() && total >= group_bytes / 6) break; // enough for a group: the partial member waits for the next one
                if (craw.size() < craw_n + (8u << 20)) craw.resize(craw_n + (8u << 20));
                const ssize_t got = ::read(bfd, craw.data() + craw_n, 8u << 20);
                if (got < 0) throw std::runtime_error("BGZF: read failed");
                if (got == 0) craw_eof = true;
                craw_n += (size_t)got;
            }
            if (ms.empty()) return false; // end of file
            if (buf.size() < total) buf.resize(total);
            const unsigned n_threads = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), 8, ms.size() / 8 + 1}));
            std::vector<int> bad(n_threads, 0);
            auto work = [&](unsigned t) {
                for (size_t i = t; i < ms.size(); i += n_threads) {
                    const Member &m = ms[i];
                    if (!m.out_bytes) continue; // (the end-of-file marker, or an empty flush)
                    z_stream zs{};
                    if (inflateInit2(&zs, -15) != Z_OK) {
                        bad[t] = 1;
                        return;
                    }
                    zs.next_in = craw.data() + m.at + 18;
                    zs.avail_in = (uInt)(m.bytes - 18 - 8);
                    zs.next_out = (Bytef *)buf.data() + m.out_at;
                    zs.avail_out = (uInt)m.out_bytes;
                    const int rc = inflate(&zs, Z_FINISH);
                    inflateEnd(&zs);
                    const unsigned char *c = craw.data() + m.at + m.bytes - 8;
                    const uLong want_crc = c[0] | (uLong)c[1] << 8 | (uLong)c[2] << 16 | (uLong)c[3] << 24;
                    if (rc != Z_STREAM_END || zs.avail_out != 0 || crc32(crc32(0L, Z_NULL, 0), (const Bytef *)buf.data() + m.out_at, (uInt)m.out_bytes) != want_crc) {
                        bad[t] = 1;
                        return;
                    }
                }
            };
            std::vector<std::thread> th;
            for (unsigned t = 1; t < n_threads; ++t) th.emplace_back(work, t);
            work(0);
            for (auto &x : th) x.join();
            for (int b : bad)
                if (b) throw std::runtime_error("BGZF: corrupt block");
            memmove(craw.data(), craw.data() + at, craw_n - at); // the partial member at the end, if any
            craw_n -= at;
            if (!total) continue; // only empty members: go on
            pos = 0;
            end = total;
            return true;
        }
    }
    bool fill()
    {
        if (bgzf) return fill_bgzf();
        const int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n <= 0) return false;
        pos = 0;
        end = (size_t)n;
        return true;
    }
    bool raw(void *dst, size_t n) // n bytes of the (inflated) stream; false at a clean end of file before the first byte
    {
        char *d = (char *)dst;
        size_t got = 0;
        while (got < n) {
            if (pos == end && !fill()) {
                if (got == 0) return false;
                throw std::runtime_error("BCF: truncated file");
            }
            const size_t t = std::min(n - got, end - pos);
            memcpy(d + got, buf.data() + pos, t);
            pos += t;
            got += t;
        }
        return true;
    }
    static std::string attr(const std::string &line, const char *key) // value of key= inside <...>, unquoted
    {
        const std::string k = std::string(key) + "=";
        size_t at = line.find("<");
        while (at != std::string::npos) {
            at = line.find(k, at);
            if (at == std::string::npos) break;
            if (line[at - 1] == '<' || line[at - 1] == ',') {
                size_t b = at + k.size(), e = b;
                if (b < line.size() && line[b] == '"') {
                    e = line.find('"', ++b);
                } else
                    while (e < line.size() && line[e] != ',' && line[e] != '>') ++e;
                return line.substr(b, e == std::string::npos ? std::string::npos : e - b);
            }
            ++at;
        }
        return std::string();
    }
    void bcf_open()
    {
        uint32_t l_text;
        if (!raw(&l_text, 4)) throw std::runtime_error("BCF: no header");
        std::string text(l_text, '\0');
        raw(&text[0], l_text);
        while (!text.empty() && text.back() == '\0') text.pop_back();
        size_t a = 0;
        while (a < text.size()) {
            size_t b = text.find('\n', a);
            if (b == std::string::npos) b = text.size();
            if (b > a) bcf_header.push_back(text.substr(a, b - a));
            a = b + 1;
        }
        // dictionaries (VCF 4.3 section 6.2.1): PASS is entry 0 unless declared elsewhere; ids in order of first
        // appearance over FILTER / INFO / FORMAT lines, an explicit IDX= wins; contigs likewise from ##contig lines
        auto put = [](std::vector<std::string> &d, const std::string &id, const std::string &idx, size_t next) {
            const size_t at = idx.empty() ? next : (size_t)strtoul(idx.c_str(), nullptr, 10);
            if (d.size() <= at) d.resize(at + 1);
            d[at] = id;
        };
        dict.push_back("PASS");
        size_t next_contig = 0;
        for (const std::string &line : bcf_header) {
            const bool sd = line.rfind("##INFO=", 0) == 0 || line.rfind("##FORMAT=", 0) == 0 || line.rfind("##FILTER=", 0) == 0;
            if (sd) {
                const std::string id = attr(line, "ID"), idx = attr(line, "IDX");
                if (std::find(dict.begin(), dict.end(), id) != dict.end() && idx.empty()) continue;
                if (id == "PASS" && idx.empty()) continue;
                put(dict, id, idx, dict.size());
            } else if (line.rfind("##contig=", 0) == 0) {
                const std::string idx = attr(line, "IDX");
                put(contigs, attr(line, "ID"), idx, next_contig);
                next_contig = contigs.size();
            }
        }
    }
    // one typed value descriptor: type in the low nibble, count in the high one (15 = a typed integer follows)
    struct Cur {
        const unsigned char *p, *e;
        void need(size_t n) const
        {
            if ((size_t)(e - p) < n) throw std::runtime_error("BCF: record overruns its length");
        }
        uint32_t u8() { need(1); return *p++; }
        int32_t i_of(int type)
        {
            if (type == 1) { need(1); const int8_t v = (int8_t)*p; p += 1; return v; }
            if (type == 2) { need(2); int16_t v; memcpy(&v, p, 2); p += 2; return v; }
            if (type == 3) { need(4); int32_t v; memcpy(&v, p, 4); p += 4; return v; }
            throw std::runtime_error("BCF: integer expected");
        }
        void desc(int &type, uint32_t &n)
        {
            const uint32_t d = u8();
            type = (int)(d & 15);
            n = d >> 4;
            if (n == 15) {
                int t2;
                uint32_t one;
                desc(t2, one);
                n = (uint32_t)i_of(t2);
            }
        }
        int32_t typed_int()
        {
            int t;
            uint32_t n;
            desc(t, n);
            if (n != 1) throw std::runtime_error("BCF: scalar expected");
            return i_of(t);
        }
        std::string typed_str()
        {
            int t;
            uint32_t n;
            desc(t, n);
            if (n && t != 7) throw std::runtime_error("BCF: string expected");
            need(n);
            std::string s((const char *)p, n);
            p += n;
            return s;
        }
    };
    static bool int_missing(int type, int32_t v) { return type == 1 ? v == -128 : type == 2 ? v == -32768 : v == INT32_MIN; }
    static bool int_eov(int type, int32_t v) { return type == 1 ? v == -127 : type == 2 ? v == -32767 : v == INT32_MIN + 1; }
    static void put_float(std::string &out, uint32_t bits)
    {
        if (bits == 0x7F800001u) { out += '.'; return; }
        float v;
        memcpy(&v, &bits, 4);
        char b[40];
        snprintf(b, sizeof b, "%.9g", (double)v);
        out += b;
    }
    // values of one vector (INFO value, or one sample's FORMAT value) as VCF text
    void put_vector(std::string &out, Cur &c, int type, uint32_t n)
    {
        if (type == 7) {
            c.need(n);
            size_t len = n;
            while (len && c.p[len - 1] == 0) --len;
            out.append((const char *)c.p, len);
            if (len == 0) out += '.';
            c.p += n;
            return;
        }
        bool first = true;
        for (uint32_t i = 0; i < n; ++i) {
            if (type == 5) {
                c.need(4);
                uint32_t bits;
                memcpy(&bits, c.p, 4);
                c.p += 4;
                if (bits == 0x7F800002u) continue; // end of vector
                if (!first) out += ',';
                put_float(out, bits);
            } else {
                const int32_t v = c.i_of(type);
                if (int_eov(type, v)) continue;
                if (!first) out += ',';
                if (int_missing(type, v)) out += '.';
                else out += std::to_string(v);
            }
            first = false;
        }
        if (first) out += '.';
    }
    bool bcf_next(std::string &line)
    {
        uint32_t len[2];
        if (!raw(len, 8)) return false;
        rec.resize((size_t)len[0] + len[1]);
        if (!rec.empty() && !raw(rec.data(), rec.size())) throw std::runtime_error("BCF: truncated record");
        Cur c{(const unsigned char *)rec.data(), (const unsigned char *)rec.data() + len[0]};
        c.need(24);
        int32_t chrom, p0, rlen;
        uint32_t qual_bits, nai, nfs;
        memcpy(&chrom, c.p, 4); memcpy(&p0, c.p + 4, 4); memcpy(&rlen, c.p + 8, 4); memcpy(&qual_bits, c.p + 12, 4);
        memcpy(&nai, c.p + 16, 4); memcpy(&nfs, c.p + 20, 4);
        c.p += 24;
        const uint32_t n_allele = nai >> 16, n_info = nai & 0xFFFF, n_fmt = nfs >> 24, n_sample = nfs & 0xFFFFFF;
        if (chrom < 0 || (size_t)chrom >= contigs.size()) throw std::runtime_error("BCF: contig index outside the header's ##contig lines");
        line = contigs[(size_t)chrom];
        line += '\t';
        line += std::to_string((long)p0 + 1);
        line += '\t';
        const std::string id = c.typed_str();
        line += id.empty() ? "." : id;
        for (uint32_t a = 0; a < n_allele; ++a) {
            line += a < 2 ? '\t' : ',';
            line += c.typed_str();
        }
        if (n_allele < 2) line += "\t.";
        line += '\t';
        put_float(line, qual_bits);
        line += '\t';
        {
            int t;
            uint32_t n;
            c.desc(t, n);
            if (n == 0) line += '.';
            for (uint32_t i = 0; i < n; ++i) {
                const int32_t v = c.i_of(t);
                if (i) line += ';';
                line += (size_t)v < dict.size() ? dict[(size_t)v] : ".";
            }
        }
        line += '\t';
        if (n_info == 0) line += '.';
        for (uint32_t i = 0; i < n_info; ++i) {
            const int32_t key = c.typed_int();
            if (i) line += ';';
            line += (size_t)key < dict.size() ? dict[(size_t)key] : ".";
            int t;
            uint32_t n;
            c.desc(t, n);
            if (n == 0 || t == 0) continue; // a flag
            line += '=';
            put_vector(line, c, t, n);
        }
        if (n_fmt == 0 || n_sample == 0) return true;
        // individual block: per FORMAT field a key, one descriptor, then n_sample vectors
        Cur d{(const unsigned char *)rec.data() + len[0], (const unsigned char *)rec.data() + rec.size()};
        struct Fmt {
            int key, type;
            uint32_t n;
            const unsigned char *data;
        };
        std::vector<Fmt> fmts;
        line += '\t';
        for (uint32_t i = 0; i < n_fmt; ++i) {
            Fmt fm;
            fm.key = d.typed_int();
            d.desc(fm.type, fm.n);
            const size_t w = fm.type == 1 || fm.type == 7 ? 1 : fm.type == 2 ? 2 : 4;
            fm.data = d.p;
            d.need((size_t)n_sample * fm.n * w);
            d.p += (size_t)n_sample * fm.n * w;
            if (i) line += ':';
            line += (size_t)fm.key < dict.size() ? dict[(size_t)fm.key] : ".";
            fmts.push_back(fm);
        }
        for (uint32_t s_ = 0; s_ < n_sample; ++s_) {
            line += '\t';
            for (size_t i = 0; i < fmts.size(); ++i) {
                const Fmt &fm = fmts[i];
                const size_t w = fm.type == 1 || fm.type == 7 ? 1 : fm.type == 2 ? 2 : 4;
                Cur v{fm.data + (size_t)s_ * fm.n * w, fm.data + (size_t)(s_ + 1) * fm.n * w};
                if (i) line += ':';
                if ((size_t)fm.key < dict.size() && dict[(size_t)fm.key] == "GT") { // (allele + 1) << 1 | phased; 0 = missing
                    bool any = false;
                    for (uint32_t q = 0; q < fm.n; ++q) {
                        const int32_t g = v.i_of(fm.type);
                        if (int_eov(fm.type, g)) break;
                        if (q) line += (g & 1) ? '|' : '/';
                        if ((g >> 1) == 0) line += '.';
                        else line += std::to_string((g >> 1) - 1);
                        any = true;
                    }
                    if (!any) line += '.';
                } else
                    put_vector(line, v, fm.type, fm.n);
            }
        }
        return true;
    }

  public:
    explicit LineReader(const std::string &path) : buf(1 << 20)
    {
        {
            unsigned char h[18];
            size_t mb;
            const int fd = ::open(path.c_str(), O_RDONLY);
            if (fd >= 0 && ::read(fd, h, 18) == 18 && bgzf_header(h, 18, &mb) && lseek(fd, 0, SEEK_SET) == 0 && !getenv("MALVA_GENO_NO_BGZF")) {
                bgzf = true;
                bfd = fd;
            } else if (fd >= 0)
                ::close(fd);
        }
        if (!bgzf) {
            f = gzopen(path.c_str(), "rb");
            if (!f) return;
            gzbuffer(f, 1 << 20);
        }
        if (fill() && end >= 5 && memcmp(buf.data(), "BCF\2\2", 5) == 0) {
            bcf = true;
            pos = 5;
            bcf_open();
        }
    }
    ~LineReader()
    {
        if (f) gzclose(f);
        if (bfd >= 0) ::close(bfd);
    }
    LineReader(const LineReader &) = delete;
    LineReader &operator=(const LineReader &) = delete;
    bool ok() const { return f != nullptr || bgzf; }
    bool is_bcf() const { return bcf; }
    // Text files only: about `want` bytes of whole lines (terminators included; the last line of the file may lack one) appended
    // to `blk`; false at end of file.  One bulk copy per block instead of one string per line: the thread that reads a
    // panel of millions of short records is otherwise what the decoding threads wait for.
    bool next_block(std::string &blk, size_t want)
    {
        const size_t at0 = blk.size();
        if (pos < end) { // what an earlier line-wise read left in the buffer
            const size_t take = std::min(end - pos, want);
            blk.append(buf.data() + pos, take);
            pos += take;
        }
        while (!bgzf && pos == end && blk.size() - at0 < want) { // bulk: straight into the block (zlib hands large requests on a plain file to read())
            const size_t old = blk.size(), ask = std::min<size_t>(want - (old - at0), 1u << 30);
            blk.resize(old + ask);
            const int n = gzread(f, &blk[old], (unsigned)ask);
            blk.resize(old + (n > 0 ? (size_t)n : 0));
            if (n <= 0) break;
        }
        for (;;) {
            if (!bgzf && blk.size() - at0 < want && pos == end) break; // end of file inside the bulk part
            if (pos == end && !fill()) break;
            const size_t have = blk.size() - at0;
            if (have < want) { // (only when the buffer held more than the block wanted)
                const size_t take = std::min(end - pos, want - have);
                blk.append(buf.data() + pos, take);
                pos += take;
                continue;
            }
            if (blk.back() == '\n') break; // ... then up to the end of the line
            const char *nl = (const char *)memchr(buf.data() + pos, '\n', end - pos);
            const size_t take = nl ? (size_t)(nl - (buf.data() + pos)) + 1 : end - pos;
            blk.append(buf.data() + pos, take);
            pos += take;
            if (nl) break;
        }
        return blk.size() > at0;
    }
    // next line without its terminator ('\n' or "\r\n"); false at end of file
    bool next(std::string &line)
    {
        if (bcf) {
            if (bcf_hdr_at < bcf_header.size()) {
                line = bcf_header[bcf_hdr_at++];
                return true;
            }
            return bcf_next(line);
        }
        line.clear();
        for (;;) {
            if (pos == end) {
                if (!fill()) {
                    if (line.empty()) return false;
                    break;
                }
            }
            const char *nl = (const char *)memchr(buf.data() + pos, '\n', end - pos);
            if (nl) {
                line.append(buf.data() + pos, nl - (buf.data() + pos));
                pos = (size_t)(nl - buf.data()) + 1;
                break;
            }
            line.append(buf.data() + pos, end - pos);
            pos = end;
        }
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
    }
};

inline void upper_inplace(std::string &s)
{
    for (char &c : s) c = (char)toupper((unsigned char)c);
}

// main.cpp:283-295: id = first word of the header line, "chr" optionally stripped, sequence upper-cased
struct Reference {
    std::vector<std::string> names;
    std::map<std::string, std::string> seqs;
};
inline bool read_fasta(const std::string &path, bool strip_chr, Reference &out)
{
    LineReader in(path);
    if (!in.ok()) return false;
    std::string line, name;
    std::string *cur = nullptr;
    while (in.next(line)) {
        if (!line.empty() && line[0] == '>') {
            size_t e = 1;
            while (e < line.size() && !isspace((unsigned char)line[e])) ++e;
            name = line.substr(1, e - 1);
            if (strip_chr && name.compare(0, 3, "chr") == 0) name = name.substr(3);
            if (!out.seqs.count(name)) out.names.push_back(name);
            cur = &out.seqs[name];
            cur->clear();
        } else if (cur) { // sequence line: appended whole, upper-cased in place; white space inside a line is rare
            const size_t at = cur->size();
            cur->append(line);
            bool spaces = false;
            char *p = &(*cur)[0];
            for (size_t i = at, e = cur->size(); i < e; ++i) { // toupper / isspace of the "C" locale, inlined
                const unsigned char c = (unsigned char)p[i];
                p[i] = (char)(c >= 'a' && c <= 'z' ? c - 32 : c);
                spaces = spaces || c == ' ' || (c >= '\t' && c <= '\r');
            }
            if (spaces) cur->erase(std::remove_if(cur->begin() + (long)at, cur->end(), [](char c) { return isspace((unsigned char)c) != 0; }), cur->end());
        }
    }
    return true;
}

inline void split(std::string_view s, char sep, std::vector<std::string_view> &out)
{
    out.clear();
    size_t a = 0;
    for (;;) {
        const size_t b = s.find(sep, a);
        if (b == std::string_view::npos) {
            out.push_back(s.substr(a));
            return;
        }
        out.push_back(s.substr(a, b - a));
        a = b + 1;
    }
}

// The record model of variant.hpp:43-62, minus what only printing needs elsewhere.
// A vector whose first N elements live inside the object.  A panel record's per-record containers are tiny -- one or two ALT
// alleles, two or three frequencies and coverages, a handful of genotypes unless the panel is a large one -- and as std::vectors
// they cost five allocations per record on the decoding thread and five frees on whichever thread drops the record: at a million
// records a second the allocator, not the parsing, was what the host loop waited for.
template <class T, size_t N> class SmallVec {
    T *p_;
    size_t n_ = 0, cap_ = N;
    alignas(T) unsigned char buf_[N * sizeof(T)];
    T *inline_buf() { return reinterpret_cast<T *>(buf_); }
    bool is_inline() const { return p_ == reinterpret_cast<const T *>(buf_); }
    void grow(size_t want)
    {
        size_t cap = cap_;
        while (cap < want) cap *= 2;
        T *q = static_cast<T *>(::operator new(cap * sizeof(T)));
        for (size_t i = 0; i < n_; ++i) {
            new (q + i) T(std::move(p_[i]));
            p_[i].~T();
        }
        if (!is_inline()) ::operator delete(p_);
        p_ = q;
        cap_ = cap;
    }
    void take(SmallVec &&o)
    {
        if (o.is_inline()) {
            p_ = inline_buf();
            cap_ = N;
            n_ = o.n_;
            for (size_t i = 0; i < n_; ++i) {
                new (p_ + i) T(std::move(o.p_[i]));
                o.p_[i].~T();
            }
        } else {
            p_ = o.p_;
            cap_ = o.cap_;
            n_ = o.n_;
            o.p_ = o.inline_buf();
            o.cap_ = N;
        }
        o.n_ = 0;
    }

  public:
    typedef T value_type;
    SmallVec() : p_(inline_buf()) {}
    SmallVec(const SmallVec &o) : p_(inline_buf())
    {
        reserve(o.n_);
        for (size_t i = 0; i < o.n_; ++i) new (p_ + i) T(o.p_[i]);
        n_ = o.n_;
    }
    SmallVec(SmallVec &&o) noexcept : p_(inline_buf()) { take(std::move(o)); }
    SmallVec &operator=(const SmallVec &o)
    {
        if (this != &o) {
            clear();
            reserve(o.n_);
            for (size_t i = 0; i < o.n_; ++i) new (p_ + i) T(o.p_[i]);
            n_ = o.n_;
        }
        return *this;
    }
    SmallVec &operator=(SmallVec &&o) noexcept
    {
        if (this != &o) {
            clear();
            if (!is_inline()) ::operator delete(p_);
            take(std::move(o));
        }
        return *this;
    }
    ~SmallVec()
    {
        clear();
        if (!is_inline()) ::operator delete(p_);
    }
    size_t size() const { return n_; }
    bool empty() const { return n_ == 0; }
    T *data() { return p_; }
    const T *data() const { return p_; }
    T *begin() { return p_; }
    T *end() { return p_ + n_; }
    const T *begin() const { return p_; }
    const T *end() const { return p_ + n_; }
    T &operator[](size_t i) { return p_[i]; }
    const T &operator[](size_t i) const { return p_[i]; }
    T &at(size_t i)
    {
        if (i >= n_) throw std::out_of_range("SmallVec::at");
        return p_[i];
    }
    const T &at(size_t i) const
    {
        if (i >= n_) throw std::out_of_range("SmallVec::at");
        return p_[i];
    }
    T &back() { return p_[n_ - 1]; }
    const T &back() const { return p_[n_ - 1]; }
    void reserve(size_t want)
    {
        if (want > cap_) grow(want);
    }
    void clear()
    {
        for (size_t i = 0; i < n_; ++i) p_[i].~T();
        n_ = 0;
    }
    void push_back(const T &v)
    {
        reserve(n_ + 1);
        new (p_ + n_++) T(v);
    }
    void push_back(T &&v)
    {
        reserve(n_ + 1);
        new (p_ + n_++) T(std::move(v));
    }
    template <class... A> T &emplace_back(A &&...a)
    {
        reserve(n_ + 1);
        new (p_ + n_) T(std::forward<A>(a)...);
        return p_[n_++];
    }
    void resize(size_t n)
    {
        reserve(n);
        for (size_t i = n; i < n_; ++i) p_[i].~T();
        for (size_t i = n_; i < n; ++i) new (p_ + i) T();
        n_ = n;
    }
    void assign(size_t n, const T &v)
    {
        clear();
        reserve(n);
        for (size_t i = 0; i < n; ++i) new (p_ + i) T(v);
        n_ = n;
    }
};

struct Variant {
    std::string seq_name;
    int ref_pos = 0; // 0-based
    std::string idx;
    std::string ref_sub;
    SmallVec<std::string, 2> alts; // symbolic <...> alleles dropped (variant.hpp:79-88)
    float quality = NAN;
    SmallVec<std::pair<int, int>, 4> genotypes; // per kept sample
    SmallVec<uint8_t, 8> phasing;
    int ref_size = 0, min_size = 0, max_size = 0;
    bool has_alts = true, is_present = true;
    SmallVec<float, 4> frequencies;
    SmallVec<uint32_t, 4> coverages;
    // VcfReader::defer_genotypes: the sample columns are not decoded by the reader; the record says where they are (a span of
    // the block of text it was cut from, kept alive by the pointer) and the device decodes them (mg_decode_gt_text), after
    // which sp_* hold the kept samples whose genotype word is not sp_default -- the record loop's sparse layout
    std::shared_ptr<const std::string> gt_text;
    uint32_t gt_off = 0, gt_len = 0, n_kept = 0, max_allele = 0;
    int32_t gt_index = -1;
    bool gt_deferred = false;
    uint16_t sp_default = 0;
    uint64_t raw_mask = 0; // raw allele numbers (mod 64) that occur among the kept samples (first alleles only in haploid mode)
    std::vector<uint32_t> sp_sample;
    std::vector<uint16_t> sp_gt;
    size_t n_genotypes() const { return gt_deferred ? n_kept : genotypes.size(); }
    std::string text_prefix; // VcfReader::want_prefix: CHROM .. QUAL of the output record (output_variants, var_block.hpp:337-352), made by the decoding thread

    int n_alleles() const { return (int)alts.size() + 1; }
    const std::string &allele(int i) const { return i == 0 ? ref_sub : alts.at((size_t)i - 1); }
    // variant.hpp:228-240: first allele with this text (duplicated ALT strings collapse)
    int allele_index(const std::string &a) const
    {
        if (ref_sub == a) return 0;
        for (size_t i = 0; i < alts.size(); ++i)
            if (alts[i] == a) return (int)i + 1;
        return -1;
    }
};

// Text VCF: header lines, sample subset (-s), and Variant decode as variant.hpp:66-211
// sees it through htslib.
class VcfReader {
    LineReader in;
    std::string pending;
    bool have_pending = false;
    // per-thread parsing scratch (records are parsed by a pool of threads, see next())
    struct Scratch {
        std::vector<std::string_view> cols, fmt, fld, vals;
        std::vector<int> tok_allele;      // GT tokens of the current record, all kept samples
        std::vector<uint8_t> tok_phased;  // separator in front of each token was '|'
        std::vector<uint32_t> tok_off;    // per kept sample: its first token
        std::shared_ptr<const std::string> block; // the block of text the record is being cut from (pool mode, text files), or null
    };
    std::vector<uint8_t> keep_mask;       // per sample column: kept

    // Records are independent, and on a panel (thousands of sample columns per line) decoding them is what `index`
    // and `call` spend their time on: one thread reads lines into chunks, a pool decodes the chunks, next() hands
    // the records out in file order.  An exception raised by a record is rethrown by next() when that record's
    // turn comes, after every record in front of it.
    struct Chunk {
        std::shared_ptr<std::string> text = std::make_shared<std::string>(); // text files: a block of whole lines, cut into records by the decoding thread ...
        std::vector<std::string> lines; // ... BCF: the records as the reader translated them
        std::vector<Variant> vars;
        size_t n_ok = 0;          // records decoded before `err` (all of them when err is empty)
        std::exception_ptr err;
        bool done = false;
    };
    std::mutex mu;
    std::condition_variable cv;
    std::deque<std::shared_ptr<Chunk>> in_order, todo;
    bool eof = false, stop = false, started = false, threaded = false;
    Scratch inline_scratch;
    std::string inline_line;
    std::thread reader;
    std::vector<std::thread> pool;
    std::shared_ptr<Chunk> cur;
    size_t cur_at = 0;

  public:
    bool defer_genotypes = false; // leave the sample columns to mg_decode_gt_text (pool mode over a text file only; set before the first next())
    // a deferred record decoded on the host after all (an allele number the device's words cannot hold; the host enumerator)
    void genotypes_on_host(Variant &v)
    {
        if (!v.gt_deferred) return;
        const char *at = v.gt_text->data() + v.gt_off;
        parse_samples(v.gt_len ? at : nullptr, at + v.gt_len, v.gt_index, v, inline_scratch); // (no columns and an empty ninth column read the same: every sample ".")
        v.gt_deferred = false;
        v.sp_sample.clear();
        v.sp_gt.clear();
        v.gt_text.reset();
    }
    bool want_prefix = false;    // `call`: fill Variant::text_prefix (set before the first next())
    uint64_t file_bytes = 0;     // size of the file on disk (0 when unknown): decides whether decoding goes to the pool
    std::vector<std::string> header_lines; // the ## lines
    std::vector<std::string> samples;
    std::vector<int> keep; // kept sample columns, VCF order (htslib keeps a mask)
    std::string error;

    VcfReader(const std::string &path, const std::string &samples_opt) : in(path)
    {
        if (!in.ok()) {
            error = "cannot open " + path;
            return;
        }
        {
            struct stat st;
            if (stat(path.c_str(), &st) == 0) file_bytes = (uint64_t)st.st_size;
        }
        std::string line;
        while (in.next(line)) {
            if (line.rfind("##", 0) == 0) header_lines.push_back(line);
            else if (!line.empty() && line[0] == '#') {
                std::vector<std::string_view> cols;
                split(line, '\t', cols);
                for (size_t i = 9; i < cols.size(); ++i) samples.emplace_back(cols[i]);
                break;
            } else {
                pending = line;
                have_pending = true;
                break;
            }
        }
        if (samples_opt == "-") {
            for (size_t i = 0; i < samples.size(); ++i) keep.push_back((int)i);
        } else {
            std::ifstream sf(samples_opt);
            if (!sf) {
                error = "ERROR: VCF samples subset (cannot open " + samples_opt + ")";
                return;
            }
            std::set<std::string> want;
            std::string s;
            while (std::getline(sf, s)) {
                while (!s.empty() && isspace((unsigned char)s.back())) s.pop_back();
                if (!s.empty()) want.insert(s);
            }
            for (const auto &w : want)
                if (std::find(samples.begin(), samples.end(), w) == samples.end()) {
                    error = "ERROR: VCF samples subset (unknown sample " + w + ")";
                    return;
                }
            for (size_t i = 0; i < samples.size(); ++i)
                if (want.count(samples[i])) keep.push_back((int)i);
        }
    }
    bool ok() const { return error.empty(); }

  private:
    // one record line -> Variant; throws std::runtime_error on records the reference would crash on
    void parse(const std::string &line, Variant &v, const std::string &freq_key, bool uniform, Scratch &sc) const
    {
        parse(line.data(), line.size(), v, freq_key, uniform, sc);
    }
    // the record's text as a span with text[size] == '\0'
    void parse(const char *text, size_t size, Variant &v, const std::string &freq_key, bool uniform, Scratch &sc) const
    {
        decode(text, size, v, freq_key, uniform, sc);
        if (want_prefix && v.has_alts) make_prefix(v);
    }
    void decode(const char *text, size_t size, Variant &v, const std::string &freq_key, bool uniform, Scratch &sc) const
    {
        auto &cols = sc.cols;
        auto &fmt = sc.fmt;
        auto &fld = sc.fld;
        auto &vals = sc.vals;
        // the nine fixed columns; the sample columns (tens of thousands on a panel) are walked in place below
        cols.clear();
        const char *const line_end = text + size;
        const char *samples_at = nullptr;
        for (const char *p = text;;) {
            const char *t = (const char *)memchr(p, '\t', (size_t)(line_end - p));
            cols.emplace_back(p, (size_t)((t ? t : line_end) - p));
            if (!t) break;
            p = t + 1;
            if (cols.size() == 9) {
                samples_at = p;
                break;
            }
        }
        if (cols.size() < 8) throw std::runtime_error("malformed VCF record: " + std::string(text, std::min<size_t>(size, 60)));
        v = Variant();
        v.seq_name = std::string(cols[0]);
        v.ref_pos = atoi(std::string(cols[1]).c_str()) - 1;
        v.idx = std::string(cols[2]);
        v.ref_sub = std::string(cols[3]);
        upper_inplace(v.ref_sub);
        v.ref_size = (int)v.ref_sub.size();
        if (cols[4] != ".") {
            split(cols[4], ',', fld);
            for (auto a : fld)
                if (!a.empty() && a[0] != '<') {
                    v.alts.emplace_back(a);
                    upper_inplace(v.alts.back());
                }
        }
        v.coverages.assign(v.alts.size() + 1, 0);
        v.quality = cols[5] == "." ? NAN : (float)strtod(std::string(cols[5]).c_str(), nullptr);
        // set_sizes, variant.hpp:108-124
        if (v.alts.empty()) v.has_alts = false;
        else {
            v.min_size = v.max_size = v.ref_size;
            for (const auto &a : v.alts) {
                v.min_size = std::min(v.min_size, (int)a.size());
                v.max_size = std::max(v.max_size, (int)a.size());
            }
        }
        if (!v.has_alts) return;
        // extract_frequencies, variant.hpp:126-156
        if (!uniform) {
            std::vector<float> raw;
            bool found = false;
            if (cols[7] != ".") {
                split(cols[7], ';', fld);
                for (auto kv : fld)
                    if (kv.size() > freq_key.size() && kv.compare(0, freq_key.size(), freq_key) == 0 && kv[freq_key.size()] == '=') {
                        split(kv.substr(freq_key.size() + 1), ',', vals);
                        for (auto x : vals) raw.push_back(x == "." ? NAN : (float)strtod(std::string(x).c_str(), nullptr));
                        found = true;
                        break;
                    }
            }
            if (!found) throw std::runtime_error("INFO key " + freq_key + " missing at " + v.seq_name + ":" + std::string(cols[1]) +
                                                 " (the reference dereferences a null pointer here)");
            if (raw.size() < v.alts.size())
                throw std::runtime_error("fewer " + freq_key + " values than ALT alleles at " + v.seq_name + ":" + std::string(cols[1]));
            v.frequencies.assign(1, 0.f);
            for (size_t i = 0; i < v.alts.size(); ++i) v.frequencies.push_back(raw[i]);
            double acc = 0.0;
            for (float f : v.frequencies) acc += f;
            v.frequencies[0] = (float)(1.0 - acc);
            if (v.frequencies[0] < 0) v.frequencies[0] = 0.0f;
        } else {
            const float u = (float)(1.0 / (double)(v.alts.size() + 1));
            v.frequencies.assign(v.alts.size() + 1, u);
        }
        if (v.frequencies[0] == 1.0) v.is_present = false;
        if (!v.is_present) return;
        // extract_genotypes, variant.hpp:158-211
        int gi = -1;
        if (cols.size() > 8) {
            split(cols[8], ':', fmt);
            for (size_t i = 0; i < fmt.size(); ++i)
                if (fmt[i] == "GT") gi = (int)i;
        }
        if (gi < 0 || keep.empty()) {
            v.has_alts = false; // variant.hpp:169-174
            return;
        }
        if (defer_genotypes && sc.block) { // left to the device: where the sample columns are
            v.gt_deferred = true;
            v.gt_text = sc.block;
            v.gt_off = (uint32_t)((samples_at ? samples_at : line_end) - sc.block->data());
            v.gt_len = (uint32_t)(samples_at ? line_end - samples_at : 0);
            v.gt_index = gi;
            v.n_kept = (uint32_t)keep.size();
            return;
        }
        parse_samples(samples_at, line_end, gi, v, sc);
    }
    // the sample columns [samples_at, line_end) (NUL at line_end; samples_at null: none) -> v.genotypes / v.phasing
    void parse_samples(const char *samples_at, const char *line_end, int gi, Variant &v, Scratch &sc) const
    {
        auto &tok_allele = sc.tok_allele;
        auto &tok_phased = sc.tok_phased;
        auto &tok_off = sc.tok_off;
        // what bcf_get_genotypes returns: per sample `ploidy` values, short samples padded with vector_end.
        // One flat token list for the record (panels carry tens of thousands of samples: no per-sample allocation).
        struct G {
            int allele; // -1 missing, -2 vector_end
            bool phased;
        };
        // Token arrays are written by index (two slots per kept sample to start with; the general path grows them).
        size_t cap = std::max(tok_allele.size(), 2 * keep.size() + 16);
        tok_allele.resize(cap);
        tok_phased.resize(cap);
        tok_off.resize(keep.size() + 1);
        tok_off[0] = 0;
        size_t nt = 0, ns = 0, ploidy = 0;
        auto put = [&](int val, bool ph) {
            if (nt == cap) {
                cap *= 2;
                tok_allele.resize(cap);
                tok_phased.resize(cap);
            }
            tok_allele[nt] = val;
            tok_phased[nt++] = ph ? 1 : 0;
        };
        auto close_sample = [&]() {
            tok_off[++ns] = (uint32_t)nt;
            ploidy = std::max(ploidy, (size_t)(tok_off[ns] - tok_off[ns - 1]));
        };
        auto one_char = [](char t) { return t >= '0' && t <= '9' ? t - '0' : t == '.' ? -1 : 0; }; // atoi of a 1-char token
        auto ends_gt = [](char t) { return t == '\t' || t == ':' || t == '\0'; };                  // the line is NUL-terminated
        const char *c = samples_at;
        for (size_t col = 0; col < samples.size(); ++col) {
            if (!c) { // record with fewer sample columns than the header: GT "."
                if (keep_mask[col]) {
                    put(-1, false);
                    close_sample();
                }
                continue;
            }
            if (keep_mask[col]) {
                if (gi == 0 && c[0] != '\t' && c[0] != '\0' && ends_gt(c[1])) { // "0"
                    put(one_char(c[0]), false);
                    c += 1;
                } else if (gi == 0 && c[0] != '\t' && c[0] != '\0' && c[0] != ':' && (c[1] == '|' || c[1] == '/') && !ends_gt(c[2]) && c[2] != '|' &&
                           c[2] != '/' && ends_gt(c[3])) { // "0|1"
                    put(one_char(c[0]), false);
                    put(one_char(c[2]), c[1] == '|');
                    c += 3;
                } else {
                    int sub = 0;
                    while (sub < gi && c < line_end && *c != '\t') { // the gi-th ':'-separated sub-field
                        if (*c == ':') ++sub;
                        ++c;
                    }
                    if (sub < gi) put(-1, false); // sub-field absent: GT "."
                    else {
                        bool ph = false;
                        for (;;) { // tokens separated by '/' or '|'; each read as atoi would (sign, leading digits)
                            const char *a = c;
                            while (c < line_end && *c != '/' && *c != '|' && *c != ':' && *c != '\t') ++c;
                            int val = -1;
                            if (c > a && !(c == a + 1 && *a == '.')) {
                                const char *q = a;
                                bool neg = false;
                                if (*q == '-' || *q == '+') neg = *q++ == '-';
                                long acc = 0;
                                while (q < c && *q >= '0' && *q <= '9') acc = acc * 10 + (*q++ - '0');
                                val = (int)(neg ? -acc : acc);
                            }
                            put(val, ph);
                            if (c < line_end && (*c == '/' || *c == '|')) {
                                ph = *c == '|';
                                ++c;
                                continue;
                            }
                            break;
                        }
                    }
                }
                close_sample();
            }
            while (c < line_end && *c != '\t') ++c; // the rest of the column (nothing, when it holds only GT)
            c = c < line_end ? c + 1 : nullptr;
        }
        const size_t n_flat = keep.size() * ploidy;
        const bool uniform_ploidy = nt == n_flat; // every sample has `ploidy` tokens: the padded array IS the token list
        auto flat_at = [&](size_t idx) -> G { // element idx of the padded [sample][ploidy] array
            if (uniform_ploidy) return G{tok_allele[idx], tok_phased[idx] != 0};
            const size_t smp = idx / ploidy, j = idx % ploidy;
            const uint32_t o = tok_off[smp], cnt = tok_off[smp + 1] - o;
            return j < cnt ? G{tok_allele[o + j], tok_phased[o + j] != 0} : G{-2, false};
        };
        v.genotypes.resize(keep.size());
        v.phasing.resize(keep.size());
        for (size_t i = 0; i < keep.size(); ++i) {
            const G first = flat_at(i * ploidy);
            // curr_gt[1]: with ploidy 1 this is the NEXT sample's value (variant.hpp:184); past the last
            // sample the reference reads beyond the array -- treated as vector_end here.
            const G second = i * ploidy + 1 < n_flat ? flat_at(i * ploidy + 1) : G{-2, false};
            int a1, a2;
            bool is_ph;
            if (second.allele == -2) {
                a1 = a2 = first.allele;
                is_ph = true;
            } else {
                a1 = first.allele;
                a2 = second.allele;
                is_ph = second.phased;
            }
            v.genotypes[i] = {a1 < 0 ? 0 : a1, a2 < 0 ? 0 : a2};
            v.phasing[i] = is_ph ? 1 : 0;
        }
    }
    static void make_prefix(Variant &v)
    {
        std::string &s = v.text_prefix;
        char num[64];
        s.reserve(v.seq_name.size() + v.idx.size() + v.ref_sub.size() + 40);
        s = v.seq_name;
        snprintf(num, sizeof num, "\t%d\t", v.ref_pos + 1);
        s += num;
        s += v.idx;
        s += '\t';
        s += v.ref_sub;
        s += '\t';
        for (size_t i = 0; i < v.alts.size(); ++i) {
            if (i) s += ',';
            s += v.alts[i];
        }
        s += '\t';
        if (std::isnan(v.quality)) s += '.';
        else { // ostream << float, default precision
            snprintf(num, sizeof num, "%g", v.quality);
            s += num;
        }
    }

    void start(const std::string &freq_key, bool uniform)
    {
        started = true;
        keep_mask.assign(samples.size(), 0);
        for (int i : keep) keep_mask[(size_t)i] = 1;
        const unsigned n_threads = std::max(1u, std::min(std::thread::hardware_concurrency(), 16u));
        const size_t max_in_flight = 2 * n_threads;
        reader = std::thread([this, max_in_flight]() {
            std::string line;
            for (;;) {
                auto ch = std::make_shared<Chunk>();
                size_t bytes = 0;
                if (!in.is_bcf()) {
                    if (have_pending) {
                        *ch->text = pending;
                        *ch->text += '\n';
                        have_pending = false;
                    }
                    // (deferred genotypes: one device call decodes a block's records, so a block holds hundreds of panel lines)
                    const size_t want = defer_genotypes ? 16u << 20 : 1u << 20;
                    ch->text->reserve(want + (64u << 10));
                    in.next_block(*ch->text, want);
                }
                while (in.is_bcf() && ch->lines.size() < 2048 && bytes < (8u << 20)) {
                    if (have_pending) {
                        line.swap(pending);
                        have_pending = false;
                    } else if (!in.next(line))
                        break;
                    if (line.empty() || line[0] == '#') continue;
                    bytes += line.size();
                    ch->lines.emplace_back(std::move(line));
                    line.clear();
                }
                std::unique_lock<std::mutex> lk(mu);
                if (ch->lines.empty() && ch->text->empty()) {
                    eof = true;
                    cv.notify_all();
                    return;
                }
                cv.wait(lk, [&] { return in_order.size() < max_in_flight || stop; });
                if (stop) return;
                in_order.push_back(ch);
                todo.push_back(ch);
                cv.notify_all();
            }
        });
        for (unsigned t = 0; t < n_threads; ++t)
            pool.emplace_back([this, freq_key, uniform]() {
                Scratch sc;
                for (;;) {
                    std::shared_ptr<Chunk> ch;
                    {
                        std::unique_lock<std::mutex> lk(mu);
                        cv.wait(lk, [&] { return !todo.empty() || eof || stop; });
                        if (stop || todo.empty()) return; // eof and nothing left
                        ch = todo.front();
                        todo.pop_front();
                    }
                    size_t i = 0;
                    try {
                        if (!ch->text->empty()) { // lines cut in place: each terminator becomes the NUL the decoder stops at
                            char *p = &(*ch->text)[0], *const e = p + ch->text->size();
                            ch->vars.reserve(ch->text->size() / 40 < 4096 ? ch->text->size() / 40 + 1 : 4096);
                            sc.block = ch->text;
                            while (p < e) {
                                char *nl = (char *)memchr(p, '\n', (size_t)(e - p));
                                char *le = nl ? nl : e;
                                size_t n = (size_t)(le - p);
                                if (n && p[n - 1] == '\r') --n;
                                p[n] = '\0'; // (at the very end of the block this is the string's own terminator)
                                if (n && p[0] != '#') {
                                    ch->vars.emplace_back();
                                    parse(p, n, ch->vars.back(), freq_key, uniform, sc);
                                    ++i;
                                }
                                p = le + 1;
                            }
                        } else {
                            ch->vars.resize(ch->lines.size());
                            for (; i < ch->lines.size(); ++i) parse(ch->lines[i], ch->vars[i], freq_key, uniform, sc);
                        }
                    } catch (...) {
                        ch->err = std::current_exception();
                        if (!ch->text->empty()) ch->vars.pop_back(); // the record that threw
                    }
                    sc.block.reset();
                    ch->n_ok = i;
                    ch->lines.clear();
                    ch->lines.shrink_to_fit();
                    ch->text.reset(); // (records with deferred genotypes hold on to it)
                    std::lock_guard<std::mutex> lk(mu);
                    ch->done = true;
                    cv.notify_all();
                }
            });
    }

  public:
    // Start decoding now (pool mode only), before the first next(): a caller with other start-up work to do -- the devices, the
    // index, the sample's table -- finds the first 2 x threads blocks decoded when it comes for them.  Same arguments as next().
    void decode_ahead(const std::string &freq_key, bool uniform)
    {
        if (started) return;
        threaded = keep.size() >= 32 || file_bytes >= (8u << 20);
        if (const char *e = getenv("MALVA_GENO_VCF_POOL")) threaded = atoi(e) != 0;
        if (threaded) start(freq_key, uniform);
    }
    // false at end of file; throws what parse() threw for the record whose turn it is
    bool next(Variant &v, const std::string &freq_key, bool uniform)
    {
        if (!started) {
            // a handful of sample columns: decoding a record costs less than handing it between threads (measured:
            // 0.53 us inline, 0.88 us through the pool); a panel's line is tens of kilobytes and the pool pays
            // -- unless the file is large: then the thread that calls next() has the batches to build, and every
            // microsecond of decoding it does not do itself shortens the run (1e6 records x 2 samples: 0.36 s of 0.83)
            threaded = keep.size() >= 32 || file_bytes >= (8u << 20);
            if (const char *e = getenv("MALVA_GENO_VCF_POOL")) threaded = atoi(e) != 0; // tests force either path
            if (threaded) start(freq_key, uniform);
            else {
                started = true;
                keep_mask.assign(samples.size(), 0);
                for (int i : keep) keep_mask[(size_t)i] = 1;
            }
        }
        if (!threaded) {
            for (;;) {
                if (have_pending) {
                    inline_line.swap(pending);
                    have_pending = false;
                } else if (!in.next(inline_line))
                    return false;
                if (!inline_line.empty() && inline_line[0] != '#') break;
            }
            parse(inline_line, v, freq_key, uniform, inline_scratch);
            return true;
        }
        for (;;) {
            if (cur) {
                if (cur_at < cur->n_ok) {
                    v = std::move(cur->vars[cur_at++]);
                    return true;
                }
                if (cur->err) {
                    const std::exception_ptr e = cur->err;
                    cur->err = nullptr; // a caller that goes on after the exception continues with the next chunk
                    std::rethrow_exception(e);
                }
                cur.reset();
            }
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return (!in_order.empty() && in_order.front()->done) || (eof && in_order.empty()); });
            if (in_order.empty()) return false;
            cur = in_order.front();
            in_order.pop_front();
            cur_at = 0;
            cv.notify_all(); // room for the reader
        }
    }

    ~VcfReader()
    {
        {
            std::lock_guard<std::mutex> lk(mu);
            stop = true;
            cv.notify_all();
        }
        if (reader.joinable()) reader.join();
        for (auto &t : pool)
            if (t.joinable()) t.join();
    }
    VcfReader(const VcfReader &) = delete;
    VcfReader &operator=(const VcfReader &) = delete;
};

// print_cleaned_header (main.cpp:190-219) as htslib renders it
inline std::string cleaned_header(const std::vector<std::string> &lines_in, bool verbose)
{
    std::vector<std::string> lines = lines_in;
    auto has_id = [&](const std::string &prefix) {
        for (const auto &l : lines)
            if (l.compare(0, prefix.size(), prefix) == 0 && (l[prefix.size()] == ',' || l[prefix.size()] == '>')) return true;
        return false;
    };
    if (!has_id("##FILTER=<ID=PASS")) {
        const size_t at = (!lines.empty() && lines[0].rfind("##fileformat", 0) == 0) ? 1 : 0;
        lines.insert(lines.begin() + (long)at, "##FILTER=<ID=PASS,Description=\"All filters passed\">");
    }
    if (!has_id("##FORMAT=<ID=GT")) lines.push_back("##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">");
    if (!has_id("##FORMAT=<ID=GQ")) lines.push_back("##FORMAT=<ID=GQ,Number=1,Type=Integer,Description=\"Genotype Quality\">");
    if (verbose) {
        if (!has_id("##INFO=<ID=COVS")) lines.push_back("##INFO=<ID=COVS,Number=R,Type=Integer,Description=\"Allele coverages\">");
        if (!has_id("##INFO=<ID=GTS")) lines.push_back("##INFO=<ID=GTS,Number=.,Type=String,Description=\"Genotypes Likelihood\">");
    }
    std::string out;
    for (const auto &l : lines) out += l + "\n";
    out += "#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\tDONOR\n";
    return out;
}

} // namespace malva
