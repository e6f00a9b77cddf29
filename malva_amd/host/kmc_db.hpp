// KMC database files (<db>.kmc_pre / <db>.kmc_suf) opened for listing: what CKMCFile::OpenForListing + Info give
// main.cpp:444-449, 482-484.  The KMC API is a third-party library the reference links (-lkmc, README.md:23) and is
// not part of its checkout; this reader follows KMC's published database layout (KMC >= 2, "0x200" format):
//
//   <db>.kmc_pre  "KMCP" | u64 table[n_bins * 4^lut_prefix_len] | u32 signature_map[4^signature_len + 1] |
//                 header (its size in the byte at file end - 8; its last field is the u32 version 0x200) |
//                 u32 header size | "KMCP"
//                 header: u32 k, mode, counter_size, lut_prefix_len, signature_len, min_count, max_count(low),
//                         u64 total_kmers, u8 !both_strands, 3 bytes, u32 max_count(high), padding, u32 version
//   <db>.kmc_suf  "KMCS" | total_kmers records of (k - lut_prefix_len) / 4 suffix bytes + counter_size counter bytes | "KMCS"
//
// Nothing is decoded here: the table and the raw records go to the device (mg_kmc_set_lut / mg_kmc_scan_records),
// which rebuilds each k-mer from its record and the table entry that covers the record's index.
#pragma once
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace malva {

struct KmcDb {
    uint32_t k = 0, mode = 0, counter_size = 0, lut_prefix_len = 0, signature_len = 0, min_count = 0;
    uint64_t max_count = 0, total = 0;
    bool both_strands = true;
    uint32_t suffix_bytes = 0, rec_bytes = 0;
    std::vector<uint64_t> lut;
    const unsigned char *records = nullptr; // total * rec_bytes, inside the mapping

    KmcDb() = default;
    KmcDb(const KmcDb &) = delete;
    KmcDb &operator=(const KmcDb &) = delete;
    ~KmcDb()
    {
        if (map_) munmap(map_, map_len_);
    }

    static bool present(const std::string &prefix)
    {
        struct stat a, b;
        return stat((prefix + ".kmc_pre").c_str(), &a) == 0 && S_ISREG(a.st_mode) && stat((prefix + ".kmc_suf").c_str(), &b) == 0 && S_ISREG(b.st_mode);
    }

    void open(const std::string &prefix)
    {
        std::vector<unsigned char> pre;
        {
            FILE *f = fopen((prefix + ".kmc_pre").c_str(), "rb");
            if (!f) throw std::runtime_error("cannot open " + prefix + ".kmc_pre");
            fseek(f, 0, SEEK_END);
            const long sz = ftell(f);
            rewind(f);
            pre.resize(sz > 0 ? (size_t)sz : 0);
            const size_t got = pre.empty() ? 0 : fread(pre.data(), 1, pre.size(), f);
            fclose(f);
            if (got != pre.size()) throw std::runtime_error("cannot read " + prefix + ".kmc_pre");
        }
        const size_t n = pre.size();
        if (n < 4 + 8 + 12 || memcmp(pre.data(), "KMCP", 4) != 0 || memcmp(pre.data() + n - 4, "KMCP", 4) != 0)
            throw std::runtime_error(prefix + ".kmc_pre is not a KMC prefix file");
        const uint32_t version = rd<uint32_t>(pre, n - 12);
        if (version != 0x200) throw std::runtime_error("KMC database version " + std::to_string(version) + ": only the KMC 2/3 format (0x200) is read");
        const size_t header_size = pre[n - 8]; // the KMC API reads one byte here
        if (header_size < 44 || header_size + 8 + 4 > n) throw std::runtime_error(prefix + ".kmc_pre: bad header size");
        const size_t h0 = n - 8 - header_size;
        k = rd<uint32_t>(pre, h0);
        mode = rd<uint32_t>(pre, h0 + 4);
        counter_size = rd<uint32_t>(pre, h0 + 8);
        lut_prefix_len = rd<uint32_t>(pre, h0 + 12);
        signature_len = rd<uint32_t>(pre, h0 + 16);
        min_count = rd<uint32_t>(pre, h0 + 20);
        max_count = rd<uint32_t>(pre, h0 + 24);
        total = rd<uint64_t>(pre, h0 + 28);
        both_strands = pre[h0 + 36] == 0;
        max_count |= (uint64_t)rd<uint32_t>(pre, h0 + 40) << 32;
        if (mode != 0) throw std::runtime_error("KMC database in Quake mode (float counters) is not supported");
        if (lut_prefix_len < 1 || lut_prefix_len > 15 || signature_len > 12 || k <= lut_prefix_len || (k - lut_prefix_len) % 4 != 0)
            throw std::runtime_error(prefix + ".kmc_pre: implausible header (k " + std::to_string(k) + ", prefix " + std::to_string(lut_prefix_len) + ")");
        suffix_bytes = (k - lut_prefix_len) / 4;
        rec_bytes = suffix_bytes + counter_size;
        const size_t sig_bytes = 4 * ((1ULL << (2 * signature_len)) + 1), single = 1ULL << (2 * lut_prefix_len);
        if (h0 < 4 + sig_bytes + 8 * single || (h0 - 4 - sig_bytes) % (8 * single) != 0)
            throw std::runtime_error(prefix + ".kmc_pre: the prefix table is not a whole number of 4^" + std::to_string(lut_prefix_len) + "-entry bins");
        lut.resize((h0 - 4 - sig_bytes) / 8);
        memcpy(lut.data(), pre.data() + 4, lut.size() * 8);

        const int fd = ::open((prefix + ".kmc_suf").c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + prefix + ".kmc_suf");
        struct stat st;
        if (fstat(fd, &st) != 0 || (uint64_t)st.st_size != 8 + total * rec_bytes) {
            close(fd);
            throw std::runtime_error(prefix + ".kmc_suf does not hold " + std::to_string(total) + " records of " + std::to_string(rec_bytes) + " bytes");
        }
        map_len_ = (size_t)st.st_size;
        map_ = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (map_ == MAP_FAILED) {
            map_ = nullptr;
            throw std::runtime_error("cannot map " + prefix + ".kmc_suf");
        }
        madvise(map_, map_len_, MADV_SEQUENTIAL);
        const unsigned char *b = (const unsigned char *)map_;
        if (memcmp(b, "KMCS", 4) != 0 || memcmp(b + map_len_ - 4, "KMCS", 4) != 0) throw std::runtime_error(prefix + ".kmc_suf is not a KMC suffix file");
        records = b + 4;
    }

  private:
    void *map_ = nullptr;
    size_t map_len_ = 0;
    template <class T> static T rd(const std::vector<unsigned char> &v, size_t off)
    {
        T x;
        memcpy(&x, v.data() + off, sizeof x);
        return x;
    }
};

} // namespace malva
