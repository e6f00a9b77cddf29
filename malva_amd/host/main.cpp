// malva-geno index|call -- host driver of the MI355X-native hot path.
//
// Keeps the reference's command line (argument_parser.hpp:51-159), control flow
// (index_main main.cpp:251-419, call_main :421-594) and VCF output, and hands
// every k-mer store operation, both scans, coverage and likelihoods to
// libmalva_hip.so through include/malva_hip.h.  There is no CPU implementation of
// those in this program: without a GPU it stops at mg_create.
#include <fcntl.h>
#include <future>
#include <getopt.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <deque>
#include <iostream>
#include <mutex>
#include <sstream>
#include <thread>

#include "block.hpp"
#include "index_file.hpp"
#include "kmc_db.hpp"
extern "C" {
#include "malva_hip.h"
}

using namespace malva;

namespace {

const char *USAGE =
    "Usage: malva-geno <index|call> [-k KMER-SIZE] [-r REF-KMER-SIZE] [-c MAX-COV] "
    "<reference.fa> <variants.vcf> <kmc_output_prefix>\n"
    "\n"
    "      -h, --help                        display this help and exit\n"
    "      -k, --kmer-size                   size of the kmers to index (default:35)\n"
    "      -r, --ref-kmer-size               size of the reference kmers to index (default:43)\n"
    "      -e, --error-rate                  expected sample error rate (default:0.001)\n"
    "      -s, --samples                     file containing the list of (VCF) samples to consider (default:-, i.e. all samples)\n"
    "      -f, --freq-key                    a priori frequency key in the INFO column of the input VCF (default:AF)\n"
    "      -c, --max-coverage                maximum coverage for variant alleles (default:200)\n"
    "      -b, --bf-size                     bloom filter size in GB (default:4)\n"
    "      -p, --strip-chr                   strip \"chr\" from sequence names (default:false)\n"
    "      -u, --uniform                     use uniform a priori probabilities (default:false)\n"
    "      -v, --verbose                     output COVS and GTS in INFO column (default: false)\n"
    "      -1, --haploid                     run MALVA in haploid mode (default: false)\n"
    "      -d, --device                      first GPU to run on (default:0)                     [this build]\n"
    "      -g, --gpus                        GPUs to use, devices -d .. -d+N-1 (default:1)       [this build; N > 1 has run\n"
    "                                        only with all contexts on ONE device: untested on multi-GPU hardware]\n"
    "                                        call: the k-mer table is sharded over them, the per-allele counters\n"
    "                                        are all-reduced over RCCL, the variants are split between them\n"
    "\n"
    "  <kmc_output_prefix>: a KMC database (<prefix>.kmc_pre + <prefix>.kmc_suf, KMC 2/3 format), read directly;\n"
    "  or <prefix>.txt / <prefix> holding `kmc_tools transform <db> dump` text (one `KMER<tab>count` per line).\n"
    "  index file: <variants.vcf>.c<r>.k<k>.malvax.zst, the reference's container (sdsl + zstd); `call` also reads this\n"
    "  build's compact <...>.malvax.hipz (written when MALVA_GENO_INDEX_FORMAT=hipz).\n"
    "  extra sub-commands (no GPU needed): dump-kmers (signature k-mers of every block); index-convert <fa> <vcf> zst|hipz\n"
    "\n"
    "  Verified inputs: the text k-mer dump and text / gzip / bgzip VCF (checked against the reference's own example).  The three\n"
    "  BINARY formats -- a KMC database, a BCF panel, the reference's sdsl + zstd index container -- are read and written from their\n"
    "  published layouts and checked against an independent second implementation only: no file written by KMC, bcftools or the\n"
    "  reference binary was available to this build (formats UNPINNED).  When in doubt convert: `kmc_tools transform <db> dump`,\n"
    "  `bcftools view -Ov`, and let this build write its own index.\n";

struct Options { // argument_parser.hpp:51-66
    unsigned k = 35, ref_k = 43;
    float error_rate = 0.001f;
    std::string samples = "-", freq_key = "AF";
    unsigned max_coverage = 200;
    uint64_t bf_size = 1ULL << 35;
    bool strip_chr = false, uniform = false, verbose = false, haploid = false;
    int device = 0, gpus = 1;
    std::string fasta_path, vcf_path, kmc_path;
};

bool parse_arguments(int argc, char **argv, Options &o)
{
    // same short options and long names as the reference, including its quirks: --strip-chr / --uniform are
    // declared with required_argument and --haploid is spelt "haplod" (argument_parser.hpp:70-84)
    static const option longopts[] = {{"kmer-size", required_argument, nullptr, 'k'},   {"ref-kmer-size", required_argument, nullptr, 'r'},
                                      {"error-rate", required_argument, nullptr, 'e'},  {"freq-key", required_argument, nullptr, 'f'},
                                      {"samples", required_argument, nullptr, 's'},     {"max-coverage", required_argument, nullptr, 'c'},
                                      {"bf-size", required_argument, nullptr, 'b'},     {"strip-chr", required_argument, nullptr, 'p'},
                                      {"uniform", required_argument, nullptr, 'u'},     {"verbose", no_argument, nullptr, 'v'},
                                      {"haplod", no_argument, nullptr, '1'},            {"haploid", no_argument, nullptr, '1'},
                                      {"device", required_argument, nullptr, 'd'},      {"help", no_argument, nullptr, 'h'},
                                      {"gpus", required_argument, nullptr, 'g'},
                                      {nullptr, 0, nullptr, 0}};
    bool die = false;
    optind = 1;
    for (int c; (c = getopt_long(argc, argv, "k:r:e:s:f:c:b:d:g:hpuv1", longopts, nullptr)) != -1;) {
        std::istringstream arg(optarg ? optarg : "");
        switch (c) {
        case 'p': o.strip_chr = true; break;
        case 'u': o.uniform = true; break;
        case 'k': arg >> o.k; break;
        case 'r': arg >> o.ref_k; break;
        case 'e': arg >> o.error_rate; break;
        case 's': arg >> o.samples; break;
        case 'f': arg >> o.freq_key; break;
        case 'c': arg >> o.max_coverage; break;
        case 'b':
            arg >> o.bf_size;
            o.bf_size *= 1ULL << 33; // "GB" on the command line, 2^33 bits each (argument_parser.hpp:119-123)
            break;
        case 'd': arg >> o.device; break;
        case 'g': arg >> o.gpus; break;
        case 'v': o.verbose = true; break;
        case '1': o.haploid = true; break;
        case '?': die = true; break;
        case 'h': std::cout << USAGE; exit(EXIT_SUCCESS);
        }
    }
    if (argc - optind < 3) {
        std::cerr << "malva : missing arguments\n";
        die = true;
    } else if (argc - optind > 3) {
        std::cerr << "malva : too many arguments\n";
        die = true;
    }
    if (o.gpus < 1 || o.gpus > 64) {
        std::cerr << "malva : --gpus must be 1..64\n";
        die = true;
    }
    if (die) {
        std::cerr << "\n" << USAGE;
        return false;
    }
    o.fasta_path = argv[optind++];
    o.vcf_path = argv[optind++];
    o.kmc_path = argv[optind++];
    if (const char *bits = getenv("MALVA_GENO_BF_BITS")) // tests: filters smaller than -b's 2^33-bit granule
        if (atoll(bits) > 0) o.bf_size = (uint64_t)atoll(bits);
    return true;
}

// MALVA_GENO_TIMERS=1: where the host threads spend their time, summed per label, printed at exit (profiles/*cli*)
struct Timers {
    bool on = getenv("MALVA_GENO_TIMERS") != nullptr;
    std::mutex mu;
    std::map<std::string, double> acc;
    void add(const char *what, double s)
    {
        std::lock_guard<std::mutex> lk(mu);
        acc[what] += s;
    }
    ~Timers()
    {
        if (!on) return;
        for (const auto &kv : acc) fprintf(stderr, "[malva-geno/timer] %-28s %.3fs\n", kv.first.c_str(), kv.second);
    }
};
Timers g_timers;
struct Timed {
    const char *what;
    std::chrono::steady_clock::time_point t0;
    explicit Timed(const char *w) : what(w), t0(std::chrono::steady_clock::now()) {}
    ~Timed()
    {
        if (g_timers.on) g_timers.add(what, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
    }
};

// pelapsed(), main.cpp:93-115
auto t_start = std::chrono::steady_clock::now();
auto t_last = t_start;
void pelapsed(const std::string &s, bool rollback = false)
{
    const auto now = std::chrono::steady_clock::now();
    rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    char buf[512];
    snprintf(buf, sizeof buf, "[malva-geno/%s] Execution Time %.4gs\n[malva-geno/%s] Time elapsed %.4gs\n[malva-geno/%s] Used CPU-time elapsed %.4gs\n"
                              "[malva-geno/%s] Maximum memory used %ldMb\n",
             s.c_str(), std::chrono::duration<double>(now - t_last).count(), s.c_str(), std::chrono::duration<double>(now - t_start).count(), s.c_str(),
             ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6, s.c_str(), ru.ru_maxrss / 1024);
    std::cerr << buf << (rollback ? "\r" : "\n");
    t_last = now;
}

struct Device {
    mg_ctx *ctx = nullptr;
    std::mutex mu; // calls on one context are serialised by the caller (include/malva_hip.h): held by whoever drives the context from a second thread
    ~Device() { mg_destroy(ctx); }
    void check(int rc, const char *what)
    {
        if (rc != MG_OK) throw std::runtime_error(std::string(what) + ": " + mg_last_error(ctx));
    }
};

constexpr size_t STRIDE = 136; // MG_MAX_KMER + NUL, rounded to 8

// fixed-stride ASCII rows for the batch calls
struct Rows {
    std::vector<char> data;
    size_t n = 0;
    void add(const std::string &kmer)
    {
        if (kmer.empty() || kmer.size() > MG_MAX_KMER) throw std::runtime_error("signature k-mer of length " + std::to_string(kmer.size()) + " (1.." +
                                                                                std::to_string(MG_MAX_KMER) + " supported)");
        data.resize(data.size() + STRIDE, 0);
        memcpy(&data[n * STRIDE], kmer.data(), kmer.size());
        ++n;
    }
    void clear()
    {
        data.clear();
        n = 0;
    }
};

bool file_exists(const std::string &p)
{
    struct stat st;
    return stat(p.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}
// main.cpp:407 / :456: <vcf>.c<ref_k>.k<k>.malvax + ".zst" (the reference's container) or ".hipz" (this build's compact one)
std::string index_path(const Options &o, const char *suffix)
{
    return o.vcf_path + ".c" + std::to_string(o.ref_k) + ".k" + std::to_string(o.k) + ".malvax" + suffix;
}

// ---- genotypes the reader left to the device (VcfReader::defer_genotypes) ---------------------------------------------------
// A panel of a thousand samples or more has its sample columns decoded on the device (MALVA_GENO_GT_DEVICE=0 / 1 forces either way;
// the reader defers only where it can: a text file read through the pool).
inline bool device_gt_wanted(const VcfReader &vcf)
{
    if (const char *e = getenv("MALVA_GENO_GT_DEVICE")) return atoi(e) != 0;
    return vcf.keep.size() >= 1024;
}
// The records of one cut batch, block of text by block of text: mg_decode_gt_text over the spans, the entries back into the records.
inline void decode_deferred(std::vector<Variant> &kept, Device &dev, VcfReader &vcf, const Options &o)
{
    std::vector<uint64_t> off, mask;
    std::vector<uint32_t> len, sp_off, mx, ss;
    std::vector<int32_t> gi;
    std::vector<uint16_t> sg;
    std::vector<uint8_t> keep_cols(vcf.samples.size(), 0);
    for (int i : vcf.keep) keep_cols[(size_t)i] = 1;
    for (size_t a = 0; a < kept.size();) {
        if (!kept[a].gt_deferred) {
            ++a;
            continue;
        }
        size_t b = a;
        while (b < kept.size() && kept[b].gt_deferred && kept[b].gt_text == kept[a].gt_text) ++b;
        const size_t n = b - a;
        off.resize(n); len.resize(n); gi.resize(n); sp_off.resize(n + 1); mask.resize(n); mx.resize(n);
        for (size_t r = 0; r < n; ++r) {
            off[r] = kept[a + r].gt_off;
            len[r] = kept[a + r].gt_len;
            gi[r] = kept[a + r].gt_index;
        }
        uint16_t dflt = 0;
        uint64_t n_entries = 0;
        {
            Timed t("main: mg_decode_gt_text (+ lock)");
            std::lock_guard<std::mutex> lk(dev.mu);
            const std::string &text = *kept[a].gt_text;
            dev.check(mg_decode_gt_text(dev.ctx, text.data(), text.size(), n, off.data(), len.data(), gi.data(), (uint32_t)keep_cols.size(), keep_cols.data(), o.haploid,
                                        &dflt, sp_off.data(), mask.data(), mx.data(), &n_entries),
                      "mg_decode_gt_text");
            ss.resize(n_entries);
            sg.resize(n_entries);
            dev.check(mg_decode_gt_entries(dev.ctx, ss.data(), sg.data()), "mg_decode_gt_entries");
        }
        for (size_t r = 0; r < n; ++r) {
            Variant &v = kept[a + r];
            v.sp_default = dflt;
            v.raw_mask = mask[r];
            v.max_allele = mx[r];
            if (mx[r] > 127) { // an allele number the 7-bit words cannot hold: this record is decoded here after all
                vcf.genotypes_on_host(v);
                continue;
            }
            v.sp_sample.assign(ss.begin() + sp_off[r], ss.begin() + sp_off[r + 1]);
            v.sp_gt.assign(sg.begin() + sp_off[r], sg.begin() + sp_off[r + 1]);
            v.gt_text.reset(); // (the block of text goes when its last record has let go)
        }
        a = b;
    }
}
// v.genotypes / v.phasing of a deferred record, for the host enumerator (a block the device handed back, dump-kmers' cousin paths)
inline void genotypes_from_entries(Variant &v)
{
    if (!v.gt_deferred) return;
    const auto pair_of = [](uint16_t w) { return std::make_pair((int)(w & 127), (int)((w >> 7) & 127)); };
    v.genotypes.assign(v.n_kept, pair_of(v.sp_default));
    v.phasing.assign(v.n_kept, (uint8_t)((v.sp_default >> 14) & 1));
    for (size_t e = 0; e < v.sp_sample.size(); ++e) {
        v.genotypes[v.sp_sample[e]] = pair_of(v.sp_gt[e]);
        v.phasing[v.sp_sample[e]] = (uint8_t)((v.sp_gt[e] >> 14) & 1);
    }
    v.gt_deferred = false;
    v.sp_sample.clear();
    v.sp_gt.clear();
}
inline void genotypes_from_entries(Block &b)
{
    for (Variant &v : b.vars) genotypes_from_entries(v);
}
// build_alleles_combs on a chain of one (var_block.hpp:734-786): the (canonical) alleles some kept haplotype carries
inline uint64_t carried_mask(const Variant &v, bool haploid)
{
    uint64_t mask = 0;
    if (v.gt_deferred) {
        if (v.max_allele >= (uint32_t)v.n_alleles()) (void)v.alts.at(v.max_allele); // (throws what the loop below would)
        for (int a = 0; a < v.n_alleles() && a < 64; ++a)
            if ((v.raw_mask >> a) & 1) mask |= 1ULL << v.allele_index(v.allele(a));
        return mask;
    }
    for (size_t g = 0; g < v.genotypes.size(); ++g) {
        mask |= 1ULL << v.allele_index(v.allele(v.genotypes[g].first));
        if (!haploid) mask |= 1ULL << v.allele_index(v.allele(v.genotypes[g].second));
    }
    return mask;
}

// The record loop shared by index_main (main.cpp:309-370) and call_main (:522-579).  on_block(block, reference
// of `last_seq_name`) is called for every closed block.  Keeps the reference's control flow, including that
// last_seq_name is refreshed only when a block is flushed.
// `cutter` (optional): the device that makes the cuts (mg_cut_blocks) -- the kept records are then collected a batch at a
// time and the loop below is replayed over the device's block offsets; without it (dump-kmers, MALVA_GENO_HOST_CUT=1) the
// cuts are made record by record on the host.  Same blocks either way (tests/test_gpu_cli.py runs both).
template <class F> size_t for_each_block(VcfReader &vcf, const Options &o, const Reference &refs, bool for_index, std::vector<std::string> *used, F on_block,
                                         Device *cutter = nullptr)
{
    static const std::string empty;
    std::string ref_name;            // (a panel names a few dozen sequences over millions of records: the last answer is nearly always the next one)
    const std::string *ref_seq = nullptr;
    auto ref_of = [&](const std::string &name) -> const std::string & {
        if (!ref_seq || name != ref_name) {
            auto it = refs.seqs.find(name);
            ref_seq = it == refs.seqs.end() ? &empty : &it->second;
            ref_name = name;
        }
        return *ref_seq;
    };
    if (cutter && !getenv("MALVA_GENO_HOST_CUT")) {
        // records per cut batch; MALVA_GENO_CUT_BATCH exists so tests can put a batch seam inside every block
        const size_t batch_max = getenv("MALVA_GENO_CUT_BATCH") ? (size_t)std::max(1L, atol(getenv("MALVA_GENO_CUT_BATCH"))) : 100000;
        Block vb((int)o.k);          // the open block: the records since the last cut
        std::string block_name;      // `last_seq_name` of the reference loop: the name the open block will be flushed under
        uint32_t last_cid = 0;       // contig id the open block's last record presents to the next one
        std::map<std::string, uint32_t> ids;
        auto id_of = [&](const std::string &name) { return ids.emplace(name, (uint32_t)ids.size()).first->second; };
        std::vector<Variant> kept;
        std::vector<int32_t> pos;
        std::vector<uint32_t> ref_size, min_size, cid, off;
        size_t i = 0, cells = 0, n_cut_blocks = 0, n_batches = 0;
        bool more = true, seen_kept = false;
        Variant v;
        while (more) {
            kept.clear();
            cells = 0;
            Timed *t_parse = new Timed("main: vcf.next");
            while (kept.size() < batch_max && cells < (64u << 20) && (more = vcf.next(v, o.freq_key, o.uniform))) {
                ++i;
                if (i % 5000 == 0) pelapsed("Processed " + std::to_string(i) + " variants", true);
                if (block_name.empty()) { // the file's first record names the first block, kept or not (main.cpp:319-323)
                    block_name = v.seq_name;
                    if (used) used->push_back(block_name);
                }
                if (for_index ? (!v.has_alts || !v.is_present) : !v.has_alts) continue;
                cells += v.n_genotypes();
                kept.push_back(std::move(v));
            }
            delete t_parse;
            if (kept.empty()) continue;
            decode_deferred(kept, *cutter, vcf, o);
            const bool carry = !vb.empty();
            const size_t n = kept.size() + (carry ? 1 : 0);
            pos.resize(n); ref_size.resize(n); min_size.resize(n); cid.resize(n); off.resize(n + 1);
            if (carry) {
                const Variant &c = vb.vars.back();
                pos[0] = c.ref_pos; ref_size[0] = (uint32_t)c.ref_size; min_size[0] = (uint32_t)c.min_size; cid[0] = last_cid;
            }
            for (size_t j = 0; j < kept.size(); ++j) {
                const Variant &r = kept[j];
                const size_t q = j + (carry ? 1 : 0);
                pos[q] = r.ref_pos; ref_size[q] = (uint32_t)r.ref_size; min_size[q] = (uint32_t)r.min_size;
                // the very first kept record is added to an empty block unseen; what the NEXT record compares its name with
                // is `last_seq_name`, still the file's first name then
                cid[q] = seen_kept ? id_of(r.seq_name) : id_of(block_name);
                seen_kept = true;
            }
            size_t nb = 0;
            {
                Timed t_cut("main: mg_cut_blocks (+ lock)");
                std::lock_guard<std::mutex> lk(cutter->mu);
                cutter->check(mg_cut_blocks(cutter->ctx, n, pos.data(), ref_size.data(), min_size.data(), cid.data(), off.data(), &nb), "mg_cut_blocks");
            }
            last_cid = cid[n - 1];
            ++n_batches;
            size_t b = 0; // off[b] = next block start at or after the current element
            Timed t_blocks("main: blocks -> batches");
            for (size_t q = carry ? 1 : 0; q < n; ++q) {
                while (b < nb && off[b] < q) ++b;
                const bool cut = b < nb && off[b] == q;
                Variant &r = kept[q - (carry ? 1 : 0)];
                if (cut && !vb.empty()) {
                    on_block(vb, block_name, ref_of(block_name));
                    ++n_cut_blocks;
                    vb.clear();
                    if (block_name != r.seq_name) {
                        block_name = r.seq_name;
                        if (used) used->push_back(block_name);
                    }
                }
                vb.add(std::move(r));
            }
        }
        if (!vb.empty()) {
            on_block(vb, block_name, ref_of(block_name));
            ++n_cut_blocks;
            vb.clear();
        }
        std::cerr << "[malva-geno] " << n_cut_blocks << " block(s) cut on the device in " << n_batches << " batch(es)" << std::endl;
        return i;
    }
    Block vb((int)o.k);
    std::string last_seq_name;
    Variant v;
    size_t i = 0;
    while (vcf.next(v, o.freq_key, o.uniform)) {
        ++i;
        if (i % 5000 == 0) pelapsed("Processed " + std::to_string(i) + " variants", true);
        if (last_seq_name.empty()) {
            last_seq_name = v.seq_name;
            if (used) used->push_back(last_seq_name);
        }
        if (for_index ? (!v.has_alts || !v.is_present) : !v.has_alts) continue;
        if (vb.empty()) {
            vb.add(std::move(v));
            continue;
        }
        if (!vb.near_to_last(v) || last_seq_name != v.seq_name) {
            on_block(vb, last_seq_name, ref_of(last_seq_name));
            vb.clear();
            if (last_seq_name != v.seq_name) {
                last_seq_name = v.seq_name;
                if (used) used->push_back(last_seq_name);
            }
        }
        vb.add(std::move(v));
    }
    if (!vb.empty()) {
        on_block(vb, last_seq_name, ref_of(last_seq_name));
        vb.clear();
    }
    return i;
}

// ---- the panel genotypes of a batch of blocks, as the device takes them ---------------------------------------------------
// Dense: one [variant][sample] matrix of a1 | a2 << 7 | phased << 14.  Sparse (panels of more than SPARSE_GT_SAMPLES samples:
// nearly every genotype of such a panel is 0|0): per variant only the samples that carry anything else, in sample order --
// what crosses PCIe for the 27,934-sample SARS-CoV-2 panel drops from 56 KB per record to a few bytes per carrier.
constexpr uint32_t SPARSE_GT_SAMPLES = 64;
struct PanelGenotypes {
    bool sparse = false;
    std::vector<uint16_t> gt;                 // dense
    std::vector<uint32_t> sp_off{0}, sp_sample; // sparse
    std::vector<uint16_t> sp_gt;
    uint16_t sp_default = (uint16_t)(1u << 14);
    size_t bytes() const { return sparse ? 4 * sp_off.size() + 6 * sp_sample.size() : 2 * gt.size(); }
};
// one sample's genotype as the device takes it: a1 | a2 << 7 | phased << 14
inline uint16_t genotype_word(const Variant &v, size_t s_, bool haploid)
{
    const auto &p2 = v.genotypes[s_];
    if (p2.first >= v.n_alleles() || p2.second >= v.n_alleles())
        throw std::runtime_error("GT allele beyond the kept ALT list at " + v.seq_name + ":" + std::to_string(v.ref_pos + 1) +
                                 " (the reference reads out of bounds here)");
    if (haploid) return (uint16_t)(p2.first | (1u << 14));
    return (uint16_t)(p2.first | (p2.second << 7) | ((v.phasing[s_] ? 1 : 0) << 14));
}
inline PanelGenotypes pack_genotypes(const std::vector<const Variant *> &vars, size_t n_vars, uint32_t n_samples, bool haploid)
{
    PanelGenotypes g;
    g.sparse = n_samples > SPARSE_GT_SAMPLES;
    for (const Variant *vp : vars) g.sparse = g.sparse || vp->gt_deferred; // (records decoded on the device arrive in the sparse form whatever the panel's size)
    if (!g.sparse) g.gt.assign(n_vars * n_samples, 0);
    // haploid mode reads the first allele only (var_block.hpp:751): the word is reduced to it, so that a ploidy-1 panel --
    // whose second "allele" is whatever htslib's layout puts behind the first -- is as sparse as it looks
    auto word = [&](const Variant &v, size_t s_) -> uint16_t { return genotype_word(v, s_, haploid); };
    if (g.sparse) { // the default word: 0|0 phased or 0/0 unphased, whichever the batch holds more of (a sample of it decides)
        size_t phased0 = 0, unphased0 = 0, seen = 0;
        for (const Variant *vp : vars) {
            const Variant &v = *vp;
            if (v.gt_deferred) { // decoded on the device: its batch's default stands for its samples
                (v.sp_default ? phased0 : unphased0) += 64;
                seen += 64;
            } else
                for (size_t s_ = 0; s_ < v.genotypes.size() && seen < 200000; s_ += 7, ++seen) {
                    const uint16_t w = word(v, s_);
                    phased0 += w == (1u << 14);
                    unphased0 += w == 0;
                }
            if (seen >= 200000) break;
        }
        g.sp_default = unphased0 > phased0 ? 0 : (uint16_t)(1u << 14);
    }
    size_t row = 0;
    for (const Variant *vp : vars)
        {
            const Variant &v = *vp;
            if (v.gt_deferred) { // already the sparse layout (mg_decode_gt_text), against its own batch's default
                if (!g.sparse) throw std::runtime_error("internal: deferred genotypes on a panel of few samples");
                if (v.max_allele >= (uint32_t)v.n_alleles())
                    throw std::runtime_error("GT allele beyond the kept ALT list at " + v.seq_name + ":" + std::to_string(v.ref_pos + 1) +
                                             " (the reference reads out of bounds here)");
                if (v.sp_default == g.sp_default) {
                    g.sp_sample.insert(g.sp_sample.end(), v.sp_sample.begin(), v.sp_sample.end());
                    g.sp_gt.insert(g.sp_gt.end(), v.sp_gt.begin(), v.sp_gt.end());
                } else { // (a panel that mixes phased and unphased records: the samples WITHOUT an entry are the ones that differ here)
                    size_t e = 0;
                    for (uint32_t s_ = 0; s_ < v.n_kept; ++s_) {
                        uint16_t w = v.sp_default;
                        if (e < v.sp_sample.size() && v.sp_sample[e] == s_) w = v.sp_gt[e++];
                        if (w != g.sp_default) {
                            g.sp_sample.push_back(s_);
                            g.sp_gt.push_back(w);
                        }
                    }
                }
                g.sp_off.push_back((uint32_t)g.sp_sample.size());
                ++row;
                continue;
            }
            for (size_t s_ = 0; s_ < v.genotypes.size(); ++s_) {
                const uint16_t w = word(v, s_);
                if (!g.sparse) g.gt[row * n_samples + s_] = w;
                else if (w != g.sp_default) {
                    g.sp_sample.push_back((uint32_t)s_);
                    g.sp_gt.push_back(w);
                }
            }
            if (g.sparse) g.sp_off.push_back((uint32_t)g.sp_sample.size());
            ++row;
        }
    return g;
}

inline PanelGenotypes pack_genotypes(const std::vector<Block> &blocks, size_t n_vars, uint32_t n_samples, bool haploid)
{
    std::vector<const Variant *> vars;
    vars.reserve(n_vars);
    for (const Block &b : blocks)
        for (const Variant &v : b.vars) vars.push_back(&v);
    return pack_genotypes(vars, n_vars, n_samples, haploid);
}
inline PanelGenotypes pack_genotypes(const std::vector<Variant> &records, size_t n_vars, uint32_t n_samples, bool haploid)
{
    std::vector<const Variant *> vars;
    vars.reserve(records.size());
    for (const Variant &v : records) vars.push_back(&v);
    return pack_genotypes(vars, n_vars, n_samples, haploid);
}

// ---- index file (index_file.hpp): payload out of / into the contexts ---------------------------------------------------
void save_index(Device &dev, const Options &o)
{
    IndexPayload p;
    const int which[2] = {MG_BF_CTX, MG_BF_ALT}; // payload order of main.cpp:409-411: context_bf, bf, ref_bf
    for (int i = 0; i < 2; ++i) {
        uint64_t size, nset;
        int mode;
        dev.check(mg_bf_info(dev.ctx, which[i], &size, &nset, &mode), "mg_bf_info");
        p.filt[i].mode = (uint64_t)mode;
        p.filt[i].pos.resize(nset);
        p.filt[i].cnt.resize(nset);
        dev.check(mg_bf_export_sparse(dev.ctx, which[i], p.filt[i].pos.data(), p.filt[i].cnt.data()), "mg_bf_export_sparse");
    }
    uint64_t nkeys = 0;
    dev.check(mg_map_size(dev.ctx, &nkeys), "mg_map_size");
    p.stride = STRIDE;
    p.rows.resize(nkeys * STRIDE);
    p.vals.resize(nkeys);
    if (nkeys) dev.check(mg_map_export(dev.ctx, p.rows.data(), STRIDE, p.vals.data()), "mg_map_export");
    // Both containers by default: the reference's (.zst: one zstd stream over the filters' full bit vectors -- gigabytes of
    // zeros through a single-threaded codec, ~1.5 s to write and ~1.2 s to read at -b 4) for the reference's own binary, and
    // this build's sparse one (.hipz, milliseconds), which `call` prefers while it is not older than the .zst beside it.
    // MALVA_GENO_INDEX_FORMAT=zst | hipz writes one of them only.  Each is written under a temporary name and renamed: a run
    // that dies mid-write leaves no truncated index for `call` to find.
    const char *fmt = getenv("MALVA_GENO_INDEX_FORMAT");
    const std::string want = fmt ? fmt : "both";
    if (want != "both" && want != "zst" && want != "hipz") throw std::runtime_error("MALVA_GENO_INDEX_FORMAT: zst, hipz or both");
    auto write_zst = [&]() {
        const std::string final_path = index_path(o, ".zst");
        if (want == "hipz") {
            unlink(final_path.c_str()); // no stale index of the other kind beside the new one
            return;
        }
        const std::string tmp_path = final_path + ".tmp." + std::to_string((long)getpid());
        try {
            Timed t("index file: .zst");
            save_index_zst(tmp_path, p, o.bf_size);
        } catch (...) {
            unlink(tmp_path.c_str());
            throw;
        }
        if (rename(tmp_path.c_str(), final_path.c_str()) != 0) {
            unlink(tmp_path.c_str());
            throw std::runtime_error("cannot write " + final_path);
        }
    };
    // side by side (the .zst takes seconds, single-threaded inside libzstd); the .hipz is renamed into place after the .zst,
    // so it is the newer file of the two
    auto zst_job = std::async(std::launch::async, write_zst);
    std::exception_ptr hipz_err;
    const std::string hipz_final = index_path(o, ".hipz"), hipz_tmp = hipz_final + ".tmp." + std::to_string((long)getpid());
    if (want == "zst") unlink(hipz_final.c_str());
    else {
        try {
            Timed t("index file: .hipz");
            save_index_hipz(hipz_tmp, p, o.k, o.ref_k, o.bf_size, hipz::panel_tie_of(o.vcf_path));
        } catch (...) {
            unlink(hipz_tmp.c_str());
            hipz_err = std::current_exception();
        }
    }
    try {
        zst_job.get();
    } catch (...) {
        unlink(hipz_tmp.c_str());
        throw;
    }
    if (hipz_err) std::rethrow_exception(hipz_err);
    if (want != "zst") {
        if (rename(hipz_tmp.c_str(), hipz_final.c_str()) != 0) {
            unlink(hipz_tmp.c_str());
            throw std::runtime_error("cannot write " + hipz_final);
        }
        utimensat(AT_FDCWD, hipz_final.c_str(), nullptr, 0); // (a rename keeps the time of the last write, which was before the .zst's)
    }
}

// every device of a multi-GPU call holds the whole index (the read-only structures are replicated):
// the file is read once and imported into all contexts side by side
template <class F> void on_all_devices(std::vector<Device> &devs, F f)
{
    if (devs.size() == 1) return f(devs[0], 0);
    std::vector<std::exception_ptr> errs(devs.size());
    std::vector<std::thread> pool;
    for (size_t d = 0; d < devs.size(); ++d)
        pool.emplace_back([&, d]() {
            try {
                f(devs[d], d);
            } catch (...) {
                errs[d] = std::current_exception();
            }
        });
    for (auto &t : pool) t.join();
    for (auto &e : errs)
        if (e) std::rethrow_exception(e);
}

// the index file -> payload (host only: runs beside the devices' start-up), then payload -> every context
void read_index(const Options &o, IndexPayload &p)
{
    Timed t("startup: index file -> payload");
    const std::string zst = index_path(o, ".zst"), hipz = index_path(o, ".hipz");
    struct stat sz, sh;
    const bool has_zst = stat(zst.c_str(), &sz) == 0, has_hipz = stat(hipz.c_str(), &sh) == 0;
    auto newer = [](const struct stat &a, const struct stat &b) { // a strictly newer than b
        return a.st_mtim.tv_sec != b.st_mtim.tv_sec ? a.st_mtim.tv_sec > b.st_mtim.tv_sec : a.st_mtim.tv_nsec > b.st_mtim.tv_nsec;
    };
    // the sparse container unless the reference's one is newer (an index brought over from the reference's binary) -- or unless the
    // sparse one says it was built from another panel than the VCF given here (its header holds that VCF's size and modification
    // time: file dates alone do not survive a copy or a restore) while the reference's container is there to be read instead
    bool stale = false;
    if (has_hipz && has_zst) {
        const hipz::PanelTie tie = index_hipz_tie(hipz);
        stale = tie.known() && !(tie == hipz::panel_tie_of(o.vcf_path));
        if (stale) std::cerr << "[malva-geno] " << hipz << " was built from another version of " << o.vcf_path << ": reading " << zst << std::endl;
    }
    if (has_hipz && !stale && !(has_zst && newer(sz, sh))) load_index_hipz(hipz, p, o.k, o.ref_k, o.bf_size, STRIDE);
    else if (has_zst) load_index_zst(zst, p, o.bf_size, STRIDE);
    else throw std::runtime_error("cannot open index " + index_path(o, ".zst") + " (run `malva-geno index` with the same -k -r -b first)");
}
void import_index(std::vector<Device> &devs, const Options &o, const IndexPayload &p)
{
    Timed t("startup: payload -> device");
    on_all_devices(devs, [&](Device &dev, size_t) {
        const int which[2] = {MG_BF_CTX, MG_BF_ALT};
        for (int i = 0; i < 2; ++i)
            dev.check(mg_bf_import_sparse(dev.ctx, which[i], (int)p.filt[i].mode, o.bf_size, p.filt[i].pos.data(), p.filt[i].cnt.data(), p.filt[i].pos.size()),
                      "mg_bf_import_sparse");
        if (!p.vals.empty()) dev.check(mg_map_import(dev.ctx, p.rows.data(), STRIDE, p.vals.size(), p.vals.data()), "mg_map_import");
    });
}

// ---------------------------------------------------------------------------------------------------------------
int index_main(const Options &o)
{
    Reference refs;
    if (!read_fasta(o.fasta_path, o.strip_chr, refs)) {
        std::cerr << "ERROR: cannot open " << o.fasta_path << std::endl;
        return 1;
    }
    VcfReader vcf(o.vcf_path, o.samples);
    if (!vcf.ok()) {
        std::cerr << vcf.error << std::endl;
        return 1;
    }
    vcf.defer_genotypes = device_gt_wanted(vcf) && !getenv("MALVA_GENO_HOST_CUT"); // (the decode rides on the device cutter's batches)
    vcf.decode_ahead(o.freq_key, o.uniform);
    pelapsed("Reference processed");
    Device dev;
    if (mg_create(&dev.ctx, o.device, o.k, o.ref_k, o.bf_size) != MG_OK) {
        std::cerr << "ERROR: no usable MI355X/HIP device " << o.device << " (or not enough memory for two filters of " << o.bf_size
                  << " bits); this build has no CPU path" << std::endl;
        return 1;
    }
    Rows ref_rows, alt_rows;
    auto flush = [&](bool force) {
        if (ref_rows.n && (force || ref_rows.n >= (1u << 20))) {
            dev.check(mg_map_insert(dev.ctx, ref_rows.data.data(), STRIDE, ref_rows.n), "mg_map_insert");
            ref_rows.clear();
        }
        if (alt_rows.n && (force || alt_rows.n >= (1u << 20))) {
            dev.check(mg_bf_insert(dev.ctx, MG_BF_ALT, alt_rows.data.data(), STRIDE, alt_rows.n), "mg_bf_insert");
            alt_rows.clear();
        }
    };
    std::vector<std::string> used;
    // General blocks are enumerated ON THE DEVICE, a few thousand at a time (mg_index_blocks: chains, distinct haplotype
    // picks, signature assembly, exact-map insert / filter bit -- extract_kmers + add_kmers_to_bf, main.cpp:349-350);
    // blocks the device hands back (a capacity exceeded, a window clipped by a contig end, a REF k-mer with a base
    // outside ACGT) are enumerated by all host cores and inserted through the batch calls.
    std::map<std::string, uint64_t> contig_base;
    {
        std::string all;
        for (const auto &name : refs.names) {
            contig_base[name] = all.size();
            all += refs.seqs.at(name);
        }
        dev.check(mg_reference_upload(dev.ctx, all.data(), all.size()), "mg_reference_upload");
    }
    const bool host_only = o.k > MG_MAX_PACKED_K || getenv("MALVA_GENO_HOST_ENUM"); // the variable forces the host enumerator (tests)
    std::vector<Block> waiting;
    std::vector<const std::string *> waiting_ref;
    std::vector<uint64_t> waiting_base;
    size_t waiting_cells = 0, n_host_blocks = 0, n_general_blocks = 0;
    auto enumerate_waiting = [&]() {
        const size_t nb = waiting.size();
        if (!nb) return;
        n_general_blocks += nb;
        std::vector<uint8_t> redo(nb, host_only ? 1 : 0);
        if (!host_only) {
            std::vector<uint64_t> blk_base;
            std::vector<uint32_t> blk_len, blk_var_off{0}, ref_size, min_size, var_allele_off{0}, allele_off{0};
            std::vector<int32_t> ipos;
            std::vector<uint8_t> is_present, canon;
            std::vector<char> pool;
            bool device_ok = true;
            for (size_t b = 0; b < nb; ++b) {
                blk_base.push_back(waiting_base[b]);
                blk_len.push_back((uint32_t)waiting_ref[b]->size());
                for (const Variant &v : waiting[b].vars) {
                    const uint32_t A = (uint32_t)v.n_alleles();
                    if (A > 127) device_ok = false;
                    ipos.push_back(v.ref_pos);
                    ref_size.push_back((uint32_t)v.ref_size);
                    min_size.push_back((uint32_t)v.min_size);
                    is_present.push_back(v.is_present);
                    for (uint32_t a = 0; a < A; ++a) {
                        const std::string &al = v.allele((int)a);
                        pool.insert(pool.end(), al.begin(), al.end());
                        allele_off.push_back((uint32_t)pool.size());
                        canon.push_back((uint8_t)std::min(255, v.allele_index(al)));
                    }
                    var_allele_off.push_back(var_allele_off.back() + A);
                }
                blk_var_off.push_back((uint32_t)ipos.size());
            }
            const size_t nv = ipos.size();
            const uint32_t n_samples = (uint32_t)vcf.keep.size();
            const PanelGenotypes pg = pack_genotypes(waiting, nv, n_samples, o.haploid);
            std::vector<uint8_t> overflow(nv, 1);
            if (device_ok && pg.sparse)
                dev.check(mg_index_blocks_sparse(dev.ctx, nb, blk_base.data(), blk_len.data(), blk_var_off.data(), nv, ipos.data(), ref_size.data(), min_size.data(),
                                                 is_present.data(), var_allele_off.data(), allele_off.data(), pool.data(), pool.size(), canon.data(), pg.sp_off.data(),
                                                 pg.sp_sample.data(), pg.sp_gt.data(), pg.sp_default, n_samples, o.haploid, overflow.data()),
                          "mg_index_blocks_sparse");
            else if (device_ok)
                dev.check(mg_index_blocks(dev.ctx, nb, blk_base.data(), blk_len.data(), blk_var_off.data(), nv, ipos.data(), ref_size.data(), min_size.data(),
                                          is_present.data(), var_allele_off.data(), allele_off.data(), pool.data(), pool.size(), canon.data(), pg.gt.data(),
                                          n_samples, o.haploid, overflow.data()),
                          "mg_index_blocks");
            for (size_t b = 0; b < nb; ++b)
                for (uint32_t v = blk_var_off[b]; v < blk_var_off[b + 1]; ++v) redo[b] = redo[b] || overflow[v];
        }
        std::vector<size_t> todo;
        for (size_t b = 0; b < nb; ++b)
            if (redo[b]) todo.push_back(b);
        n_host_blocks += todo.size();
        const size_t nt = todo.size();
        std::vector<std::vector<AlleleSignatures>> sigs(nt);
        const unsigned n_threads = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), 16, std::max<size_t>(nt, 1)}));
        std::vector<std::exception_ptr> errs(n_threads);
        std::vector<size_t> err_at(n_threads, SIZE_MAX);
        std::vector<std::thread> pool_t;
        for (unsigned t = 0; t < n_threads && nt; ++t)
            pool_t.emplace_back([&, t]() {
                for (size_t q = t; q < nt; q += n_threads) {
                    try {
                        genotypes_from_entries(waiting[todo[q]]);
                        sigs[q] = waiting[todo[q]].extract(*waiting_ref[todo[q]], o.haploid); // main.cpp:349
                    } catch (...) {
                        errs[t] = std::current_exception();
                        err_at[t] = q;
                        return;
                    }
                }
            });
        for (auto &th : pool_t) th.join();
        size_t first = SIZE_MAX;
        for (unsigned t = 0; t < n_threads; ++t)
            if (errs[t] && (first == SIZE_MAX || err_at[t] < err_at[first])) first = t;
        if (first != SIZE_MAX) std::rethrow_exception(errs[first]);
        for (size_t q = 0; q < nt; ++q) {
            for (const auto &per_allele : sigs[q]) // add_kmers_to_bf, main.cpp:122-144
                for (const auto &as : per_allele)
                    for (const auto &sig : as.second)
                        for (const auto &kmer : sig) (as.first == 0 ? ref_rows : alt_rows).add(kmer);
            flush(false);
        }
        waiting.clear();
        waiting_ref.clear();
        waiting_base.clear();
        waiting_cells = 0;
    };
    // Blocks of one variant whose alleles are all shorter than k -- nearly every block of a SNP panel -- are indexed ON THE
    // DEVICE too, a batch at a time (mg_index_isolated: signature assembly from the uploaded reference, exact-map insert /
    // filter bit); the host keeps the few the device hands back (a base outside ACGT in the window) and those whose
    // right flank a contig end clips (the reference then makes a shorter k-mer, var_block.hpp:187).
    struct LoneBatch {
        std::vector<uint64_t> pos, present;
        std::vector<uint32_t> var_allele_off{0}, allele_off{0};
        std::vector<char> pool;
        std::vector<uint8_t> flags;
        std::vector<Block> blocks; // kept for the variants handed back
        std::vector<const std::string *> refs;
        size_t cells = 0;
        size_t n() const { return pos.size(); }
    } lone;
    size_t n_lone_device = 0, n_lone_host = 0;
    auto index_lone = [&]() {
        if (!lone.n()) return;
        std::vector<uint8_t> overflow(lone.n(), 1);
        dev.check(mg_index_isolated(dev.ctx, lone.n(), lone.pos.data(), lone.var_allele_off.data(), lone.allele_off.data(), lone.pool.data(), lone.pool.size(),
                                    lone.present.data(), lone.flags.data(), overflow.data()),
                  "mg_index_isolated");
        for (size_t v = 0; v < lone.n(); ++v) {
            if (!overflow[v]) {
                ++n_lone_device;
                continue;
            }
            ++n_lone_host;
            genotypes_from_entries(lone.blocks[v]);
            lone.blocks[v].extract_lone(*lone.refs[v], o.haploid, [&](int a, const std::string &kmer) { (a == 0 ? ref_rows : alt_rows).add(kmer); });
            flush(false);
        }
        lone = LoneBatch();
    };
    const size_t n = for_each_block(vcf, o, refs, true, &used, [&](Block &vb, const std::string &seq_name, const std::string &reference) {
        if (vb.is_lone_short()) {
            const Variant &v = vb.vars[0];
            const bool on_device = !host_only && contig_base.count(seq_name) && v.n_alleles() <= 64 &&
                                   (long)v.ref_pos + v.ref_size + (long)(o.k + 1) / 2 <= (long)reference.size();
            if (!on_device) { // one k-mer per carried allele, no containers
                genotypes_from_entries(vb);
                vb.extract_lone(reference, o.haploid, [&](int a, const std::string &kmer) { (a == 0 ? ref_rows : alt_rows).add(kmer); });
                flush(false);
                return;
            }
            const bool eligible = v.is_present && v.ref_pos >= (int)o.k && v.ref_pos <= (int)reference.size() - (int)o.k; // var_block.hpp:104
            const uint64_t mask = eligible ? carried_mask(v, o.haploid) : 0;
            lone.pos.push_back(contig_base.at(seq_name) + (uint64_t)std::max(v.ref_pos, 0));
            lone.present.push_back(mask);
            lone.flags.push_back(eligible ? 1 : 0);
            for (uint32_t a = 0; a < (uint32_t)v.n_alleles(); ++a) {
                const std::string &al = v.allele((int)a);
                lone.pool.insert(lone.pool.end(), al.begin(), al.end());
                lone.allele_off.push_back((uint32_t)lone.pool.size());
            }
            lone.var_allele_off.push_back(lone.var_allele_off.back() + (uint32_t)v.n_alleles());
            lone.cells += v.n_genotypes();
            lone.refs.push_back(&reference);
            lone.blocks.push_back(std::move(vb));
            vb = Block((int)o.k);
            if (lone.n() >= 200000 || lone.cells >= (200u << 20)) index_lone();
            return;
        }
        auto cb = contig_base.find(seq_name);
        waiting_base.push_back(cb == contig_base.end() ? 0 : cb->second);
        waiting_ref.push_back(&reference);
        for (const Variant &v : vb.vars) waiting_cells += v.n_genotypes();
        waiting.push_back(std::move(vb));
        vb = Block((int)o.k);
        if (waiting.size() >= 4096 || waiting_cells >= (200u << 20)) enumerate_waiting(); // bound the panel genotypes held in memory
    }, &dev);
    enumerate_waiting();
    index_lone();
    flush(true);
    if (n_lone_device + n_lone_host)
        std::cerr << "[malva-geno] " << n_lone_device + n_lone_host << " lone variant(s): " << n_lone_device << " indexed on the device, " << n_lone_host
                  << " on the host" << std::endl;
    if (n_general_blocks)
        std::cerr << "[malva-geno] " << n_general_blocks << " general block(s): " << n_general_blocks - n_host_blocks << " enumerated on the device, " << n_host_blocks
                  << " on the host" << std::endl;
    pelapsed("Processed " + std::to_string(n) + " variants");
    dev.check(mg_bf_finalize(dev.ctx, MG_BF_ALT), "mg_bf_finalize(bf)"); // main.cpp:378
    pelapsed("BF creation complete");
    for (const auto &name : used) { // main.cpp:383-401
        auto it = refs.seqs.find(name);
        static const std::string empty;
        const std::string &seq = it == refs.seqs.end() ? empty : it->second;
        auto cb = contig_base.find(name);
        if (cb != contig_base.end()) // the contig is already in HBM (mg_reference_upload above): nothing crosses PCIe again
            dev.check(mg_ref_scan_resident(dev.ctx, cb->second, seq.size()), "mg_ref_scan_resident");
        else
            dev.check(mg_ref_scan(dev.ctx, seq.data(), seq.size()), "mg_ref_scan");
    }
    pelapsed("Reference BF creation complete");
    dev.check(mg_bf_finalize(dev.ctx, MG_BF_CTX), "mg_bf_finalize(context_bf)"); // main.cpp:404
    save_index(dev, o);
    return 0;
}

// ---- call -----------------------------------------------------------------------------------------------------
// k-mer table: text dump, `KMER count` per line -> SoA 2-bit table in pieces
struct TablePiece {
    std::vector<uint64_t> hi, lo;
    std::vector<uint32_t> cnt;
    void clear()
    {
        hi.clear();
        lo.clear();
        cnt.clear();
    }
};

// Rows with a symbol outside ACGT (KMC never lists any) go through the exact ASCII calls at the end.
struct OddRows {
    Rows ctx, kmer;
    std::vector<uint32_t> cnt;
};
// One line of the dump, [p, e) without its terminator: `KMER<ws>count`.  main.cpp:491 upper-cases the k-mer.
inline void parse_table_line(const char *p, const char *e, const Options &o, TablePiece &t, OddRows &odd)
{
    static const struct Lut {
        int8_t code[256];
        Lut()
        {
            for (int i = 0; i < 256; ++i) code[i] = -1;
            code['A'] = code['a'] = 0;
            code['C'] = code['c'] = 1;
            code['G'] = code['g'] = 2;
            code['T'] = code['t'] = 3;
        }
    } lut;
    if (e > p && e[-1] == '\r') --e;
    if (p == e) return;
    const char *q = p;
    uint64_t hi = 0, lo = 0;
    bool acgt = true;
    while (q < e && !isspace((unsigned char)*q)) {
        const int c = lut.code[(unsigned char)*q++];
        acgt = acgt && c >= 0;
        hi = (hi << 2) | (lo >> 62);
        lo = (lo << 2) | (uint64_t)(c & 3);
    }
    const size_t len = (size_t)(q - p);
    if (len != o.ref_k) throw std::runtime_error("k-mer table holds a " + std::to_string(len) + "-mer, expected -r " + std::to_string(o.ref_k));
    while (q < e && isspace((unsigned char)*q)) ++q;
    uint64_t count = 0; // strtoul: leading digits
    while (q < e && *q >= '0' && *q <= '9') count = count * 10 + (uint64_t)(*q++ - '0');
    if (!acgt || o.ref_k > MG_MAX_PACKED_K) { // (a row the 2-bit table cannot hold -- or any row, beyond 64 bases: the ASCII batch forms below)
        std::string ctx(p, len);
        upper_inplace(ctx);
        odd.ctx.add(ctx);
        odd.kmer.add(ctx.substr((o.ref_k - o.k) / 2, o.k));
        odd.cnt.push_back((uint32_t)count);
        return;
    }
    t.hi.push_back(hi);
    t.lo.push_back(lo);
    t.cnt.push_back((uint32_t)count);
}

// The dump of a whole-genome sample is >100 GB of text: a plain (uncompressed) file is mapped and parsed by all
// host cores, 64 MiB of text per task (a line belongs to the task that holds its first byte), and the parsed
// pieces are scanned by THIS thread in whatever order they complete -- the counter updates commute.  A
// compressed dump is inflated and parsed by one thread.
//
// With several GPUs the table shards by rows: every device has a consumer thread that takes the next parsed piece
// and scans it into ITS context's counters; one all-reduce over RCCL then makes every device hold the whole
// table's counters.
void scan_table(std::vector<Device> &devs, const Options &o, const std::string &path)
{
    std::atomic<uint64_t> total{0};
    OddRows odd;
    Device &dev = devs[0];
    auto scan_piece_on = [&](Device &d, TablePiece &t) {
        if (!t.cnt.empty()) d.check(mg_kmc_scan(d.ctx, t.hi.data(), t.lo.data(), t.cnt.data(), t.cnt.size()), "mg_kmc_scan");
        total += t.cnt.size();
        t.clear();
    };
    bool gz = false;
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        unsigned char m[2] = {0, 0};
        gz = fread(m, 1, 2, f) == 2 && m[0] == 0x1f && m[1] == 0x8b;
        fclose(f);
    }
    struct stat st;
    if (!gz && stat(path.c_str(), &st) == 0 && st.st_size > 0) {
        const size_t size = (size_t)st.st_size;
        const int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        const char *base = (const char *)mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        close(fd);
        if (base == MAP_FAILED) throw std::runtime_error("cannot map " + path);
        madvise((void *)base, size, MADV_SEQUENTIAL);
        const char *tb = getenv("MALVA_GENO_TABLE_TASK"); // bytes of text per parsing task (tests shrink it to cross many boundaries)
        const size_t task_bytes = tb && atol(tb) > 0 ? (size_t)atol(tb) : (64u << 20), n_tasks = (size + task_bytes - 1) / task_bytes;
        const unsigned n_threads = (unsigned)std::max<size_t>(1, std::min<size_t>({(size_t)std::thread::hardware_concurrency(), 16, n_tasks}));
        std::atomic<size_t> next_task{0};
        std::mutex mu;
        std::condition_variable cv;
        std::deque<TablePiece> done;   // parsed, waiting for the device
        size_t finished_threads = 0;
        std::string error;
        auto worker = [&]() {
            TablePiece t;
            OddRows mine;
            try {
                for (;;) {
                    const size_t task = next_task.fetch_add(1);
                    if (task >= n_tasks) break;
                    const char *lim = base + std::min(size, (task + 1) * task_bytes), *end = base + size;
                    const char *p = base + task * task_bytes;
                    if (task) { // skip the line that started in the previous task
                        const char *nl = (const char *)memchr(p - 1, '\n', (size_t)(end - (p - 1)));
                        p = nl ? nl + 1 : end;
                    }
                    while (p < lim) {
                        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
                        const char *e = nl ? nl : end;
                        parse_table_line(p, e, o, t, mine);
                        p = e + 1;
                    }
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return done.size() < 2 * n_threads || !error.empty(); }); // bound what is in flight
                    done.emplace_back(std::move(t));
                    t = TablePiece();
                    cv.notify_all();
                }
            } catch (const std::exception &e) {
                std::lock_guard<std::mutex> lk(mu);
                if (error.empty()) error = e.what();
            }
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < mine.cnt.size(); ++i) {
                odd.ctx.add(std::string(&mine.ctx.data[i * STRIDE]));
                odd.kmer.add(std::string(&mine.kmer.data[i * STRIDE]));
                odd.cnt.push_back(mine.cnt[i]);
            }
            ++finished_threads;
            cv.notify_all();
        };
        std::vector<std::thread> pool;
        for (unsigned i = 0; i < n_threads; ++i) pool.emplace_back(worker);
        auto consumer = [&](Device &d) { // one per device; this thread is device 0's
            for (;;) {
                TablePiece t;
                bool failed;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    cv.wait(lk, [&] { return !done.empty() || finished_threads == n_threads; });
                    if (done.empty()) break;
                    t = std::move(done.front());
                    done.pop_front();
                    failed = !error.empty();
                    cv.notify_all();
                }
                try {
                    if (!failed) scan_piece_on(d, t);
                } catch (const std::exception &e) {
                    std::lock_guard<std::mutex> lk(mu);
                    if (error.empty()) error = e.what();
                    cv.notify_all();
                }
            }
        };
        std::vector<std::thread> consumers;
        for (size_t d = 1; d < devs.size(); ++d) consumers.emplace_back([&, d]() { consumer(devs[d]); });
        consumer(devs[0]);
        for (auto &th : consumers) th.join();
        for (auto &th : pool) th.join();
        munmap((void *)base, size);
        if (!error.empty()) throw std::runtime_error(error);
    } else {
        LineReader in(path);
        if (!in.ok()) throw std::runtime_error("cannot open " + path);
        TablePiece t;
        std::string line;
        const size_t piece = 1u << 24;
        size_t turn = 0; // one inflating thread: the pieces still go round the devices, so the exchange is exercised
        while (in.next(line)) {
            parse_table_line(line.data(), line.data() + line.size(), o, t, odd);
            if (t.cnt.size() == piece) scan_piece_on(devs[turn++ % devs.size()], t);
        }
        scan_piece_on(devs[turn % devs.size()], t);
    }
    total += odd.cnt.size();
    if (odd.cnt.size()) { // main.cpp:495-499 through the ASCII batch calls
        std::vector<int32_t> ic(odd.cnt.begin(), odd.cnt.end());
        dev.check(mg_map_increment(dev.ctx, odd.kmer.data.data(), STRIDE, odd.kmer.n, ic.data()), "mg_map_increment");
        std::vector<uint8_t> in_ctx(odd.cnt.size());
        dev.check(mg_bf_test(dev.ctx, MG_BF_CTX, odd.ctx.data.data(), STRIDE, odd.ctx.n, in_ctx.data()), "mg_bf_test");
        Rows pass;
        std::vector<uint32_t> pc;
        for (size_t i = 0; i < odd.cnt.size(); ++i)
            if (!in_ctx[i]) {
                pass.add(std::string(&odd.kmer.data[i * STRIDE]));
                pc.push_back(odd.cnt[i]);
            }
        if (pass.n) dev.check(mg_bf_increment(dev.ctx, MG_BF_ALT, pass.data.data(), STRIDE, pass.n, pc.data()), "mg_bf_increment");
        // device 0's counters join the all-reduce below; exact-map keys the packed table cannot hold live in a host
        // list inside each context, which no collective reaches: the other devices get those increments directly
        Rows irr;
        std::vector<int32_t> irr_c;
        for (size_t i = 0; i < odd.cnt.size(); ++i) {
            const std::string kmer(&odd.kmer.data[i * STRIDE]);
            if (kmer.find_first_not_of("ACGT") != std::string::npos) {
                irr.add(kmer);
                irr_c.push_back((int32_t)odd.cnt[i]);
            }
        }
        for (size_t d = 1; d < devs.size() && irr.n; ++d)
            devs[d].check(mg_map_increment(devs[d].ctx, irr.data.data(), STRIDE, irr.n, irr_c.data()), "mg_map_increment");
    }
    if (devs.size() > 1) { // the exchange step: ncclAllReduce(sum, uint32) of [bf counters | map counters] on every device
        std::vector<mg_ctx *> ctxs;
        for (auto &d : devs) ctxs.push_back(d.ctx);
        dev.check(mg_counters_allreduce_all(ctxs.data(), (int)ctxs.size()), "mg_counters_allreduce_all");
        for (auto &d : devs) d.check(mg_synchronize(d.ctx), "mg_synchronize");
    }
    std::cerr << "[malva-geno] scanned " << total << " k-mers" << (devs.size() > 1 ? " on " + std::to_string(devs.size()) + " devices" : std::string()) << std::endl;
}

// The KMC database itself (main.cpp:444-449, 482-490): the prefix table goes to every device once, then each device
// takes a contiguous range of the records, straight out of the mapped <db>.kmc_suf -- the host decodes nothing; the
// library uploads the raw records in pieces beside the scan of the previous piece and rebuilds the k-mers on the device.
void scan_kmc_db(std::vector<Device> &devs, const Options &o, const std::string &prefix)
{
    KmcDb db;
    db.open(prefix);
    if (db.k != o.ref_k) throw std::runtime_error("KMC database holds " + std::to_string(db.k) + "-mers, expected -r " + std::to_string(o.ref_k));
    std::cerr << "[malva-geno] KMC database: " << db.total << " " << db.k << "-mers, counts " << db.min_count << ".." << db.max_count << ", "
              << db.lut.size() / (1ULL << (2 * db.lut_prefix_len)) << " bin(s) x 4^" << db.lut_prefix_len << " prefixes, " << db.rec_bytes << " B/record" << std::endl;
    on_all_devices(devs, [&](Device &d, size_t i) {
        d.check(mg_kmc_set_lut(d.ctx, db.lut.data(), db.lut.size(), db.lut_prefix_len, db.suffix_bytes, db.counter_size, db.min_count, db.max_count, db.total),
                "mg_kmc_set_lut");
        const uint64_t a = db.total * i / devs.size(), b = db.total * (i + 1) / devs.size();
        if (b > a) d.check(mg_kmc_scan_records(d.ctx, db.records + a * db.rec_bytes, b - a, a), "mg_kmc_scan_records");
    });
    if (devs.size() > 1) {
        std::vector<mg_ctx *> ctxs;
        for (auto &d : devs) ctxs.push_back(d.ctx);
        devs[0].check(mg_counters_allreduce_all(ctxs.data(), (int)ctxs.size()), "mg_counters_allreduce_all");
        for (auto &d : devs) d.check(mg_synchronize(d.ctx), "mg_synchronize");
    }
    std::cerr << "[malva-geno] scanned " << db.total << " k-mers" << (devs.size() > 1 ? " on " + std::to_string(devs.size()) + " devices" : std::string()) << std::endl;
}

// one output record waiting for its device results
struct Rec {
    std::string prefix; // CHROM .. QUAL columns
    uint32_t n_alleles;
    bool isolated;
    size_t slot;     // variant index inside its batch
    size_t allele0;  // first allele slot inside its batch
    size_t gt0;      // first genotype slot inside its batch
};
struct Batch { // inputs of mg_call_isolated (isolated) or mg_lookup_cover + mg_genotype (general)
    std::vector<uint64_t> pos, present;
    std::vector<uint32_t> var_allele_off{0}, allele_off{0};
    std::vector<char> pool;
    std::vector<float> freq;
    std::vector<uint8_t> flags;
    Rows rows;
    std::vector<uint8_t> is_ref;
    std::vector<uint64_t> sig_kmer_off{0}, allele_sig_off{0}, var_gt_off{0};
    // general blocks, enumerated on the device (mg_cover_blocks); the records are kept until the batch has run, for the
    // panel genotypes and for the blocks the device hands back (a Block is rebuilt for those alone: one vector per block
    // kept alive cost a malloc per lone record once every block came this way)
    std::vector<Variant> vars;
    std::vector<int32_t> var_slot;  // record -> its place in `vars`, or -1: a lone record the device cannot hand back, not kept
    std::vector<uint16_t> gt_dense; // a panel of few samples: the genotype words, made as the records are batched (else packed from `vars` by the worker)
    bool dense_gt = false, wide_alleles = false;
    std::vector<const std::string *> block_ref;
    std::vector<uint64_t> blk_base;
    std::vector<uint32_t> blk_len, blk_var_off{0}, ref_size, min_size;
    std::vector<int32_t> ipos;
    std::vector<uint8_t> is_present, canon;
    size_t genotype_cells = 0;
    // results
    std::vector<uint32_t> cov;
    std::vector<int32_t> g1, g2, gq;
    std::vector<uint8_t> status;
    std::vector<double> probs;
    size_t n() const { return var_allele_off.size() - 1; }
};

int call_main(const Options &o)
{
    // Start-up runs three things side by side: the FASTA (host), the index file (host) and the devices (HIP start-up is
    // ~0.4 s by itself).  The sample's table is scanned as soon as the index is on the device; the reference joins after it.
    Reference refs;
    auto fasta_read = std::async(std::launch::async, [&]() {
        Timed t("startup: FASTA");
        return read_fasta(o.fasta_path, o.strip_chr, refs);
    });
    IndexPayload payload;
    auto index_read = std::async(std::launch::async, [&]() { read_index(o, payload); });
    VcfReader vcf(o.vcf_path, o.samples); // (a panel of any size is decoded by a pool of threads: started here, it works through start-up)
    if (vcf.ok()) {
        vcf.want_prefix = true;
        vcf.defer_genotypes = device_gt_wanted(vcf) && !getenv("MALVA_GENO_HOST_CUT"); // (the decode rides on the device cutter's batches)
        vcf.decode_ahead(o.freq_key, o.uniform);
    }
    auto fail_early = [&](const std::string &msg) { // (the readers hold references to this frame: let them finish first)
        fasta_read.wait();
        index_read.wait();
        std::cerr << msg << std::endl;
        return 1;
    };
    // the sample's k-mers: the KMC database the reference opens (main.cpp:444-449), or -- when there is none -- a text dump
    std::string table = o.kmc_path;
    const bool use_db = KmcDb::present(o.kmc_path);
    if (!use_db) {
        if (file_exists(o.kmc_path + ".txt")) table = o.kmc_path + ".txt";
        if (!file_exists(table)) return fail_early("ERROR: cannot open " + o.kmc_path);
    }
    // --gpus N: one context per device -d .. -d+N-1.  MALVA_GENO_SHARE_DEVICE=1 puts all N contexts on device -d: the
    // N-way layout (sharded scan, exchange, split genotyping) rehearsed on a one-GPU box, the exchange then being a
    // kernel instead of RCCL (which rejects two ranks on one device).
    const bool share_device = getenv("MALVA_GENO_SHARE_DEVICE") && atoi(getenv("MALVA_GENO_SHARE_DEVICE")) != 0;
    std::vector<Device> devs((size_t)o.gpus);
    {
        Timed t("startup: devices");
        for (int d = 0; d < o.gpus; ++d)
            if (mg_create(&devs[(size_t)d].ctx, share_device ? o.device : o.device + d, o.k, o.ref_k, o.bf_size) != MG_OK)
                return fail_early("ERROR: no usable MI355X/HIP device " + std::to_string(share_device ? o.device : o.device + d) + "; this build has no CPU path");
    }
    if (o.gpus > 1) {
        std::vector<mg_ctx *> ctxs;
        for (auto &d : devs) ctxs.push_back(d.ctx);
        devs[0].check(mg_comm_init_all(ctxs.data(), o.gpus), "mg_comm_init_all");
        int backend = MG_COMM_NONE;
        mg_comm_info(devs[0].ctx, nullptr, nullptr, &backend);
        std::cerr << "[malva-geno] " << o.gpus << " contexts, exchange: " << (backend == MG_COMM_RCCL ? "RCCL all-reduce" : "one device, kernel sum") << std::endl;
    }
    try {
        index_read.get();
    } catch (...) {
        fasta_read.wait();
        throw;
    }
    import_index(devs, o, payload);
    payload = IndexPayload();
    pelapsed("Reference processed"); // (the phase names are the reference's, main.cpp:452-470; the FASTA itself may still be on its way)
    {
        Timed t("startup: table scan");
        if (use_db) scan_kmc_db(devs, o, o.kmc_path); // main.cpp:482-500
        else scan_table(devs, o, table);
    }
    pelapsed("BF weights created");
    if (!fasta_read.get()) {
        std::cerr << "ERROR: cannot open " << o.fasta_path << std::endl;
        return 1;
    }

    // concatenated reference for the fused isolated path
    std::map<std::string, uint64_t> contig_base;
    {
        std::string all;
        for (const auto &name : refs.names) {
            contig_base[name] = all.size();
            all += refs.seqs.at(name);
        }
        Timed t("startup: reference upload");
        on_all_devices(devs, [&](Device &d, size_t) { d.check(mg_reference_upload(d.ctx, all.data(), all.size()), "mg_reference_upload"); });
    }
    {
        VcfReader hdr(o.vcf_path, "-");
        if (!hdr.ok()) {
            std::cerr << hdr.error << std::endl;
            return 1;
        }
        std::cout << cleaned_header(hdr.header_lines, o.verbose); // main.cpp:505-510
        std::cout.flush();
    }
    if (!vcf.ok()) {
        std::cerr << vcf.error << std::endl;
        return 1;
    }
    pelapsed("VCF parsing and genotyping");

    // records per device round trip; MALVA_GENO_BATCH exists so tests can force many small batches
    const size_t batch_records = getenv("MALVA_GENO_BATCH") ? (size_t)std::max(1L, atol(getenv("MALVA_GENO_BATCH"))) : 200000;
    std::vector<Rec> recs;
    Batch iso, gen;
    // few samples and every record's genotypes decoded on the host: the words are made while batching
    const size_t n_samples_kept = vcf.keep.size();
    const bool dense_gt = n_samples_kept <= SPARSE_GT_SAMPLES && !vcf.defer_genotypes;
    gen.dense_gt = dense_gt;
    std::atomic<size_t> gt_bytes_uploaded{0}; // panel genotypes handed to mg_cover_blocks[_sparse], all batches
    const std::string best_default = o.haploid ? "0" : "0/0";
    auto n_gt = [&](uint64_t A) { return o.haploid ? A : A * (A + 1) / 2; };

    // One batch: device round trips, then the records' text.  Runs on a worker thread while this thread parses the
    // next batch (the ABI has no thread affinity); batches are handed over one at a time, so output order is kept.
    struct Job {
        std::vector<Rec> recs;
        Batch iso, gen;
        size_t device = 0; // batches go round the devices: after the exchange every one of them holds the whole table's counters
    };
    auto process = [&](Job &job) -> std::string {
        std::vector<Rec> &recs = job.recs;
        Batch &iso = job.iso, &gen = job.gen;
        Device &dev = devs[job.device];
        Timed *t_dev = new Timed("worker: device calls");
        std::unique_lock<std::mutex> device_lock(dev.mu); // (the parsing thread cuts blocks on device 0 meanwhile)
        if (iso.n()) {
            const size_t n = iso.n(), na = iso.var_allele_off.back();
            iso.cov.resize(na); iso.g1.resize(n); iso.g2.resize(n); iso.gq.resize(n); iso.status.resize(n);
            iso.probs.resize(o.verbose ? iso.var_gt_off.back() : 0);
            dev.check(mg_call_isolated(dev.ctx, n, iso.pos.data(), iso.var_allele_off.data(), iso.allele_off.data(), iso.pool.data(), iso.pool.size(),
                                       iso.freq.data(), iso.present.data(), iso.flags.data(), o.error_rate, (int)o.max_coverage, o.haploid, iso.cov.data(),
                                       iso.g1.data(), iso.g2.data(), iso.gq.data(), iso.status.data(), o.verbose ? iso.probs.data() : nullptr,
                                       o.verbose ? iso.var_gt_off.data() : nullptr),
                      "mg_call_isolated");
        }
        if (gen.n()) {
            const size_t n = gen.n(), na = gen.var_allele_off.back();
            gen.cov.resize(na); gen.g1.resize(n); gen.g2.resize(n); gen.gq.resize(n); gen.status.resize(n);
            gen.probs.resize(o.verbose ? gen.var_gt_off.back() : 0);
            // panel genotypes of the batch as one [variant][sample] matrix of a1 | a2 << 7 | phased << 14
            const uint32_t n_samples = (uint32_t)vcf.keep.size();
            std::vector<uint8_t> overflow(n, 0);
            bool device_ok = o.k <= MG_MAX_PACKED_K && !getenv("MALVA_GENO_HOST_ENUM"); // the variable forces the host enumerator (tests)
            if (gen.wide_alleles) device_ok = false;
            const size_t n_blocks = gen.blk_var_off.size() - 1;
            PanelGenotypes pg;
            if (gen.dense_gt) pg.gt = std::move(gen.gt_dense);
            else pg = pack_genotypes(gen.vars, n, n_samples, o.haploid); // (every record of such a batch is kept)
            gt_bytes_uploaded += pg.bytes();
            if (device_ok && pg.sparse)
                dev.check(mg_cover_blocks_sparse(dev.ctx, n_blocks, gen.blk_base.data(), gen.blk_len.data(), gen.blk_var_off.data(), n, gen.ipos.data(),
                                                 gen.ref_size.data(), gen.min_size.data(), gen.is_present.data(), gen.var_allele_off.data(),
                                                 gen.allele_off.data(), gen.pool.data(), gen.pool.size(), gen.canon.data(), pg.sp_off.data(), pg.sp_sample.data(),
                                                 pg.sp_gt.data(), pg.sp_default, n_samples, o.haploid, gen.cov.data(), overflow.data()),
                          "mg_cover_blocks_sparse"); // extract_kmers + set_coverages, main.cpp:556-557
            else if (device_ok)
                dev.check(mg_cover_blocks(dev.ctx, n_blocks, gen.blk_base.data(), gen.blk_len.data(), gen.blk_var_off.data(), n, gen.ipos.data(),
                                          gen.ref_size.data(), gen.min_size.data(), gen.is_present.data(), gen.var_allele_off.data(),
                                          gen.allele_off.data(), gen.pool.data(), gen.pool.size(), gen.canon.data(), pg.gt.data(), n_samples, o.haploid,
                                          gen.cov.data(), overflow.data()),
                          "mg_cover_blocks"); // extract_kmers + set_coverages, main.cpp:556-557
            else
                std::fill(overflow.begin(), overflow.end(), 1);
            // blocks the device handed back (a capacity was exceeded, or a window was clipped by a contig end):
            // host enumerator + mg_lookup_cover for exactly those blocks
            size_t n_fallback = 0;
            for (size_t b = 0; b < n_blocks; ++b) {
                bool redo = false;
                for (uint32_t v = gen.blk_var_off[b]; v < gen.blk_var_off[b + 1]; ++v) redo = redo || overflow[v];
                if (!redo) continue;
                ++n_fallback;
                Block blk((int)o.k); // (rebuilt from the batch's records: nothing reads them after this)
                for (uint32_t v = gen.blk_var_off[b]; v < gen.blk_var_off[b + 1]; ++v) {
                    if (gen.var_slot[v] < 0) throw std::runtime_error("internal: the device handed back a record tier 1 should have taken");
                    blk.add(std::move(gen.vars[(size_t)gen.var_slot[v]]));
                }
                genotypes_from_entries(blk);
                const auto sigs = blk.extract(*gen.block_ref[b], o.haploid);
                Rows rows;
                std::vector<uint8_t> is_ref;
                std::vector<uint64_t> sig_off{0}, al_off{0};
                for (size_t vi = 0; vi < blk.vars.size(); ++vi)
                    for (int a = 0; a < blk.vars[vi].n_alleles(); ++a) {
                        auto it = sigs[vi].find(a);
                        if (it != sigs[vi].end())
                            for (const auto &sig : it->second) {
                                for (const auto &kmer : sig) {
                                    rows.add(kmer);
                                    is_ref.push_back(a == 0);
                                }
                                sig_off.push_back(rows.n);
                            }
                        al_off.push_back(sig_off.size() - 1);
                    }
                const size_t slot0 = gen.var_allele_off[gen.blk_var_off[b]];
                dev.check(mg_lookup_cover(dev.ctx, rows.data.data(), STRIDE, rows.n, is_ref.data(), sig_off.data(), sig_off.size() - 1, al_off.data(),
                                          al_off.size() - 1, gen.cov.data() + slot0),
                          "mg_lookup_cover");
            }
            if (n_fallback) std::cerr << "[malva-geno] " << n_fallback << " block(s) enumerated on the host" << std::endl;
            dev.check(mg_genotype(dev.ctx, gen.cov.data(), gen.freq.data(), gen.var_allele_off.data(), n, o.error_rate, (int)o.max_coverage, o.haploid,
                                  gen.g1.data(), gen.g2.data(), gen.gq.data(), gen.status.data(), o.verbose ? gen.probs.data() : nullptr,
                                  o.verbose ? gen.var_gt_off.data() : nullptr),
                      "mg_genotype"); // vb.genotype + the GT/GQ part of output_variants, main.cpp:558-559
        }
        device_lock.unlock(); // the records' text needs no device
        delete t_dev;
        Timed t_text("worker: records' text");
        std::string out;
        char num[64];
        for (const Rec &r : recs) { // output_variants, var_block.hpp:337-396
            const Batch &b = r.isolated ? iso : gen;
            out += r.prefix;
            out += "\tPASS\t";
            const uint32_t *cov = &b.cov[r.allele0];
            const uint8_t st = b.status[r.slot];
            auto gname = [&](int a, int c) { return o.haploid ? std::to_string(a) : std::to_string(a) + "/" + std::to_string(c); };
            if (o.verbose) {
                out += "COVS=";
                for (uint32_t a = 0; a < r.n_alleles; ++a) {
                    out += std::to_string((int)cov[a]);
                    out += a + 1 < r.n_alleles ? "," : "";
                }
                out += ";GTS=";
                if (st == MG_GT_NORMAL) {
                    size_t q = r.gt0;
                    bool first = true;
                    for (uint32_t a = 0; a < r.n_alleles; ++a)
                        for (uint32_t c = a; c < (o.haploid ? a + 1 : r.n_alleles); ++c, ++q) {
                            if (std::isnan(b.probs[q])) snprintf(num, sizeof num, "-nan"); // 0.0/0.0 on x86, as the reference prints it
                            else snprintf(num, sizeof num, "%f", b.probs[q]);               // std::to_string(double)
                            out += (first ? "" : ",") + gname((int)a, (int)c) + ":" + num;
                            first = false;
                        }
                } else {
                    // early-outs: one (best_geno, 0) per over-covered allele, or a single entry; 0/0 prints -nan
                    size_t entries = 1;
                    if (st == MG_GT_OVERCOV) {
                        entries = 0;
                        for (uint32_t a = 0; a < r.n_alleles; ++a) entries += (int)cov[a] > (int)o.max_coverage;
                    }
                    for (size_t e = 0; e < entries; ++e) out += (e ? "," : "") + best_default + (st == MG_GT_SINGLE ? ":1.000000" : ":-nan");
                }
            } else
                out += ".";
            out += "\tGT:GQ\t";
            out += gname(b.g1[r.slot], b.g2[r.slot]); // early-outs and "nothing beats 0.0" come back as 0 / 0/0
            out += ":" + std::to_string(b.gq[r.slot]) + "\n";
        }
        return out;
    };
    // as many batches in flight as there are devices; their text leaves in submission order
    std::deque<std::future<std::string>> in_flight;
    size_t jobs_started = 0;
    auto drain = [&](size_t keep) {
        Timed t_drain("main: wait for worker + write");
        while (in_flight.size() > keep) {
            const std::string text = in_flight.front().get(); // (or the batch's exception comes back here)
            in_flight.pop_front();
            if (fwrite(text.data(), 1, text.size(), stdout) != text.size()) throw std::runtime_error("cannot write the output");
        }
    };
    auto reserve_general = [&](Batch &b) { // (a batch's vectors at their final size at once: fifteen of them grew by doubling, record by record)
        const size_t n = batch_records + 64;
        for (auto *v32 : {&b.blk_len, &b.blk_var_off, &b.ref_size, &b.min_size, &b.var_allele_off}) v32->reserve(n + 1);
        b.allele_off.reserve(2 * n + 1);
        b.blk_base.reserve(n); b.ipos.reserve(n); b.is_present.reserve(n); b.canon.reserve(2 * n); b.freq.reserve(2 * n); b.pool.reserve(4 * n);
        b.var_gt_off.reserve(n + 1); b.var_slot.reserve(n); b.block_ref.reserve(n);
        if (b.dense_gt) b.gt_dense.reserve(n * n_samples_kept);
    };
    reserve_general(gen);
    auto run_and_print = [&]() {
        drain(devs.size() - 1);
        auto job = std::make_shared<Job>();
        job->recs = std::move(recs);
        job->iso = std::move(iso);
        job->gen = std::move(gen);
        job->device = jobs_started++ % devs.size();
        recs.clear();
        iso = Batch();
        gen = Batch();
        gen.dense_gt = dense_gt;
        reserve_general(gen);
        recs.reserve(batch_records + 64);
        in_flight.push_back(std::async(std::launch::async, [&process, job]() { return process(*job); }));
    };

    auto prefix_of = [&](Variant &v) { return std::move(v.text_prefix); }; // (made by the thread that decoded the record)

    const bool iso_path = getenv("MALVA_GENO_ISO_PATH") && atoi(getenv("MALVA_GENO_ISO_PATH")) != 0;
    std::string base_name;
    bool base_known = false, base_found = false;
    uint64_t base_value = 0;
    const size_t n = for_each_block(vcf, o, refs, false, nullptr, [&](Block &vb, const std::string &seq_name, const std::string &reference) {
        // The fused kernel reads ceil(k/2) bases to the right of the REF allele.  A lone variant whose right flank
        // would cross the contig end (a deletion longer than about k/2 within k of it) gets a CLIPPED, shorter
        // k-mer in the reference (std::string(reference, pos, n), var_block.hpp:187) and in extract_lone at index
        // time: such a variant takes the general path, whose device enumerator hands clipped windows to the host.
        if (!base_known || seq_name != base_name) { // (as ref_of: one map lookup per sequence, not per record)
            auto cb = contig_base.find(seq_name);
            base_found = cb != contig_base.end();
            base_value = base_found ? cb->second : 0;
            base_name = seq_name;
            base_known = true;
        }
        // Every block takes the resident record loop (mg_cover_blocks: the panel batch goes up once, then tier 1 for the lone
        // records -- classification and the panel's genotypes gathered on the device --, tiers 2 and 3 for the others: the
        // kernels bench.py times).  MALVA_GENO_ISO_PATH=1 keeps the older split, where a lone record goes through the fused
        // mg_call_isolated with a presence mask made here (tests run both: same bytes).
        // (k beyond the packed forms: exact-map keys live in the context's host-side list, which only the ASCII batch lookups reach)
        const bool lone = iso_path && o.k <= MG_MAX_PACKED_K && vb.is_lone_short() && base_found &&
                          (long)vb.vars[0].ref_pos + vb.vars[0].ref_size + (long)(o.k + 1) / 2 <= (long)reference.size();
        if (lone) {
            Variant &v = vb.vars[0];
            const uint32_t A = (uint32_t)v.n_alleles();
            recs.push_back({prefix_of(v), A, true, iso.n(), iso.var_allele_off.back(), iso.var_gt_off.back()});
            const bool eligible = v.is_present && v.ref_pos >= (int)o.k && v.ref_pos <= (int)reference.size() - (int)o.k; // var_block.hpp:104
            const uint64_t mask = eligible ? carried_mask(v, o.haploid) : 0;
            iso.pos.push_back(base_value + (uint64_t)std::max(v.ref_pos, 0));
            iso.present.push_back(mask);
            iso.flags.push_back(eligible ? 1 : 0);
            for (uint32_t a = 0; a < A; ++a) {
                const std::string &al = v.allele((int)a);
                iso.pool.insert(iso.pool.end(), al.begin(), al.end());
                iso.allele_off.push_back((uint32_t)iso.pool.size());
                iso.freq.push_back(v.frequencies[a]);
            }
            iso.var_allele_off.push_back(iso.var_allele_off.back() + A);
            iso.var_gt_off.push_back(iso.var_gt_off.back() + n_gt(A));
        } else {
            // main.cpp:556-557: extract_kmers + set_coverages happen on the device for the whole batch of blocks
            gen.blk_base.push_back(base_value);
            gen.blk_len.push_back((uint32_t)reference.size());
            // a record tier 1 is sure to take (the device's own test, block_pipeline.h: classify_lone) can never be handed back:
            // with its genotype words made here, nothing of it needs keeping
            const bool sure_lone = gen.dense_gt && o.k <= MG_MAX_PACKED_K && vb.is_lone_short() && base_found && vb.vars[0].ref_pos >= (int)o.k / 2 &&
                                   (long)vb.vars[0].ref_pos + vb.vars[0].ref_size + (long)(o.k + 1) / 2 <= (long)reference.size();
            for (Variant &v : vb.vars) {
                const uint32_t A = (uint32_t)v.n_alleles();
                if (A > 127) gen.wide_alleles = true;
                recs.push_back({prefix_of(v), A, false, gen.n(), gen.var_allele_off.back(), gen.var_gt_off.back()});
                gen.ipos.push_back(v.ref_pos);
                gen.ref_size.push_back((uint32_t)v.ref_size);
                gen.min_size.push_back((uint32_t)v.min_size);
                gen.is_present.push_back(v.is_present);
                for (uint32_t a = 0; a < A; ++a) {
                    const std::string &al = v.allele((int)a);
                    gen.pool.insert(gen.pool.end(), al.begin(), al.end());
                    gen.allele_off.push_back((uint32_t)gen.pool.size());
                    gen.canon.push_back((uint8_t)std::min(255, v.allele_index(al)));
                    gen.freq.push_back(v.frequencies[a]);
                }
                gen.var_allele_off.push_back(gen.var_allele_off.back() + A);
                gen.var_gt_off.push_back(gen.var_gt_off.back() + n_gt(A));
                gen.genotype_cells += v.n_genotypes();
                if (gen.dense_gt) { // one row of n_samples words per record (a record whose genotypes were not read -- it carries nothing -- keeps a row of zeros)
                    const size_t have = std::min(v.genotypes.size(), n_samples_kept);
                    for (size_t s_ = 0; s_ < have; ++s_) gen.gt_dense.push_back(genotype_word(v, s_, o.haploid));
                    gen.gt_dense.resize(gen.gt_dense.size() + (n_samples_kept - have), 0);
                }
                if (sure_lone) gen.var_slot.push_back(-1);
                else {
                    gen.var_slot.push_back((int32_t)gen.vars.size());
                    gen.vars.push_back(std::move(v)); // (the caller clears the block: its vector is used again)
                }
            }
            gen.blk_var_off.push_back((uint32_t)gen.n());
            gen.block_ref.push_back(&reference);
        }
        if (gen.genotype_cells >= (200u << 20)) run_and_print(); // bound the panel genotypes held in memory
        if (recs.size() >= batch_records) run_and_print();
    }, &devs[0]);
    run_and_print();
    drain(0);
    fflush(stdout);
    if (gt_bytes_uploaded) std::cerr << "[malva-geno] panel genotypes of the general blocks: " << gt_bytes_uploaded.load() << " bytes uploaded" << std::endl;
    pelapsed("Processed " + std::to_string(n) + " variants");
    pelapsed("Execution completed");
    return 0;
}

// index-convert: rewrite the index of <variants.vcf> in the other container (third argument: zst | hipz); host only
int convert_main(const Options &o)
{
    IndexPayload p;
    const bool to_zst = o.kmc_path == "zst";
    if (!to_zst && o.kmc_path != "hipz") throw std::runtime_error("index-convert: the third argument names the target container, zst or hipz");
    if (to_zst) {
        load_index_hipz(index_path(o, ".hipz"), p, o.k, o.ref_k, o.bf_size, STRIDE);
        save_index_zst(index_path(o, ".zst"), p, o.bf_size);
    } else {
        load_index_zst(index_path(o, ".zst"), p, o.bf_size, STRIDE);
        save_index_hipz(index_path(o, ".hipz"), p, o.k, o.ref_k, o.bf_size);
    }
    std::cerr << "[malva-geno] index rewritten as " << index_path(o, to_zst ? ".zst" : ".hipz") << ": " << p.filt[0].pos.size() << " + " << p.filt[1].pos.size()
              << " filter bits, " << p.vals.size() << " keys" << std::endl;
    return 0;
}

// dump-kmers: host-only view of the enumerator (tests compare it with the oracle's block model)
int dump_main(const Options &o)
{
    Reference refs;
    if (!read_fasta(o.fasta_path, o.strip_chr, refs)) {
        std::cerr << "ERROR: cannot open " << o.fasta_path << std::endl;
        return 1;
    }
    VcfReader vcf(o.vcf_path, o.samples);
    if (!vcf.ok()) {
        std::cerr << vcf.error << std::endl;
        return 1;
    }
    const bool for_index = o.kmc_path == "index";
    for_each_block(vcf, o, refs, for_index, nullptr, [&](Block &vb, const std::string &, const std::string &reference) {
        const auto sigs = vb.signatures(reference, o.haploid);
        std::cout << "BLOCK " << vb.vars.size() << (vb.is_lone_short() ? " lone" : "") << "\n";
        for (size_t vi = 0; vi < vb.vars.size(); ++vi) {
            const Variant &v = vb.vars[vi];
            std::cout << "VAR " << v.seq_name << " " << v.ref_pos + 1 << " " << v.ref_sub;
            for (const auto &a : v.alts) std::cout << " " << a;
            std::cout << " present=" << v.is_present << "\n";
            for (const auto &as : sigs[vi]) {
                std::vector<std::string> lines;
                for (const auto &sig : as.second) {
                    std::string l;
                    for (const auto &kmer : sig) l += (l.empty() ? "" : ",") + kmer;
                    lines.push_back(l);
                }
                std::sort(lines.begin(), lines.end()); // signature order is unordered_set order in the reference
                for (const auto &l : lines) std::cout << "SIG " << as.first << " " << l << "\n";
            }
        }
    });
    return 0;
}

} // namespace

int main(int argc, char **argv)
{
    if (argc < 2) {
        std::cerr << "malva missing arguments\n" << USAGE << std::endl;
        return 1;
    }
    Options o;
    const std::string cmd = argv[1];
    try {
        if (cmd == "index-convert") {
            if (!parse_arguments(argc - 1, argv + 1, o)) return EXIT_FAILURE;
            return convert_main(o);
        }
        if (cmd.compare(0, 5, "index") == 0) {
            if (!parse_arguments(argc - 1, argv + 1, o)) return EXIT_FAILURE;
            return index_main(o);
        }
        if (cmd.compare(0, 4, "call") == 0) {
            if (!parse_arguments(argc - 1, argv + 1, o)) return EXIT_FAILURE;
            return call_main(o);
        }
        if (cmd == "dump-kmers") {
            if (!parse_arguments(argc - 1, argv + 1, o)) return EXIT_FAILURE;
            return dump_main(o);
        }
    } catch (const std::exception &e) {
        std::cerr << "ERROR: " << e.what() << std::endl;
        return 1;
    }
    std::cerr << "Could not interpret command " << argv[1] << ".\nAccepted commands are index and call." << std::endl;
    return 1;
}
