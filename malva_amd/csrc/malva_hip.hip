// malva_hip.hip -- kernels and C ABI of the MI355X-native malva-geno hot path.
// See include/malva_hip.h for the interface and the reference lines each entry
// point replaces; DESIGN.md for the data layout and the roofline of each kernel.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types and prototypes only: the library itself is opened on first use (mg_comm_*)

#include <dlfcn.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "geno_dev.h"
#include "malva_hip.h"

using namespace mg;

#define MG_EXPORT extern "C" __attribute__((visibility("default")))

// ===========================================================================
// Kernels
// ===========================================================================

namespace {

constexpr size_t TK_META_HEAD = 8; // ticket meta block: spill count, then the segment fills
constexpr int BIN_SEGS = 2048; // workgroups of the binning kernel = segments per bin
constexpr int TPB = 256;

#include "store_kernels.h"
#include "scan_kernels.h"
#include "variant_kernels.h"
#include "block_pipeline.h"
#include "gt_text_kernels.h"

} // namespace

// ===========================================================================
// Host side: context, memory, launches
// ===========================================================================

struct BFState {
    u64 size = 0, nwords = 0, n_blk = 0, nset = 0;
    u64 *words = nullptr;
    u32 *blk = nullptr;
    u32 *counts = nullptr;
    u64 *gate = nullptr; // only `bf` (MG_BF_ALT) owns one
    u64 n_gate_bits = 0;
    u32 gate_shift = 6;
    u64 *pos_set = nullptr; // `context_bf`, call time: its set positions as a hash set (BFView::pos_set); built at the first scan
    u32 pos_set_log2 = 0;
    bool pos_set_valid = false;
    u64 *pregate = nullptr; // coarse L2-sized gate in front of a gate that outgrew L2
    u32 pre_shift = 6;
    int mode = 0;
    ModDesc mod{};
};
struct MapState {
    u32 cap_log2 = 0;
    MapSlot *slots = nullptr; // `tags` in the comments below = the tag field of these records
    u32 *vals = nullptr;
    u64 rows_total = 0; // insertion rows so far (upper bound on distinct keys; ids index space)
    u64 vals_cap = 0;
    std::unordered_map<std::string, int32_t> irregular; // keys the packed table cannot hold (N / NUL-truncated)
};
struct Scratch {
    void *p = nullptr;
    size_t cap = 0;
};

struct mg_ctx {
    int device = 0;
    hipStream_t stream = nullptr, own_stream = nullptr;
    u32 k = 0, ref_k = 0;
    BFState bf[2];
    MapState map;
    Scratch s_rows, s_aux, s_out, s_irr, s_open[3], s_hit[4], s_misc[8], s_blk[16], s_gt[10], s_scan;
    void *h_gt_stage = nullptr;                       // pinned staging for mg_decode_gt_text's text (a pageable source is copied by the runtime in small synchronous pieces)
    size_t h_gt_stage_cap = 0;
    u32 gt_records = 0, gt_keep = 0, gt_default = 0; // the batch mg_decode_gt_text left on the device for mg_decode_gt_entries
    u64 gt_entries = 0;
    unsigned long long *d_gen_count = nullptr; // [0] records listed for cover_blocks_kernel, [1] insertion-row cursor / block count of a host-form batch,
                                               // [2] signature k-mers of the lone records, [3] of the others (mg_blocks_stats)
    hipEvent_t ev_b[4] = {nullptr, nullptr, nullptr, nullptr}; // mg_cover_blocks_device: before / after the preparation, the lone kernels, the enumerating kernel
    bool blocks_stats_valid = false;
    int blocks_grid[3] = {0, 0, 0};            // persistent grid of cover_blocks_kernel<MODE> (found at first use)
    int use_flat_tier = 1;                     // 0: every general record takes the workgroup kernel (A/B, tests)
    // the records' own copies of their counters (MapSlot::cval / cbf): current for epoch rec_epoch while rec_ok.  Every ABI call
    // drops rec_ok on entry unless it is one of those that cannot change a counter behind the copies' back (DeviceGuard, KEEP).
    int use_record_counters = 1;
    mutable bool rec_ok = false;
    // lazy vectors: scans of the sub-slice form add to the records' copies alone while those are current (MapView::lazy); whoever
    // needs vals[] / counts[] afterwards -- an export, a per-k-mer call, an exchange, anything that ends the copies' validity --
    // first brings them up to date (vectors_current: one pass over the record table).  vec_zero: the vectors are known to be all
    // zero (mg_counters_reset then has nothing to clear: 0.7 GB per step at whole-genome scale)
    int lazy_vectors = 1;
    mutable bool vec_stale = false, vec_zero = false;
    bool rec_off = false;                      // the counter vector has been handed out (mg_counters_view): whoever holds it may write it
    u32 rec_epoch = 1;
    int blocks_round_log2 = 24;                // see blocks_setup
    int use_hit_entries = 1;                   // scan: the probe kernel hands the hit kernel each row's filter entry (counter index, record) with the row
    int use_snp_chains = 1;                    // record loop: chains of SNPs assembled as the reference window with the members' bases put in
    u64 last_rounds = 0;                       // rounds of tier 2 in the most recent record loop (blocks_listed_chains)
    int use_chain_kernel = 1;                  // record loop: the picks of a chain evaluated by the wave that holds them (fw_chain_kernel) instead of picks -> items -> eval
    int use_snp_kernel = 1;                    // record loop: chains of SNPs on panels of up to 8 diploid / 16 haploid samples in one kernel (fw_snp_kernel) instead of picks + eval
    int use_chain_order = 1;                   // record loop: what fw_snp_kernel left of a round's chains, listed in order of their number of members, before the
                                               // chain kernel takes them (fw_order_kernel: less divergence in the per-member loops, and no walk over descriptors
                                               // already dealt with)
    int use_packed_pool = 1;                   // record loop: signature k-mers assembled from 2-bit alleles (mg_panel_dev.pool_bytes) instead of bytes
    int map_ordered = 1;                       // records in order of the filter slot (map_home); fixed before the first key or filter entry goes in
    int map_dense = 0;                         // 1: record tables beyond 4 GB are sized at load 1/2 instead of 1/4 (measured at C4: the probe kernel got 25 % SLOWER -- longer walks, same translation cost)
    int use_ctx_set = 1;                       // 1: large context filters answer the scan's hit kernel from the set of their set positions (A/B)
    int use_packed_ref_scan = 1;               // 0: the byte-wise reference scan for every window (A/B, tests)
    unsigned long long *d_hit_count = nullptr;
    double *d_ln = nullptr;
    float *d_eps = nullptr; // [2 * MG_EPS_TABLE]
    float eps_for = -1.f;
    u8 *d_ref = nullptr;
    size_t ref_len = 0;
    u64 *d_ref2 = nullptr;   // the same reference as 2-bit codes (ref_pack_kernel) ...
    u32 *d_refbad = nullptr; // ... and one bit per base: not ACGT
    // scan timing (mg_scan_stats): four events per launch group (chunk) of the most recent scan, the first SCAN_EV_CHUNKS of them
    std::vector<hipEvent_t> ev;
    std::vector<u64> ev_rows; // rows of each timed chunk
    bool stats_valid = false;
    u32 *joined = nullptr; // when set: one allocation holding [bf counters | map counters] (mg_counters_view)
    int use_summary = 1;
    bool gate_dirty = false; // something has been inserted into `bf`
    bool gate_fixed = false; // gate_log2 was set by the caller: do not resize at finalize
    int scan_rows = 2;    // table rows per thread per iteration of the filter kernel (swept: 2 is best)
    int scan_grid = 8192; // workgroups of the filter kernel (32 per CU; swept 2048..8192)
    int scan_ablate = 0;  // timing-only diagnostic, see scan_filter_kernel
    u32 iso_call_no = 0;  // mg_call_isolated calls so far (see iso_cover_kernel)
    u32 blocks_set_limit = BK_SET_CAP / 2; // cover_blocks_kernel: distinct picks per chain kept in LDS before it evaluates every pick
    int scan_variant = 2; // filter-kernel VAR bits (staging / load width): 16-byte loads measured best
    int pre_k = 1;      // bits per entry of the coarse gate (chosen at finalize from the load)
    int use_pregate = 1;
    bool pre_skip = false; // the coarse gate is saturated at this index size (decided at finalize): scans go straight to the fine gate
    int pregate_log2 = 25; // coarse gate size: 4 MiB, what stays resident in an XCD's L2 next to the table stream
    int use_partition = 0; // bin the coarse gate's survivors by fine-gate slice (SoA tables, fine gates of 4..64 MiB).  Off: since the ticket and sub-slice forms took the
                           // gates of 32 MiB and more, what is left to this form is where the plain filter kernel beats it (C5, 16 MiB gate: 0.84 against 1.03 ms per
                           // 1e8 rows; C3-shaped indexes of 4e6 and 8e6 entries: 0.82 / 1.09 against 0.96 / 1.23)
    int probe_grid = 2048, hits_grid = 1024; // workgroups of the two list kernels (swept)
    int use_tickets = 1;      // gates of 2^ticket_min_log2 bits and more: 8-byte tickets filed by 2 MiB gate slice, the slices then walked out of L2
                              // (scan_ticket_sort_kernel + scan_ticket_gate_kernel) instead of one random HBM sector per table row
    int tkg_grid = 0;         // pass two's grid: one workgroup per CU (found at first use)
    int chunk_log2 = 27;      // rows per launch group (tests shrink it to put many chunks into a small table)
    int ticket_min_log2 = 28; // smallest fine gate (log2 bits) that takes the ticket form: 32 MiB.  Measured on a C4 share (3.75e8 rows, compact rows):
                              // 16 MiB gate 3.6 ms two-level direct / 4.8 tickets; 32 MiB 7.7 / 6.2; 256 MiB 9.3 / 7.3 (profiles/r02_c4share_forms.txt)
    Scratch s_tk[2];
    unsigned long long *d_tk_meta = nullptr; // spill count, then u32 [TK_MAXP][BIN_SEGS] segment fills
    // the sub-slice form (scan_sub_sort_kernel + scan_sub_gate_kernel): tickets filed under LDS-sized pieces of the gate
    int use_sub = 1;          // gates of 2^sub_min_log2 bits and more (takes precedence over the ticket form; 0: A/B)
    int sub_min_log2 = 28;    // smallest fine gate (log2 bits) that takes it
    int sub_words_log2 = SB_WORDS_LOG2; // gate words per sub-slice (tests shrink it to put many bins into a small gate)
    int sub_bins_log2 = 10;   // the gate is sized (at finalize) to split into at most 2^sub_bins_log2 sub-slices (<= SB_MAXB).  Measured at one GPU's share of
                              // C4: 2,048 pieces (a 256 MiB gate, ~0.5 % false positives) leave pass one in runs of 8 tickets = 64 B and cost its stores and pass
                              // two 0.5 ms per 2^27 rows more than 1,024 pieces (128 MiB, ~4 %) cost the probe kernel (0.23 ms)
    int sub_split = 4;        // workgroups of the probe kernel per region of open rows
    int sub_grid = 0;         // pass one's grid (0: one workgroup per CU)
    int n_cus = 0;            // CUs of the device (found at first use)
    Scratch s_sb[4];          // tickets, spill list, open tickets, meta block (spill count, region counts, segment fills)
    int last_subs = 0;        // bins the most recent scan filed tickets under (0: it did not take this form)
    int bin_ring = 0, bin_rows = 4; // A/B: staging ring per bin (0 = as large as LDS allows), rows per thread of the binning kernel
    u64 bin_cap = 0;       // rows per bin segment; 0 = 1.5x an even share of the chunk (tests set it small to reach the spill path)
    int last_bins = 0;     // bins used by the most recent scan (0: direct form)
    int last_tickets = 0;  // gate slices the most recent scan filed tickets under (0: it did not)
    Scratch s_bin[3], s_spill[3];
    unsigned long long *d_bin_meta = nullptr; // spill count, then u32 [BIN_MAXP][BIN_SEGS] segment fills
    int gate_k = 4;     // gate bits per entry (blocked Bloom filter inside one 64-bit word; swept 2..4)
    int gate_log2 = 25; // gate of at most 2^gate_log2 bits = 4 MiB (swept 24..26: 25 gives the best whole-scan time)
    // multi-GPU exchange (mg_comm_*): an RCCL communicator, or -- contexts of one process that share a device, where
    // RCCL refuses duplicate ranks -- the list of contexts whose counters a kernel sums
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 0;
    std::vector<mg_ctx *> local_group;
    hipEvent_t ev_x = nullptr; // orders the local-group exchange between the contexts' streams
    // the exchange's own stream (mg_counters_allreduce_begin / _end: what needs no counters runs on the context's stream meanwhile),
    // its timing events, and the 16-bit packed form: two counters per word on the wire when no rank's partial counter can carry
    hipStream_t xstream = nullptr;
    hipEvent_t ev_xs[4] = {nullptr, nullptr, nullptr, nullptr}; // scan done / exchange done (ordering), exchange start / end (timing)
    bool x_pending = false, x_timed = false;
    int exchange_pack = 1;          // 0 never, 1 vectors of exchange_pack_min_mb and more, 2 always (when exact)
    int exchange_pack_min_mb = 32;
    int last_exchange_packed = 0;
    u32 *d_xmax = nullptr;
    Scratch s_pack;
    // host-fed scans (mg_kmc_scan, mg_kmc_scan_records): two staging slots, uploads on their own stream beside the scan
    hipStream_t copy_stream = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    Scratch s_stage[2][3], s_raw[2];
    u64 *d_kmc_lut = nullptr; // <db>.kmc_pre's prefix table (+ guard), mg_kmc_set_lut
    u64 kmc_n_lut = 0;
    u32 kmc_prefix_len = 0, kmc_suffix_bytes = 0, kmc_counter_bytes = 0, kmc_min_count = 0;
    u64 kmc_max_count = 0;
    std::string err;
};

namespace {

int fail(mg_ctx *c, int code, const char *fmt, ...)
{
    if (c) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof buf, fmt, ap);
        va_end(ap);
        c->err = buf;
    }
    return code;
}
#define HIP_TRY(c, expr)                                                                                             \
    do {                                                                                                             \
        hipError_t e_ = (expr);                                                                                      \
        if (e_ != hipSuccess)                                                                                        \
            return fail(c, e_ == hipErrorOutOfMemory ? MG_ERR_NOMEM : MG_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define TRY(expr)                 \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != MG_OK) return rc_; \
    } while (0)

inline unsigned nblocks(u64 n) { return (unsigned)((n + TPB - 1) / TPB); }
int comm_drop(mg_ctx *c); // multi-GPU section

int scratch(mg_ctx *c, Scratch &s, size_t bytes, void **out)
{
    if (bytes > s.cap) {
        if (s.p) HIP_TRY(c, hipFree(s.p));
        s.p = nullptr;
        s.cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        HIP_TRY(c, hipMalloc(&s.p, want));
        s.cap = want;
    }
    *out = s.p;
    return MG_OK;
}
// exclusive scan of n_tiles u32 sums in place, their total (u64) to *d_total: store_kernels.h
int launch_tile_scan(mg_ctx *c, u32 *d_tiles, u64 n_tiles, unsigned long long *d_total)
{
    const u64 n_part = (n_tiles + SCAN_CHUNK - 1) / SCAN_CHUNK;
    void *part;
    TRY(scratch(c, c->s_scan, 8 * (n_part ? n_part : 1), &part));
    if (n_part) hipLaunchKernelGGL(tile_reduce_kernel, dim3((unsigned)n_part), dim3(SCAN_TPB), 0, c->stream, (const u32 *)d_tiles, n_tiles, (unsigned long long *)part);
    hipLaunchKernelGGL(part_scan_kernel, dim3(1), dim3(SCAN_TPB), 0, c->stream, (unsigned long long *)part, n_part, d_total);
    if (n_part) hipLaunchKernelGGL(tile_rescan_kernel, dim3((unsigned)n_part), dim3(SCAN_TPB), 0, c->stream, d_tiles, n_tiles, (const unsigned long long *)part);
    HIP_TRY(c, hipGetLastError());
    return MG_OK;
}
int upload(mg_ctx *c, Scratch &s, const void *host, size_t bytes, void **dev)
{
    TRY(scratch(c, s, bytes ? bytes : 1, dev));
    if (bytes) HIP_TRY(c, hipMemcpyAsync(*dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    return MG_OK;
}

ModDesc make_mod(u64 size)
{
    ModDesc m{};
    m.size = size;
    u32 sh = 0;
    u64 odd = size;
    while (odd && !(odd & 1)) {
        odd >>= 1;
        ++sh;
    }
    m.shift = sh;
    m.odd = odd;
    if (odd == 1) m.kind = 0;
    else if (odd < (1ULL << 32) && sh >= 32) m.kind = 1;
    else m.kind = 2;
    return m;
}

// use_pregate: 0 never, 1 unless saturated (pre_skip), 2 always (tests)
bool pregate_on(const mg_ctx *c) { return c->use_pregate == 2 || (c->use_pregate == 1 && !c->pre_skip); }

BFView view(const mg_ctx *c, int which)
{
    const BFState &b = c->bf[which];
    BFView v{};
    v.words = b.words;
    v.blk = b.blk;
    v.counts = b.counts;
    v.gate = b.gate;
    v.mod = b.mod;
    v.gate_shift = b.gate_shift;
    v.gate_k = (u32)c->gate_k;
    v.pregate = pregate_on(c) ? b.pregate : nullptr;
    v.pregate_fill = b.pregate; // filled whenever it exists, so use_pregate / pre_skip may change between scans
    v.pre_shift = b.pre_shift;
    v.pre_k = (u32)c->pre_k;
    v.use_gate = (c->use_summary && b.gate) ? 1 : 0;
    v.pos_set = b.pos_set_valid ? b.pos_set : nullptr;
    v.pos_set_log2 = b.pos_set_log2;
    return v;
}
// Are the records' counter copies worth keeping?  They cost the scan a compare-and-swap per hit beside its add to the vector
// and save a lookup its second random line.  While vals[] / counts[] stay in the caches (C3: 8 MB) that line is cheap and
// the copies lose (measured: 1.043 against 1.000 ms per C3 step); beyond that they win (use_record_counters: 0 never, 2 always).
bool records_wanted(const mg_ctx *c)
{
    if (!c->use_record_counters || c->rec_off || c->comm_world > 1 || !c->local_group.empty()) return false;
    return c->use_record_counters >= 2 || (u64)(c->bf[MG_BF_ALT].nset + c->map.rows_total) * 4 > (256ull << 20);
}
MapView view(const mg_ctx *c)
{
    const MapState &m = c->map;
    MapView v{};
    v.slots = m.slots;
    v.vals = m.vals;
    v.cap_log2 = m.cap_log2;
    v.klen = c->k;
    v.home_mul = c->map_ordered ? ~0ULL / c->bf[MG_BF_ALT].mod.size : 0x9E3779B97F4A7C15ULL;
    v.epoch = c->rec_ok && records_wanted(c) ? c->rec_epoch : 0;
    v.lazy = v.epoch && c->lazy_vectors ? 1u : 0u;
    return v;
}

// give the two counter arrays their own allocations again (before either has to be resized)
int unjoin(mg_ctx *c)
{
    if (!c->joined) return MG_OK;
    BFState &b = c->bf[MG_BF_ALT];
    MapState &m = c->map;
    u32 *nc = nullptr, *nv = nullptr;
    HIP_TRY(c, hipMalloc(&nc, (b.nset ? b.nset : 1) * 4));
    HIP_TRY(c, hipMalloc(&nv, (m.vals_cap ? m.vals_cap : 1) * 4));
    if (b.nset) HIP_TRY(c, hipMemcpyAsync(nc, b.counts, b.nset * 4, hipMemcpyDeviceToDevice, c->stream));
    if (m.vals_cap) HIP_TRY(c, hipMemcpyAsync(nv, m.vals, m.vals_cap * 4, hipMemcpyDeviceToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    hipFree(c->joined);
    c->joined = nullptr;
    b.counts = nc;
    m.vals = nv;
    return MG_OK;
}

// (re)allocate the gate of `bf`: at most 2^gate_log2 bits, one per 2^gate_shift filter bits
int alloc_gate(mg_ctx *c)
{
    BFState &b = c->bf[MG_BF_ALT];
    u32 S = 6;
    while (((b.size + (1ULL << S) - 1) >> S) > (1ULL << c->gate_log2)) ++S;
    b.gate_shift = S;
    b.n_gate_bits = (b.size + (1ULL << S) - 1) >> S;
    if (b.gate) HIP_TRY(c, hipFree(b.gate));
    b.gate = nullptr;
    const size_t bytes = ((b.n_gate_bits + 63) / 64) * 8;
    HIP_TRY(c, hipMalloc(&b.gate, bytes));
    HIP_TRY(c, hipMemsetAsync(b.gate, 0, bytes, c->stream));
    if (b.pregate) HIP_TRY(c, hipFree(b.pregate));
    b.pregate = nullptr;
    if (c->gate_log2 > c->pregate_log2) { // the gate no longer fits L2: coarse gate of the L2-resident size in front of it
        u32 S1 = 6;
        while (((b.size + (1ULL << S1) - 1) >> S1) > (1ULL << c->pregate_log2)) ++S1;
        b.pre_shift = S1;
        const size_t pbytes = ((((b.size + (1ULL << S1) - 1) >> S1) + 63) / 64) * 8;
        HIP_TRY(c, hipMalloc(&b.pregate, pbytes));
        HIP_TRY(c, hipMemsetAsync(b.pregate, 0, pbytes, c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}

int map_alloc(mg_ctx *c, MapState &m, u32 cap_log2)
{
    const u64 cap = 1ULL << cap_log2;
    m.cap_log2 = cap_log2;
    HIP_TRY(c, hipMalloc(&m.slots, cap * sizeof(MapSlot)));
    hipLaunchKernelGGL(map_clear_kernel, dim3(nblocks(cap)), dim3(TPB), 0, c->stream, m.slots, cap);
    HIP_TRY(c, hipGetLastError());
    return MG_OK;
}
void map_free_table(MapState &m)
{
    hipFree(m.slots);
    m.slots = nullptr;
}
// (re)write the filter's directory into the records: whenever the finalised filter's bits or the table change
int build_bf_entries(mg_ctx *c)
{
    const BFState &alt = c->bf[MG_BF_ALT];
    MapState &m = c->map;
    if (!m.slots || !alt.mode || !alt.blk) return MG_OK;
    const u64 cap = 1ULL << m.cap_log2;
    hipLaunchKernelGGL(bf_entries_clear_kernel, dim3(nblocks(cap)), dim3(TPB), 0, c->stream, m.slots, cap);
    hipLaunchKernelGGL(bf_entries_build_kernel, dim3(nblocks(alt.nwords)), dim3(TPB), 0, c->stream, view(c, MG_BF_ALT), view(c), alt.nwords);
    HIP_TRY(c, hipGetLastError());
    return MG_OK;
}

// make room for `extra` more insertion rows: table load <= 1/4, vals indexable by row
int map_reserve(mg_ctx *c, u64 extra)
{
    MapState &m = c->map;
    const u64 need_rows = m.rows_total + extra;
    if (need_rows >= 0xFFFFFFFFULL) return fail(c, MG_ERR_LIMIT, "exact map: more than 2^32-1 insertion rows");
    if (need_rows > m.vals_cap) {
        TRY(unjoin(c));
        u64 ncap = need_rows + need_rows / 2 + 1024;
        u32 *nv = nullptr;
        HIP_TRY(c, hipMalloc(&nv, ncap * 4));
        HIP_TRY(c, hipMemsetAsync(nv, 0, ncap * 4, c->stream));
        if (m.vals && m.rows_total)
            HIP_TRY(c, hipMemcpyAsync(nv, m.vals, m.rows_total * 4, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (m.vals) hipFree(m.vals);
        m.vals = nv;
        m.vals_cap = ncap;
    }
    // records: load <= 1/4 for the keys, and two filter-directory entries per record at load <= 1/4 as well
    const BFState &alt = c->bf[MG_BF_ALT];
    const u64 dir_entries = alt.mode ? alt.nset : 0;
    u32 want = 10;
    while ((1ULL << want) < need_rows * 4 || (1ULL << want) < dir_entries * 2) ++want;
    if (want > 26 && c->map_dense) {
        // Beyond 2^26 records (4 GB) what a probe costs is the PAGE it lands on, not the length of its walk: at whole-genome scale
        // scan_probe_kernel spent its time in address translation (UTCL2 busy 93 %, profiles/r03_pmc_c4share_before.txt), and the next
        // record of a walk is nearly always on the same page.  Half the table: key load <= 1/2, directory load <= 1/2.
        u32 w2 = 26;
        while ((1ULL << w2) < need_rows * 2 || (1ULL << w2) < dir_entries) ++w2;
        if (w2 < want) want = w2;
    }
    if (!m.slots) {
        TRY(map_alloc(c, m, want));
        return build_bf_entries(c);
    }
    if (want > m.cap_log2) {
        MapView ov{};
        ov.slots = m.slots;
        ov.cap_log2 = m.cap_log2;
        ov.klen = c->k;
        ov.home_mul = view(c).home_mul;
        m.slots = nullptr;
        TRY(map_alloc(c, m, want));
        MapView nv = view(c);
        hipLaunchKernelGGL(map_rehash_kernel, dim3(nblocks(1ULL << ov.cap_log2)), dim3(TPB), 0, c->stream, ov, nv, c->bf[MG_BF_ALT].mod);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        hipFree(ov.slots);
        return build_bf_entries(c); // the new records start without the filter's directory
    }
    return MG_OK;
}

int check_rows(mg_ctx *c, const void *rows, size_t stride, size_t n)
{
    if (!c) return MG_ERR_ARG;
    if (n && !rows) return fail(c, MG_ERR_ARG, "rows is NULL");
    if (stride < 2) return fail(c, MG_ERR_ARG, "row stride %zu too small", stride);
    return MG_OK;
}
int check_which(mg_ctx *c, int which)
{
    if (!c) return MG_ERR_ARG;
    if (which != MG_BF_ALT && which != MG_BF_CTX) return fail(c, MG_ERR_ARG, "which must be MG_BF_ALT or MG_BF_CTX");
    return MG_OK;
}

template <int OP>
int run_rows(mg_ctx *c, int which, const char *rows, size_t stride, size_t n, const void *counters, const u8 *is_ref,
             void *host_out, size_t out_elem, u8 *host_irregular)
{
    if (n == 0) return MG_OK;
    void *d_rows, *d_cnt = nullptr, *d_isref = nullptr, *d_out = nullptr, *d_irr = nullptr;
    TRY(upload(c, c->s_rows, rows, stride * n, &d_rows));
    if (counters) TRY(upload(c, c->s_aux, counters, 4 * n, &d_cnt));
    if (is_ref) TRY(upload(c, c->s_misc[0], is_ref, n, &d_isref));
    if (host_out) TRY(scratch(c, c->s_out, out_elem * n, &d_out));
    if (host_irregular) TRY(scratch(c, c->s_irr, n, &d_irr));
    hipLaunchKernelGGL(rows_kernel<OP>, dim3(nblocks(n)), dim3(TPB), 0, c->stream, (const u8 *)d_rows, stride, n,
                       view(c, which), view(c), (const u32 *)d_cnt, (const u8 *)d_isref, d_out, (u8 *)d_irr);
    HIP_TRY(c, hipGetLastError());
    if (host_out) HIP_TRY(c, hipMemcpyAsync(host_out, d_out, out_elem * n, hipMemcpyDeviceToHost, c->stream));
    if (host_irregular) HIP_TRY(c, hipMemcpyAsync(host_irregular, d_irr, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}

// canonical key of an irregular row as KMAP::canonical returns it (kmap.hpp:86-97):
// bookkeeping for keys the device table cannot represent; the scan never sees them.
std::string host_irregular_key(const char *row, size_t stride)
{
    size_t k = strnlen(row, stride);
    std::string rc(k, '\0');
    auto comp = [](unsigned char ch) -> char {
        switch (ch) {
        case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A'; case 'N': return 'N';
        case 'a': return 'T'; case 'c': return 'G'; case 'g': return 'G'; case 't': return 'A'; case 'n': return 'N';
        default: return 0;
        }
    };
    for (size_t i = 0; i < k; ++i) rc[i] = comp((unsigned char)row[k - 1 - i]);
    std::string fw(row, k);
    bool fwd = false;
    for (size_t i = 0; i < k; ++i)
        if ((unsigned char)fw[i] != (unsigned char)rc[i]) {
            fwd = (unsigned char)fw[i] < (unsigned char)rc[i];
            break;
        }
    std::string can = fwd ? fw : rc;
    return std::string(can.c_str()); // cut at the first NUL
}

int fill_geno_params(mg_ctx *c, float error_rate, int max_cov, int haploid, GenoParams *p)
{
    if (!c->d_ln) {
        std::vector<double> t(MG_LN_TABLE);
        t[0] = 0.0;
        for (int n = 1; n < MG_LN_TABLE; ++n) t[n] = std::log((double)n);
        HIP_TRY(c, hipMalloc(&c->d_ln, sizeof(double) * MG_LN_TABLE));
        HIP_TRY(c, hipMemcpy(c->d_ln, t.data(), sizeof(double) * MG_LN_TABLE, hipMemcpyHostToDevice));
    }
    if (!c->d_eps) HIP_TRY(c, hipMalloc(&c->d_eps, sizeof(float) * 2 * MG_EPS_TABLE));
    if (!(c->eps_for == error_rate)) {
        std::vector<float> t(2 * MG_EPS_TABLE, 0.f);
        for (int A = 0; A < MG_EPS_TABLE; ++A) {
            t[A] = std::log(error_rate / (float)(unsigned long)(A - 1));                // float overload
            t[MG_EPS_TABLE + A] = std::log(error_rate / (float)(unsigned long)(A - 2)); // float overload
        }
        HIP_TRY(c, hipMemcpyAsync(c->d_eps, t.data(), sizeof(float) * 2 * MG_EPS_TABLE, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        c->eps_for = error_rate;
    }
    p->ln_tab = c->d_ln;
    p->c_err1 = c->d_eps;
    p->c_err2 = c->d_eps + MG_EPS_TABLE;
    p->c_hom = std::log(1 - error_rate);
    p->c_het = std::log((1 - error_rate) / 2);
    p->error_rate = error_rate;
    p->max_cov = max_cov;
    p->haploid = haploid;
    return MG_OK;
}

} // namespace

// ---- lifetime -------------------------------------------------------------------

// Every entry point may be called from any host thread: make the context's device current for the calling thread
// (HIP's current device is per thread; a thread that never called hipSetDevice sits on device 0).
// for DeviceGuard.  KEEP: this entry point leaves the counters alone, or changes them together with the records' copies (it may READ the
// vectors: they are brought up to date first).  LAZY: the same, and it never looks at vals[] / counts[] while the copies are current
// (the scans, the record loop's device forms, the bookkeeping calls): lazy vectors stay lazy.
constexpr int KEEP = 1, LAZY = 2;
// vals[] / counts[] from the records' copies, where scans have added to the copies alone
void vectors_current(const mg_ctx *c)
{
    if (!c->vec_stale) return;
    c->vec_stale = false;
    c->vec_zero = false;
    if (!c->map.slots || !c->rec_ok) return; // (cannot happen: lazy scans need current copies, and nothing drops them without coming through here)
    MapView m = view(c);
    m.epoch = c->rec_epoch;
    hipLaunchKernelGGL(rec_collect_kernel, dim3((unsigned)std::min<u64>(nblocks(1ULL << m.cap_log2), 1u << 20)), dim3(TPB), 0, c->stream, m,
                       c->bf[MG_BF_ALT].mode ? c->bf[MG_BF_ALT].counts : (u32 *)nullptr, c->rec_epoch);
}
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(const mg_ctx *c, int keeps_record_counters = 0)
    {
        int cur = -1;
        if (c && hipGetDevice(&cur) == hipSuccess && cur != c->device && hipSetDevice(c->device) == hipSuccess) prev = cur;
        if (c && keeps_record_counters != LAZY) vectors_current(c);
        if (c && !keeps_record_counters) c->rec_ok = false, c->vec_zero = false;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) hipSetDevice(prev); // leave the caller's thread as it was
    }
    DeviceGuard(const DeviceGuard &) = delete;
    DeviceGuard &operator=(const DeviceGuard &) = delete;
};

MG_EXPORT int mg_create(mg_ctx **out, int device, uint32_t k, uint32_t ref_k, uint64_t bf_bits)
{
    if (!out) return MG_ERR_ARG;
    *out = nullptr;
    if (k == 0 || k > MG_MAX_KMER || ref_k < k || ref_k > MG_MAX_KMER || bf_bits == 0) return MG_ERR_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) return MG_ERR_HIP;
    if (hipSetDevice(device) != hipSuccess) return MG_ERR_HIP;
    mg_ctx *c = new mg_ctx();
    c->device = device;
    c->k = k;
    c->ref_k = ref_k;
    if (hipStreamCreate(&c->own_stream) != hipSuccess) {
        delete c;
        return MG_ERR_HIP;
    }
    c->stream = c->own_stream;
    for (auto &e : c->ev)
        if (hipEventCreate(&e) != hipSuccess) {
            delete c;
            return MG_ERR_HIP;
        }
    for (int w = 0; w < 2; ++w) {
        BFState &b = c->bf[w];
        b.size = bf_bits;
        b.nwords = (bf_bits + 63) / 64;
        b.n_blk = (b.nwords + 7) / 8;
        b.mod = make_mod(bf_bits);
        // whole 512-bit blocks (zero padded): lookups read the block of a bit in one go
        if (hipMalloc(&b.words, b.n_blk * 64) != hipSuccess || hipMemsetAsync(b.words, 0, b.n_blk * 64, c->stream) != hipSuccess) {
            mg_destroy(c);
            return MG_ERR_NOMEM;
        }
    }
    if (alloc_gate(c) != MG_OK || hipMalloc(&c->d_hit_count, 32) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) {
        mg_destroy(c);
        return MG_ERR_HIP;
    }
    *out = c;
    return MG_OK;
}

MG_EXPORT int mg_destroy(mg_ctx *c)
{
    const DeviceGuard on_device(c);
    if (!c) return MG_OK;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
    comm_drop(c);
    if (c->ev_x) hipEventDestroy(c->ev_x);
    for (auto &e : c->ev_xs)
        if (e) hipEventDestroy(e);
    if (c->xstream) hipStreamDestroy(c->xstream);
    hipFree(c->d_xmax);
    hipFree(c->s_pack.p);
    for (auto &e : c->ev_b)
        if (e) hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) {
        if (c->ev_up[i]) hipEventDestroy(c->ev_up[i]);
        if (c->ev_free[i]) hipEventDestroy(c->ev_free[i]);
        for (auto &q : c->s_stage[i]) hipFree(q.p);
        hipFree(c->s_raw[i].p);
    }
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->h_gt_stage) hipHostFree(c->h_gt_stage);
    hipFree(c->d_kmc_lut);
    if (c->joined) { // the two counter arrays alias one allocation
        hipFree(c->joined);
        c->bf[MG_BF_ALT].counts = nullptr;
        c->map.vals = nullptr;
    }
    for (auto &b : c->bf) {
        hipFree(b.words);
        hipFree(b.blk);
        hipFree(b.counts);
        hipFree(b.gate);
        hipFree(b.pregate);
        hipFree(b.pos_set);
    }
    map_free_table(c->map);
    hipFree(c->map.vals);
    for (Scratch *s : {&c->s_rows, &c->s_aux, &c->s_out, &c->s_irr}) hipFree(s->p);
    for (auto &s : c->s_open) hipFree(s.p);
    for (auto &s : c->s_bin) hipFree(s.p);
    for (auto &s : c->s_spill) hipFree(s.p);
    hipFree(c->d_bin_meta);
    hipFree(c->d_tk_meta);
    for (auto &q : c->s_tk) hipFree(q.p);
    for (auto &q : c->s_sb) hipFree(q.p);
    for (auto &s : c->s_hit) hipFree(s.p);
    for (auto &s : c->s_misc) hipFree(s.p);
    for (auto &s : c->s_blk) hipFree(s.p);
    for (auto &s : c->s_gt) hipFree(s.p);
    hipFree(c->s_scan.p);
    hipFree(c->d_gen_count);
    hipFree(c->d_hit_count);
    hipFree(c->d_ln);
    hipFree(c->d_eps);
    hipFree(c->d_ref);
    hipFree(c->d_ref2);
    hipFree(c->d_refbad);
    for (auto &e : c->ev)
        if (e) hipEventDestroy(e);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    return MG_OK;
}

MG_EXPORT const char *mg_last_error(const mg_ctx *c) { return c ? c->err.c_str() : "null context"; }

MG_EXPORT int mg_set_stream(mg_ctx *c, void *s)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return MG_OK;
}
MG_EXPORT int mg_synchronize(mg_ctx *c)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}
MG_EXPORT int mg_set_option(mg_ctx *c, const char *name, int64_t value)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c || !name) return MG_ERR_ARG;
    if (!strcmp(name, "use_summary")) c->use_summary = value != 0;
    else if (!strcmp(name, "scan_rows")) c->scan_rows = (int)value;
    else if (!strcmp(name, "scan_ablate")) c->scan_ablate = (int)value;
    else if (!strcmp(name, "blocks_set_limit")) c->blocks_set_limit = (u32)std::min<int64_t>(std::max<int64_t>(value, 0), BK_SET_CAP / 2);
    else if (!strcmp(name, "use_pregate")) c->use_pregate = value < 0 ? 0 : value > 2 ? 2 : (int)value;
    else if (!strcmp(name, "use_partition")) c->use_partition = value != 0;
    else if (!strcmp(name, "use_flat_tier")) c->use_flat_tier = value != 0;
    else if (!strcmp(name, "map_dense")) c->map_dense = value != 0;
    else if (!strcmp(name, "use_packed_pool")) c->use_packed_pool = value != 0;
    else if (!strcmp(name, "use_snp_chains")) c->use_snp_chains = value != 0;
    else if (!strcmp(name, "use_hit_entries")) c->use_hit_entries = value != 0;
    else if (!strcmp(name, "blocks_round_log2")) {
        if (value < 10 || value > 24) return fail(c, MG_ERR_ARG, "blocks_round_log2: 10..24");
        c->blocks_round_log2 = (int)value;
    }
    else if (!strcmp(name, "map_ordered")) {
        if (c->map.slots && (value != 0) != (c->map_ordered != 0)) return fail(c, MG_ERR_STATE, "map_ordered is a layout: set it before the index is built or loaded");
        c->map_ordered = value != 0;
    }
    else if (!strcmp(name, "use_record_counters")) {
        c->use_record_counters = value < 0 ? 0 : value > 2 ? 2 : (int)value;
        c->rec_ok = false; // (the next scan brings the copies up to date)
    }
    else if (!strcmp(name, "use_ctx_set")) {
        c->use_ctx_set = (int)value; // (2: whatever the filter's size -- tests)
        c->bf[MG_BF_CTX].pos_set_valid = false;
    }
    else if (!strcmp(name, "use_packed_ref_scan")) c->use_packed_ref_scan = value != 0;
    else if (!strcmp(name, "probe_grid")) c->probe_grid = value > 0 ? (int)value : 2048;
    else if (!strcmp(name, "use_tickets")) c->use_tickets = value != 0;
    else if (!strcmp(name, "use_sub")) c->use_sub = value != 0;
    else if (!strcmp(name, "use_chain_order")) c->use_chain_order = value != 0;
    else if (!strcmp(name, "use_chain_kernel")) c->use_chain_kernel = value != 0;
    else if (!strcmp(name, "use_snp_kernel")) c->use_snp_kernel = value != 0;
    else if (!strcmp(name, "exchange_pack")) c->exchange_pack = (int)std::max<int64_t>(0, std::min<int64_t>(2, value));
    else if (!strcmp(name, "exchange_pack_min_mb")) c->exchange_pack_min_mb = (int)std::max<int64_t>(0, value);
    else if (!strcmp(name, "lazy_vectors")) c->lazy_vectors = value != 0;
    else if (!strcmp(name, "sub_min_log2")) c->sub_min_log2 = (int)value;
    else if (!strcmp(name, "sub_words_log2")) {
        if (value < 0 || value > SB_WORDS_LOG2) return fail(c, MG_ERR_ARG, "sub_words_log2 must be 0..%d", SB_WORDS_LOG2);
        c->sub_words_log2 = (int)value;
    } else if (!strcmp(name, "sub_bins_log2")) c->sub_bins_log2 = (int)std::max<int64_t>(1, std::min<int64_t>(10, value));
    else if (!strcmp(name, "sub_split")) c->sub_split = (int)std::max<int64_t>(1, std::min<int64_t>(64, value));
    else if (!strcmp(name, "sub_grid")) c->sub_grid = (int)std::max<int64_t>(0, std::min<int64_t>(4096, value));
    else if (!strcmp(name, "ticket_min_log2")) c->ticket_min_log2 = (int)value;
    else if (!strcmp(name, "scan_chunk_log2")) c->chunk_log2 = (int)std::min<int64_t>(27, std::max<int64_t>(10, value));
    else if (!strcmp(name, "ticket_gate_grid")) c->tkg_grid = value > 0 ? (int)std::max<int64_t>(8, std::min<int64_t>(4096, value / 8 * 8)) : 0;
    else if (!strcmp(name, "hits_grid")) c->hits_grid = value > 0 ? (int)value : 1024;
    else if (!strcmp(name, "scan_bin_cap")) c->bin_cap = value > 0 ? (u64)value : 0;
    else if (!strcmp(name, "scan_bin_ring")) {
        if (value != 0 && value != 64 && value != 128 && value != 256) return fail(c, MG_ERR_ARG, "scan_bin_ring must be 0, 64, 128 or 256");
        c->bin_ring = (int)value;
    }
    else if (!strcmp(name, "scan_bin_rows")) c->bin_rows = value == 2 ? 2 : 4;
    else if (!strcmp(name, "pregate_log2")) {
        if (c->map.rows_total || c->gate_dirty) return fail(c, MG_ERR_STATE, "pregate_log2 must be set before the first insert");
        if (value < 8 || value > 30) return fail(c, MG_ERR_ARG, "pregate_log2 must be 8..30");
        c->pregate_log2 = (int)value;
        return alloc_gate(c);
    }
    else if (!strcmp(name, "scan_variant")) c->scan_variant = (int)value & 2;
    else if (!strcmp(name, "scan_grid")) c->scan_grid = value > 0 ? (int)value : 8192;
    else if (!strcmp(name, "gate_log2") || !strcmp(name, "gate_k")) {
        if (c->map.rows_total || c->gate_dirty) return fail(c, MG_ERR_STATE, "%s must be set before the first insert", name);
        if (!strcmp(name, "gate_k")) {
            if (value < 1 || value > 4) return fail(c, MG_ERR_ARG, "gate_k must be 1..4");
            c->gate_k = (int)value;
        } else {
            c->gate_log2 = (int)value;
            c->gate_fixed = true;
        }
        return alloc_gate(c);
    }
    else return fail(c, MG_ERR_ARG, "unknown option %s", name);
    return MG_OK;
}

MG_EXPORT int mg_get_option(mg_ctx *c, const char *name, int64_t *value)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c || !name || !value) return MG_ERR_ARG;
    if (!strcmp(name, "use_summary")) *value = c->use_summary;
    else if (!strcmp(name, "use_pregate")) *value = c->use_pregate;
    else if (!strcmp(name, "use_partition")) *value = c->use_partition;
    else if (!strcmp(name, "use_flat_tier")) *value = c->use_flat_tier;
    else if (!strcmp(name, "map_dense")) *value = c->map_dense;
    else if (!strcmp(name, "use_packed_pool")) *value = c->use_packed_pool;
    else if (!strcmp(name, "use_snp_chains")) *value = c->use_snp_chains;
    else if (!strcmp(name, "use_hit_entries")) *value = c->use_hit_entries;
    else if (!strcmp(name, "use_chain_order")) *value = c->use_chain_order;
    else if (!strcmp(name, "blocks_listed_chains")) { // diagnostic: chains fw_chain_kernel handed to the list path in the most recent record loop (waits for it)
        unsigned long long total = 0;
        if (c->last_rounds && c->s_blk[5].p) {
            std::vector<unsigned long long> h((size_t)FW_ROUND_COUNTERS * c->last_rounds);
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipMemcpy(h.data(), c->s_blk[5].p, 8 * h.size(), hipMemcpyDeviceToHost));
            for (u64 r = 0; r < c->last_rounds; ++r) total += h[(size_t)FW_ROUND_COUNTERS * r + 3];
        }
        *value = (int64_t)total;
    }
    else if (!strcmp(name, "use_chain_kernel")) *value = c->use_chain_kernel;
    else if (!strcmp(name, "use_snp_kernel")) *value = c->use_snp_kernel;
    else if (!strcmp(name, "blocks_round_log2")) *value = c->blocks_round_log2;
    else if (!strcmp(name, "map_ordered")) *value = c->map_ordered;
    else if (!strcmp(name, "use_record_counters")) *value = c->use_record_counters;
    else if (!strcmp(name, "record_counters_live")) *value = view(c).epoch != 0; // diagnostic: would a lookup launched now read the records' copies?
    else if (!strcmp(name, "use_ctx_set")) *value = c->use_ctx_set;
    else if (!strcmp(name, "ctx_set_log2")) *value = c->bf[MG_BF_CTX].pos_set_valid ? c->bf[MG_BF_CTX].pos_set_log2 : 0;
    else if (!strcmp(name, "use_packed_ref_scan")) *value = c->use_packed_ref_scan;
    else if (!strcmp(name, "gate_log2")) *value = c->gate_log2;
    else if (!strcmp(name, "gate_k")) *value = c->gate_k;
    else if (!strcmp(name, "pregate_log2")) *value = c->pregate_log2;
    else if (!strcmp(name, "pregate_k")) *value = c->bf[MG_BF_ALT].pregate && pregate_on(c) ? c->pre_k : 0;
    else if (!strcmp(name, "scan_bins")) *value = c->last_bins;
    else if (!strcmp(name, "scan_tickets")) *value = c->last_tickets;
    else if (!strcmp(name, "scan_subs")) *value = c->last_subs;
    else if (!strcmp(name, "use_sub")) *value = c->use_sub;
    else if (!strcmp(name, "exchange_pack")) *value = c->exchange_pack;
    else if (!strcmp(name, "exchange_packed")) *value = c->last_exchange_packed;
    else if (!strcmp(name, "lazy_vectors")) *value = c->lazy_vectors;
    else if (!strcmp(name, "vectors_stale")) *value = c->vec_stale;
    else if (!strcmp(name, "ticket_gate_grid")) *value = c->tkg_grid;
    else if (!strcmp(name, "use_tickets")) *value = c->use_tickets;
    else if (!strcmp(name, "scan_spilled")) { // rows of the last chunk that took the spill list
        unsigned long long t = 0;
        if (c->last_bins || c->last_tickets || c->last_subs) {
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            HIP_TRY(c, hipMemcpy(&t, c->last_subs ? c->s_sb[3].p : c->last_tickets ? (void *)c->d_tk_meta : (void *)c->d_bin_meta, 8, hipMemcpyDeviceToHost));
        }
        *value = (int64_t)t;
    }
    else return fail(c, MG_ERR_ARG, "unknown option %s", name);
    return MG_OK;
}

// ---- BF ---------------------------------------------------------------------------

MG_EXPORT int mg_bf_insert(mg_ctx *c, int which, const char *rows, size_t stride, size_t n)
{
    if (c && (which == MG_BF_ALT || which == MG_BF_CTX)) c->bf[which].pos_set_valid = false; // (the bits are about to change)
    const DeviceGuard on_device(c);
    TRY(check_which(c, which));
    TRY(check_rows(c, rows, stride, n));
    // the reference lets add_key run in read mode too (the bit is set, the rank goes stale); refuse that
    if (c->bf[which].mode) return fail(c, MG_ERR_STATE, "mg_bf_insert after mg_bf_finalize");
    if (which == MG_BF_ALT) c->gate_dirty = true;
    return run_rows<OP_BF_INSERT>(c, which, rows, stride, n, nullptr, nullptr, nullptr, 0, nullptr);
}
MG_EXPORT int mg_bf_test(mg_ctx *c, int which, const char *rows, size_t stride, size_t n, uint8_t *out)
{
    const DeviceGuard on_device(c, KEEP);
    TRY(check_which(c, which));
    TRY(check_rows(c, rows, stride, n));
    if (n && !out) return fail(c, MG_ERR_ARG, "out is NULL");
    return run_rows<OP_BF_TEST>(c, which, rows, stride, n, nullptr, nullptr, out, 1, nullptr);
}
MG_EXPORT int mg_debug_bf_index(mg_ctx *c, int which, const char *rows, size_t stride, size_t n, uint64_t *out)
{
    const DeviceGuard on_device(c, KEEP);
    TRY(check_which(c, which));
    TRY(check_rows(c, rows, stride, n));
    if (n && !out) return fail(c, MG_ERR_ARG, "out is NULL");
    return run_rows<OP_BF_INDEX>(c, which, rows, stride, n, nullptr, nullptr, out, 8, nullptr);
}

MG_EXPORT int mg_bf_finalize(mg_ctx *c, int which)
{
    if (c && (which == MG_BF_ALT || which == MG_BF_CTX)) c->bf[which].pos_set_valid = false;
    const DeviceGuard on_device(c);
    TRY(check_which(c, which));
    BFState &b = c->bf[which];
    if (b.blk) {
        hipFree(b.blk);
        b.blk = nullptr;
    }
    HIP_TRY(c, hipMalloc(&b.blk, (b.n_blk + 1) * 4));
    const u64 n_tiles = nblocks(b.n_blk + 1);
    void *d_tiles;
    TRY(scratch(c, c->s_misc[1], (n_tiles + 1) * 4, &d_tiles));
    unsigned long long *d_total = c->d_hit_count;
    hipLaunchKernelGGL(blk_pop_kernel, dim3((unsigned)n_tiles), dim3(TPB), 0, c->stream, b.words, b.nwords, b.n_blk, b.blk,
                       (u32 *)d_tiles);
    TRY(launch_tile_scan(c, (u32 *)d_tiles, n_tiles, d_total));
    HIP_TRY(c, hipGetLastError());
    unsigned long long total = 0;
    HIP_TRY(c, hipMemcpyAsync(&total, d_total, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (total >= 0xFFFFFFFFULL) return fail(c, MG_ERR_LIMIT, "filter holds %llu set bits (>= 2^32-1)", total);
    hipLaunchKernelGGL(blk_add_kernel, dim3((unsigned)n_tiles), dim3(TPB), 0, c->stream, b.blk, b.n_blk, (const u32 *)d_tiles,
                       (u32)total);
    HIP_TRY(c, hipGetLastError());
    if (which == MG_BF_ALT) TRY(unjoin(c));
    b.nset = total;
    if (b.counts) hipFree(b.counts);
    b.counts = nullptr;
    HIP_TRY(c, hipMalloc(&b.counts, (total ? total : 1) * 4));
    HIP_TRY(c, hipMemsetAsync(b.counts, 0, (total ? total : 1) * 4, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    b.mode = 1;
    if (which == MG_BF_ALT && !c->gate_fixed) {
        // size the gate for what the index holds: >= 12 bits per entry (set bf bits + exact-map keys), a power
        // of two, never below the default.  2e6 entries (C3) -> 2^25 bits = 4 MiB; 2e7 -> 2^28 = 32 MiB.
        const u64 entries = b.nset + c->map.rows_total;
        int want = 25;
        while (want < 34 && (1ULL << want) < 12 * entries) ++want;
        while (want > 6 && (1ULL << want) > b.size) --want;
        // an index that takes the sub-slice form (tickets filed under LDS-sized pieces of the gate) keeps its gate within
        // 2^sub_bins_log2 <= SB_MAXB pieces: 256 MiB at full-size pieces
        if (c->use_summary && c->use_sub && want >= c->sub_min_log2) want = std::min(want, c->sub_words_log2 + 6 + c->sub_bins_log2);
        // coarse gate: the bits per entry that minimise its false-positive rate at this load (ln 2 * bits / entries)
        int pk = entries ? (int)std::lround(0.6931 * (double)(1ULL << c->pregate_log2) / (double)entries) : 4;
        pk = pk < 1 ? 1 : pk > 4 ? 4 : pk;
        // a coarse gate that would let more than 3 of 4 rows through costs an L2 probe per row and saves little
        c->pre_skip = std::pow(1.0 - std::exp(-(double)pk * (double)entries / (double)(1ULL << c->pregate_log2)), pk) > 0.75;
        if (want != c->gate_log2 || (want > c->pregate_log2 && pk != c->pre_k)) {
            c->gate_log2 = want;
            c->pre_k = pk;
            TRY(alloc_gate(c));
            hipLaunchKernelGGL(gate_from_bits_kernel, dim3(nblocks(b.nwords)), dim3(TPB), 0, c->stream, view(c, MG_BF_ALT), b.nwords);
            if (c->map.slots)
                hipLaunchKernelGGL(map_gate_kernel, dim3(nblocks(1ULL << c->map.cap_log2)), dim3(TPB), 0, c->stream, view(c),
                                   view(c, MG_BF_ALT));
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    if (which == MG_BF_ALT) { // the filter's directory inside the exact map's records (MapSlot)
        const bool had = c->map.slots != nullptr;
        const u32 before = c->map.cap_log2;
        TRY(map_reserve(c, 0)); // creates or grows the table if the directory needs it, and then writes the directory
        if (had && c->map.cap_log2 == before) TRY(build_bf_entries(c));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return MG_OK;
}

MG_EXPORT int mg_bf_increment(mg_ctx *c, int which, const char *rows, size_t stride, size_t n, const uint32_t *counters)
{
    const DeviceGuard on_device(c);
    TRY(check_which(c, which));
    TRY(check_rows(c, rows, stride, n));
    if (!c->bf[which].mode) return fail(c, MG_ERR_STATE, "BF::increment in write mode returns false");
    if (n && !counters) return fail(c, MG_ERR_ARG, "counters is NULL");
    return run_rows<OP_BF_INC>(c, which, rows, stride, n, counters, nullptr, nullptr, 0, nullptr);
}
MG_EXPORT int mg_bf_get_count(mg_ctx *c, int which, const char *rows, size_t stride, size_t n, uint16_t *out)
{
    const DeviceGuard on_device(c, KEEP);
    TRY(check_which(c, which));
    TRY(check_rows(c, rows, stride, n));
    if (n && !out) return fail(c, MG_ERR_ARG, "out is NULL");
    if (!c->bf[which].mode) { // bloom_filter.hpp:117: write mode -> 0
        memset(out, 0, 2 * n);
        return MG_OK;
    }
    return run_rows<OP_BF_GET>(c, which, rows, stride, n, nullptr, nullptr, out, 2, nullptr);
}
MG_EXPORT int mg_bf_info(mg_ctx *c, int which, uint64_t *size_bits, uint64_t *n_set, int *mode)
{
    const DeviceGuard on_device(c, LAZY);
    TRY(check_which(c, which));
    if (size_bits) *size_bits = c->bf[which].size;
    if (n_set) *n_set = c->bf[which].nset;
    if (mode) *mode = c->bf[which].mode;
    return MG_OK;
}

// ---- KMAP ---------------------------------------------------------------------------

MG_EXPORT int mg_map_insert(mg_ctx *c, const char *rows, size_t stride, size_t n)
{
    const DeviceGuard on_device(c);
    TRY(check_rows(c, rows, stride, n));
    if (n == 0) return MG_OK;
    TRY(map_reserve(c, n));
    void *d_rows, *d_irr;
    TRY(upload(c, c->s_rows, rows, stride * n, &d_rows));
    TRY(scratch(c, c->s_irr, n, &d_irr));
    hipLaunchKernelGGL(map_insert_kernel, dim3(nblocks(n)), dim3(TPB), 0, c->stream, (const u8 *)d_rows, stride, n, view(c),
                       view(c, MG_BF_ALT), (u32)c->map.rows_total, (u8 *)d_irr);
    HIP_TRY(c, hipGetLastError());
    std::vector<u8> irr(n);
    HIP_TRY(c, hipMemcpyAsync(irr.data(), d_irr, n, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->map.rows_total += n;
    for (size_t i = 0; i < n; ++i)
        if (irr[i]) c->map.irregular[host_irregular_key(rows + i * stride, stride)] = 0;
    return MG_OK;
}
MG_EXPORT int mg_map_test(mg_ctx *c, const char *rows, size_t stride, size_t n, uint8_t *out)
{
    const DeviceGuard on_device(c, KEEP);
    TRY(check_rows(c, rows, stride, n));
    if (n == 0) return MG_OK;
    if (!out) return fail(c, MG_ERR_ARG, "out is NULL");
    if (!c->map.slots) TRY(map_reserve(c, 0));
    std::vector<u8> irr(n);
    TRY(run_rows<OP_MAP_TEST>(c, 0, rows, stride, n, nullptr, nullptr, out, 1, irr.data()));
    for (size_t i = 0; i < n; ++i)
        if (irr[i]) out[i] = c->map.irregular.count(host_irregular_key(rows + i * stride, stride)) ? 1 : 0;
    return MG_OK;
}
MG_EXPORT int mg_map_increment(mg_ctx *c, const char *rows, size_t stride, size_t n, const int32_t *counters)
{
    const DeviceGuard on_device(c);
    TRY(check_rows(c, rows, stride, n));
    if (n == 0) return MG_OK;
    if (!counters) return fail(c, MG_ERR_ARG, "counters is NULL");
    if (!c->map.slots) TRY(map_reserve(c, 0));
    std::vector<u8> irr(n);
    TRY(run_rows<OP_MAP_INC>(c, 0, rows, stride, n, counters, nullptr, nullptr, 0, irr.data()));
    for (size_t i = 0; i < n; ++i)
        if (irr[i]) {
            auto it = c->map.irregular.find(host_irregular_key(rows + i * stride, stride));
            if (it != c->map.irregular.end()) it->second = (int32_t)((uint32_t)it->second + (uint32_t)counters[i]);
        }
    return MG_OK;
}
MG_EXPORT int mg_map_get_count(mg_ctx *c, const char *rows, size_t stride, size_t n, int32_t *out)
{
    const DeviceGuard on_device(c, KEEP);
    TRY(check_rows(c, rows, stride, n));
    if (n == 0) return MG_OK;
    if (!out) return fail(c, MG_ERR_ARG, "out is NULL");
    if (!c->map.slots) TRY(map_reserve(c, 0));
    std::vector<u8> irr(n);
    TRY(run_rows<OP_MAP_GET>(c, 0, rows, stride, n, nullptr, nullptr, out, 4, irr.data()));
    for (size_t i = 0; i < n; ++i)
        if (irr[i]) {
            auto it = c->map.irregular.find(host_irregular_key(rows + i * stride, stride));
            out[i] = it != c->map.irregular.end() ? it->second : 0;
        }
    return MG_OK;
}

namespace {
// distinct regular keys on the device table
int map_dump(mg_ctx *c, std::vector<u64> *klo, std::vector<u64> *khi, std::vector<u32> *ids)
{
    MapState &m = c->map;
    klo->clear();
    khi->clear();
    ids->clear();
    if (!m.slots) return MG_OK;
    const u64 cap = 1ULL << m.cap_log2;
    const u64 maxn = m.rows_total < cap ? m.rows_total : cap;
    if (maxn == 0) return MG_OK;
    void *d_lo, *d_hi, *d_id;
    TRY(scratch(c, c->s_misc[2], maxn * 8, &d_lo));
    TRY(scratch(c, c->s_misc[3], maxn * 8, &d_hi));
    TRY(scratch(c, c->s_misc[4], maxn * 4, &d_id));
    HIP_TRY(c, hipMemsetAsync(c->d_hit_count, 0, 8, c->stream));
    hipLaunchKernelGGL(map_dump_kernel, dim3(nblocks(cap)), dim3(TPB), 0, c->stream, view(c), (u64 *)d_lo, (u64 *)d_hi,
                       (u32 *)d_id, c->d_hit_count);
    HIP_TRY(c, hipGetLastError());
    unsigned long long cnt = 0;
    HIP_TRY(c, hipMemcpyAsync(&cnt, c->d_hit_count, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    klo->resize(cnt);
    khi->resize(cnt);
    ids->resize(cnt);
    if (cnt) {
        HIP_TRY(c, hipMemcpy(klo->data(), d_lo, cnt * 8, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(khi->data(), d_hi, cnt * 8, hipMemcpyDeviceToHost));
        HIP_TRY(c, hipMemcpy(ids->data(), d_id, cnt * 4, hipMemcpyDeviceToHost));
    }
    return MG_OK;
}
} // namespace

MG_EXPORT int mg_map_size(mg_ctx *c, uint64_t *n_keys)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c || !n_keys) return MG_ERR_ARG;
    std::vector<u64> a, b;
    std::vector<u32> ids;
    TRY(map_dump(c, &a, &b, &ids));
    *n_keys = a.size() + c->map.irregular.size();
    return MG_OK;
}

// ---- reference scan --------------------------------------------------------------------

namespace {
int reference_pack(mg_ctx *c, const u8 *d_ascii, size_t len, u64 *ref2, u32 *refbad)
{
    const u64 n_words = (len + 31) / 32 + 4; // (+ the padding the span readers may touch)
    hipLaunchKernelGGL(ref_pack_kernel, dim3(nblocks(n_words)), dim3(TPB), 0, c->stream, d_ascii, (u64)len, ref2, refbad, n_words);
    HIP_TRY(c, hipGetLastError());
    return MG_OK;
}
// the windows [w0, w0 + nw) of a contig whose ASCII bytes from window w0 on are at d_ascii and whose base w0 sits at `at` in
// the packed arrays: the packed kernel, then the byte-wise one for what it leaves out
int ref_scan_windows(mg_ctx *c, const u8 *d_ascii, const u64 *ref2, const u32 *refbad, u64 at, u64 w0, u64 nw)
{
    const bool packed = c->k >= 1 && c->ref_k <= MG_MAX_PACKED_K && c->use_packed_ref_scan;
    if (packed) {
        const unsigned grid = (unsigned)std::min<u64>(nblocks((nw + REF_SCAN_W - 1) / REF_SCAN_W), 1u << 16);
        // (window numbers are contig-wide only through w == 0 and w < k: the packed kernel is handed the slice's own numbering
        //  shifted by w0 through `at` -- see its `w` -- so slices after the first never meet the quirk)
        if (c->k == 35 && c->ref_k == 43)
            hipLaunchKernelGGL((ref_scan_packed_kernel<35, 43>), dim3(grid), dim3(TPB), 0, c->stream, ref2, refbad, at, w0, nw, (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT), view(c, MG_BF_CTX));
        else if (c->k == 35 && c->ref_k == 63)
            hipLaunchKernelGGL((ref_scan_packed_kernel<35, 63>), dim3(grid), dim3(TPB), 0, c->stream, ref2, refbad, at, w0, nw, (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT), view(c, MG_BF_CTX));
        else
            hipLaunchKernelGGL((ref_scan_packed_kernel<0, 0>), dim3(grid), dim3(TPB), 0, c->stream, ref2, refbad, at, w0, nw, (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT), view(c, MG_BF_CTX));
    }
    hipLaunchKernelGGL(ref_scan_kernel, dim3(nblocks(nw)), dim3(TPB), 0, c->stream, d_ascii, w0, nw, (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT), view(c, MG_BF_CTX),
                       packed ? refbad : (const u32 *)nullptr, at);
    HIP_TRY(c, hipGetLastError());
    return MG_OK;
}
int ref_scan_checks(mg_ctx *c, size_t len)
{
    c->bf[MG_BF_CTX].pos_set_valid = false;
    if (!c->bf[MG_BF_ALT].mode) return fail(c, MG_ERR_STATE, "mg_ref_scan needs `bf` finalised (main.cpp:378 precedes :383)");
    if (c->bf[MG_BF_CTX].mode) return fail(c, MG_ERR_STATE, "context filter already finalised");
    const size_t off = (c->ref_k - c->k) / 2;
    if (off > len) return fail(c, MG_ERR_ARG, "contig shorter than (ref_k-k)/2: the reference throws std::out_of_range here");
    return MG_OK;
}
// main.cpp:386-389 for a contig shorter than ref_k, both strings clipped by std::string(reference, pos, n): one test, no loop
int ref_scan_short(mg_ctx *c, const char *contig, size_t len)
{
    const size_t off = (c->ref_k - c->k) / 2;
    const size_t kn = len - off < c->k ? len - off : c->k;
    if (kn == 0) return fail(c, MG_ERR_ARG, "contig of %zu bases has no centre k-mer", len);
    std::vector<char> r1(MG_MAX_KMER + 8, 0), r2(MG_MAX_KMER + 8, 0);
    memcpy(r1.data(), contig + off, kn);
    memcpy(r2.data(), contig, len);
    uint8_t hit = 0;
    TRY(mg_bf_test(c, MG_BF_ALT, r1.data(), r1.size(), 1, &hit));
    if (hit) TRY(mg_bf_insert(c, MG_BF_CTX, r2.data(), r2.size(), 1));
    return MG_OK;
}
} // namespace

MG_EXPORT int mg_ref_scan(mg_ctx *c, const char *contig, size_t len)
{
    const DeviceGuard on_device(c);
    if (!c) return MG_ERR_ARG;
    if (len && !contig) return fail(c, MG_ERR_ARG, "contig is NULL");
    TRY(ref_scan_checks(c, len));
    if (len < c->ref_k) return ref_scan_short(c, contig, len);
    const u64 n_windows = len - c->ref_k + 1;
    // stream the contig through the device in slices (a human chromosome is a few hundred MB): bytes up, packed, scanned
    const size_t slice = 256u << 20;
    for (u64 w0 = 0; w0 < n_windows; w0 += slice) {
        const u64 nw = n_windows - w0 < slice ? n_windows - w0 : slice;
        const size_t nbytes = nw + c->ref_k - 1;
        void *d, *p2, *pb;
        TRY(scratch(c, c->s_rows, (nbytes + 63) / 64 * 64 + 256, &d));
        TRY(scratch(c, c->s_aux, ((nbytes + 31) / 32 + 4) * 8, &p2));
        TRY(scratch(c, c->s_out, ((nbytes + 31) / 32 + 4) * 4, &pb));
        HIP_TRY(c, hipMemsetAsync((u8 *)d + nbytes / 64 * 64, 0, (nbytes + 63) / 64 * 64 + 256 - nbytes / 64 * 64, c->stream)); // (the tail the pack kernel reads past the bytes)
        HIP_TRY(c, hipMemcpyAsync(d, contig + w0, nbytes, hipMemcpyHostToDevice, c->stream));
        TRY(reference_pack(c, (const u8 *)d, nbytes, (u64 *)p2, (u32 *)pb));
        TRY(ref_scan_windows(c, (const u8 *)d, (const u64 *)p2, (const u32 *)pb, 0, w0, nw));
        HIP_TRY(c, hipStreamSynchronize(c->stream)); // (the scratch is reused by the next slice; the caller may reuse its buffer)
    }
    return MG_OK;
}

// the same for a contig that lies at [offset, offset + len) of the buffer given to mg_reference_upload: nothing crosses PCIe
MG_EXPORT int mg_ref_scan_resident(mg_ctx *c, uint64_t offset, size_t len)
{
    const DeviceGuard on_device(c);
    if (!c) return MG_ERR_ARG;
    if (!c->d_ref) return fail(c, MG_ERR_STATE, "mg_reference_upload first");
    if (offset + len > c->ref_len) return fail(c, MG_ERR_ARG, "contig lies outside the uploaded reference");
    TRY(ref_scan_checks(c, len));
    if (len < c->ref_k) { // (the short-contig case goes through the row kernels: fetch its bytes)
        std::vector<char> h(len + 1, 0);
        if (len) HIP_TRY(c, hipMemcpy(h.data(), c->d_ref + offset, len, hipMemcpyDeviceToHost));
        return ref_scan_short(c, h.data(), len);
    }
    const u64 n_windows = len - c->ref_k + 1;
    TRY(ref_scan_windows(c, c->d_ref + offset, c->d_ref2, c->d_refbad, offset, 0, n_windows));
    return MG_OK;
}

// ---- KMC scan ----------------------------------------------------------------------------

namespace {
// `context_bf` for the hit kernel: its set positions as a hash set, (re)built at the first scan after the bits changed
// Before a scan: bring the records' counter copies up to date (one pass over the record table, once after anything
// other than a scan has changed a counter: an index load, an import, a per-k-mer increment).  A context that is part of
// a multi-GPU group, or whose counter vector has been handed out, keeps its counters in the vectors alone.
int records_current(mg_ctx *c)
{
    if (!records_wanted(c)) {
        vectors_current(c);
        c->rec_ok = false;
        return MG_OK;
    }
    if (c->rec_ok || !c->map.slots) return MG_OK;
    if (++c->rec_epoch == 0) c->rec_epoch = 1; // (0 means "no copies"; the pass below rewrites every record that holds anything, so no older epoch survives it)
    MapView m = view(c);
    hipLaunchKernelGGL(rec_publish_kernel, dim3((unsigned)std::min<u64>(nblocks(1ULL << m.cap_log2), 1u << 20)), dim3(TPB), 0, c->stream, m,
                       (const u32 *)(c->bf[MG_BF_ALT].mode ? c->bf[MG_BF_ALT].counts : nullptr), c->rec_epoch);
    HIP_TRY(c, hipGetLastError());
    c->rec_ok = true;
    return MG_OK;
}
int ctx_set_ready(mg_ctx *c)
{
    BFState &b = c->bf[MG_BF_CTX];
    if (b.pos_set_valid) return MG_OK;
    // worth it where the bit array is far larger than the set: filters of 2^31 bits and more (256 MB) holding few bits
    u32 log2 = 10;
    while ((1ULL << log2) < 2 * b.nset + 16) ++log2;
    const bool want = c->use_ctx_set == 2 || (c->use_ctx_set == 1 && b.size >= (1ULL << 31) && (8ULL << log2) * 4 <= b.size / 8);
    if (!want) return MG_OK;
    if (b.pos_set) hipFree(b.pos_set);
    b.pos_set = nullptr;
    HIP_TRY(c, hipMalloc(&b.pos_set, 8ULL << log2));
    HIP_TRY(c, hipMemsetAsync(b.pos_set, 0, 8ULL << log2, c->stream));
    hipLaunchKernelGGL(pos_set_build_kernel, dim3(nblocks(b.nwords)), dim3(TPB), 0, c->stream, (const u64 *)b.words, b.nwords, b.pos_set, log2);
    HIP_TRY(c, hipGetLastError());
    b.pos_set_log2 = log2;
    b.pos_set_valid = true;
    return MG_OK;
}
// ---- the ticket form's host side ---------------------------------------------------------------------------------------
u64 ticket_slices(mg_ctx *c) // slices of half the L2-resident size (2 MiB) the fine gate splits into
{
    const u32 word_shift = (u32)(c->pregate_log2 - 1 - 6);
    return (((c->bf[MG_BF_ALT].n_gate_bits + 63) / 64) + (1ULL << word_shift) - 1) >> word_shift;
}
// does this index take the ticket form?  *row_bits = bits of a ticket left for the row number (rows per launch group = 2^row_bits)
bool ticket_form(mg_ctx *c, u32 *row_bits)
{
    const BFState &alt = c->bf[MG_BF_ALT];
    const u64 TP = ticket_slices(c);
    u32 idx_bits = 1;
    while (idx_bits < 64 && (alt.size - 1) >> idx_bits) ++idx_bits;
    *row_bits = std::min<u32>(27, 64 - idx_bits);
    return c->use_summary && c->use_tickets && alt.gate && c->gate_log2 >= c->ticket_min_log2 && TP >= 2 && TP <= (u64)TK_MAXP && idx_bits <= 44;
}
// segments, staging and meta block for launch groups of up to `cap` rows
int ticket_layout(mg_ctx *c, u64 cap, u32 row_bits, TicketSet *out)
{
    TicketSet tks{};
    const u64 TP = ticket_slices(c);
    tks.nbins = (u32)TP;
    tks.word_shift = (u32)(c->pregate_log2 - 1 - 6);
    tks.row_bits = row_bits;
    tks.nseg = (u32)std::min<u64>((cap + 4 * TPB - 1) / (4 * TPB), BIN_SEGS);
    const u64 wg_rows = ((cap + TK_TILE - 1) / TK_TILE + tks.nseg - 1) / tks.nseg * TK_TILE; // a workgroup takes whole tiles
    tks.segcap = c->bin_cap ? c->bin_cap : ((wg_rows / TP) * 3 / 2 + 64 + 15) / 16 * 16; // 1.5x an even share, whole 128-byte lines
    void *q[2];
    TRY(scratch(c, c->s_tk[0], TP * tks.nseg * tks.segcap * 8, &q[0]));
    TRY(scratch(c, c->s_tk[1], cap * 8, &q[1]));
    tks.tickets = (u64 *)q[0];
    tks.spill = (u64 *)q[1];
    if (!c->d_tk_meta) HIP_TRY(c, hipMalloc(&c->d_tk_meta, TK_META_HEAD + (size_t)TK_MAXP * BIN_SEGS * 4));
    tks.spill_count = c->d_tk_meta;
    tks.ablate = (u32)c->scan_ablate >> 8; // (timing-only diagnostics of pass one: scan_ablate 256, 512)
    tks.counts = (u32 *)((char *)c->d_tk_meta + TK_META_HEAD);
    *out = tks;
    return MG_OK;
}
// ---- the sub-slice form's host side ---------------------------------------------------------------------------------------
constexpr int SCAN_EV_CHUNKS = 64; // launch groups of one scan whose kernels are timed (mg_scan_stats)
int device_cus(mg_ctx *c)
{
    if (!c->n_cus) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, c->device) != hipSuccess || cus <= 0) cus = 256;
        c->n_cus = cus;
    }
    return c->n_cus;
}
u64 sub_bins(const mg_ctx *c) // LDS-sized pieces (2^sub_words_log2 words) the fine gate splits into
{
    const u64 nwords = (c->bf[MG_BF_ALT].n_gate_bits + 63) / 64;
    return (nwords + (1ULL << c->sub_words_log2) - 1) >> c->sub_words_log2;
}
// does this index take the sub-slice form?  *row_bits as in ticket_form
bool sub_form(mg_ctx *c, u32 *row_bits)
{
    const BFState &alt = c->bf[MG_BF_ALT];
    u32 idx_bits = 1;
    while (idx_bits < 64 && (alt.size - 1) >> idx_bits) ++idx_bits;
    *row_bits = std::min<u32>(27, 64 - idx_bits);
    const u64 NB = sub_bins(c);
    return c->use_summary && c->use_sub && alt.gate && c->gate_log2 >= c->sub_min_log2 && NB >= 2 && NB <= (u64)SB_MAXB && idx_bits <= 44 &&
           c->sub_words_log2 >= 0 && c->sub_words_log2 <= SB_WORDS_LOG2;
}
// what one launch group of the sub-slice form works in: pass one / two's segments and regions, the probe kernel's hit regions
struct SubPlan {
    SubSet ss{};
    u32 split = 1;
};
// segments, regions and meta block for launch groups of up to `cap` rows
int sub_layout(mg_ctx *c, u64 cap, u32 row_bits, SubPlan *plan)
{
    SubSet *out = &plan->ss;
    const BFState &alt = c->bf[MG_BF_ALT];
    SubSet ss{};
    const u64 NB = sub_bins(c);
    const int cus = device_cus(c);
    ss.nbins = (u32)NB;
    ss.words_log2 = (u32)c->sub_words_log2;
    ss.bin_shift = alt.gate_shift + 6 + ss.words_log2;
    ss.row_bits = row_bits;
    ss.n_gate_words = (alt.n_gate_bits + 63) / 64;
    const u64 tiles = (cap + SB_TILE - 1) / SB_TILE;
    ss.nseg = (u32)std::min<u64>(tiles, (u64)(c->sub_grid > 0 ? c->sub_grid : cus));
    const u64 wg_rows = (tiles + ss.nseg - 1) / ss.nseg * SB_TILE; // a workgroup takes whole tiles
    ss.segcap = c->bin_cap ? c->bin_cap : ((wg_rows / NB) * 3 / 2 + 64 + 15) / 16 * 16; // 1.5x an even share, whole 128-byte lines
    ss.parts = (u32)std::max<u64>(1, std::min<u64>(std::min<u64>(8, ss.nseg), (u64)cus / NB)); // bins fewer than CUs: several workgroups share one
    const u64 spp = (ss.nseg + ss.parts - 1) / ss.parts, units = NB * ss.parts;
    ss.ucap = spp * ss.segcap;
    plan->split = (u32)std::max(1, c->sub_split);
    void *q[4];
    TRY(scratch(c, c->s_sb[0], NB * ss.nseg * ss.segcap * 8, &q[0]));
    TRY(scratch(c, c->s_sb[1], cap * 8, &q[1]));
    TRY(scratch(c, c->s_sb[2], (units * ss.ucap + cap) * 8, &q[2]));
    const size_t head = (8 + (units + 1) * 4 + 15) / 16 * 16;
    TRY(scratch(c, c->s_sb[3], head + NB * ss.nseg * 4, &q[3]));
    ss.tickets = (u64 *)q[0];
    ss.spill = (u64 *)q[1];
    ss.out_tk = (u64 *)q[2];
    ss.spill_count = (unsigned long long *)q[3];
    ss.out_counts = (u32 *)((char *)q[3] + 8);
    ss.counts = (u32 *)((char *)q[3] + head);
    ss.ablate = (u32)c->scan_ablate >> 8; // (timing-only diagnostic of pass one: scan_ablate 256)
    *out = ss;
    return MG_OK;
}
// bytes of the meta block's head that every launch group starts from zero: the spill count and the regions' counts
size_t sub_meta_head(const SubSet &ss) { return 8 + ((size_t)ss.nbins * ss.parts + 1) * 4; }
// the four events of launch group `chunk` of the running scan (created on first use); nullptr beyond SCAN_EV_CHUNKS
hipEvent_t *scan_events(mg_ctx *c, u64 chunk, u64 rows)
{
    if (chunk >= (u64)SCAN_EV_CHUNKS) return nullptr;
    while (c->ev.size() < 4 * (chunk + 1)) {
        hipEvent_t e = nullptr;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        c->ev.push_back(e);
    }
    if (c->ev_rows.size() <= chunk) c->ev_rows.resize(chunk + 1);
    c->ev_rows[chunk] = rows;
    return c->ev.data() + 4 * chunk;
}
template <int KC, int RC>
void launch_sub_passes(mg_ctx *c, const u64 *d_hi, const u64 *d_lo, const u32 *rows12, u64 n, const SubSet &layout)
{
    SubSet ss = layout;
    ss.nseg = (u32)std::min<u64>((n + SB_TILE - 1) / SB_TILE, layout.nseg); // (regions and segments stay as laid out: a smaller grid fills fewer of them)
    const BFView alt = view(c, MG_BF_ALT);
if (rows12) hipLaunchKernelGGL((scan_sub_sort_kernel<KC, RC, true>), dim3(ss.nseg), dim3(SB_TPB), 0, c->stream, d_hi, d_lo, rows12, n, (int)c->k, (int)c->ref_k, alt, ss);
    else hipLaunchKernelGGL((scan_sub_sort_kernel<KC, RC, false>), dim3(ss.nseg), dim3(SB_TPB), 0, c->stream, d_hi, d_lo, rows12, n, (int)c->k, (int)c->ref_k, alt, ss);
    const unsigned units = ss.nbins * ss.parts;
    const unsigned grid = std::min<unsigned>(units, (unsigned)device_cus(c));
    if (c->gate_k == 4) hipLaunchKernelGGL(scan_sub_gate_kernel<4>, dim3(grid), dim3(SB_TPB), 0, c->stream, alt, ss);
    else hipLaunchKernelGGL(scan_sub_gate_kernel<0>, dim3(grid), dim3(SB_TPB), 0, c->stream, alt, ss);
    hipLaunchKernelGGL(sub_total_kernel, dim3(1), dim3(SB_TPB), 0, c->stream, (const u32 *)ss.out_counts, units + 1, c->d_hit_count);
}
SubOpen sub_open(const SubPlan *plan)
{
    SubOpen r{};
    const SubSet *ss = &plan->ss;
    r.counts = ss->out_counts;
    r.tickets = ss->out_tk;
    r.ucap = ss->ucap;
    r.n_units = ss->nbins * ss->parts + 1;
    r.split = plan->split;
    r.row_bits = ss->row_bits;
    return r;
}
// the probe kernel of the sub-slice form (which is its hit pass too) over one launch group (ev: its four events, or NULL)
template <int KC, int RC>
void launch_sub_tail(mg_ctx *c, const SubPlan *plan, hipEvent_t *ev, const u32 *d_cnt, const u64 *d_hi, const u64 *d_lo, const u32 *rows12)
{
    const SubOpen so = sub_open(plan);
    const unsigned pgrid = std::min<unsigned>(so.n_units * so.split, 2 * (unsigned)c->probe_grid);
    hipLaunchKernelGGL((scan_sub_probe_kernel<KC, RC>), dim3(pgrid), dim3(TPB), 0, c->stream, (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT), view(c, MG_BF_CTX), view(c), so,
                       c->d_hit_count, d_cnt, d_hi, d_lo, rows12);
    if (ev) hipEventRecord(ev[2], c->stream);
    if (ev) hipEventRecord(ev[3], c->stream);
}
template <int KC, int RC, int ROWS, int VAR>
void launch_filter_var(mg_ctx *c, const u64 *d_hi, const u64 *d_lo, const u32 *d_cnt, u64 n, RowList open)
{
    const u64 per_block = (u64)TPB * ROWS;
    const unsigned grid = (unsigned)std::min<u64>((n + per_block - 1) / per_block, (u64)c->scan_grid);
    hipLaunchKernelGGL((scan_filter_kernel<KC, RC, ROWS, VAR>), dim3(grid), dim3(TPB), 0, c->stream, d_hi, d_lo, d_cnt, n, (int)c->k,
                       (int)c->ref_k, view(c, MG_BF_ALT), open, c->d_hit_count, c->scan_ablate);
}
template <int KC, int RC, int ROWS>
void launch_filter_rows(mg_ctx *c, const u64 *d_hi, const u64 *d_lo, const u32 *d_cnt, u64 n, RowList open)
{
    // 16-byte loads need 16-byte aligned table bases (chunk offsets are multiples of 2^27 rows, so only the caller's bases matter)
    const bool vec_ok = ROWS == 2 && ((((uintptr_t)d_hi | (uintptr_t)d_lo) & 15) == 0) && (((uintptr_t)d_cnt & 7) == 0);
    switch ((c->scan_variant & 2) && vec_ok ? 2 : 0) {
    case 2: launch_filter_var<KC, RC, ROWS, 2>(c, d_hi, d_lo, d_cnt, n, open); break;
    default: launch_filter_var<KC, RC, ROWS, 0>(c, d_hi, d_lo, d_cnt, n, open); break;
    }
}
// The two passes of the ticket form over one chunk of the table (SoA arrays, or compact rows when `rows12` is given).
template <int KC, int RC>
void launch_ticket_passes(mg_ctx *c, const u64 *d_hi, const u64 *d_lo, const u32 *rows12, u64 n, const TicketSet &layout, RowList open)
{
    TicketSet ts = layout;
    ts.nseg = (u32)std::min<u64>((n + 4 * TPB - 1) / (4 * TPB), layout.nseg);
    hipLaunchKernelGGL((scan_ticket_sort_kernel<KC, RC>), dim3(ts.nseg), dim3(TPB), 0, c->stream, d_hi, d_lo, rows12, n, (int)c->k, (int)c->ref_k,
                       view(c, MG_BF_ALT), ts);
    if (!c->tkg_grid) { // pass two walks the slices in step: one workgroup per CU, all resident together
        int cus = 0, dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        c->tkg_grid = std::max(8, cus / 8 * 8);
    }
    if (c->gate_k == 4)
        hipLaunchKernelGGL(scan_ticket_gate_kernel<4>, dim3(c->tkg_grid), dim3(TKG_TPB), 0, c->stream, view(c, MG_BF_ALT), ts, open.cnt, c->d_hit_count);
    else
        hipLaunchKernelGGL(scan_ticket_gate_kernel<0>, dim3(c->tkg_grid), dim3(TKG_TPB), 0, c->stream, view(c, MG_BF_ALT), ts, open.cnt, c->d_hit_count);
}
template <int KC, int RC>
void launch_scan_chunk(mg_ctx *c, const u64 *d_hi, const u64 *d_lo, const u32 *d_cnt, u64 n, RowList open, RowList hits, hipEvent_t *ev,
                       const BinSet *bins, const TicketSet *tickets, const SubPlan *subs)
{
    if (ev) hipEventRecord(ev[0], c->stream);
    if (subs) { // whole-genome index: tickets filed by LDS-sized sub-slice of the gate, then each sub-slice answered out of LDS
        launch_sub_passes<KC, RC>(c, d_hi, d_lo, nullptr, n, subs->ss);
        if (ev) hipEventRecord(ev[1], c->stream);
        launch_sub_tail<KC, RC>(c, subs, ev, d_cnt, d_hi, d_lo, nullptr); // regions of surviving tickets: the record first, the table row where the record asks for it
        return;
    } else if (tickets) { // the same with 2 MiB slices walked out of L2
        launch_ticket_passes<KC, RC>(c, d_hi, d_lo, nullptr, n, *tickets, open);
    } else if (bins) { // large index: coarse gate + binning, then the fine gate slice by slice
        BinSet bs = *bins; // the last chunk may need fewer workgroups than segments were laid out for
        bs.nseg = (u32)std::min<u64>((n + 2 * TPB - 1) / (2 * TPB), bins->nseg);
        bins = &bs;
        if (c->bin_rows == 4)
            hipLaunchKernelGGL((scan_bin_kernel<KC, RC, 4>), dim3(bs.nseg), dim3(TPB), (size_t)bs.nbins * bs.ring * 20, c->stream, d_hi, d_lo, d_cnt, n,
                               (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT), *bins, c->scan_ablate);
        else
            hipLaunchKernelGGL((scan_bin_kernel<KC, RC, 2>), dim3(bs.nseg), dim3(TPB), (size_t)bs.nbins * bs.ring * 20, c->stream, d_hi, d_lo, d_cnt, n,
                               (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT), *bins, c->scan_ablate);
        hipLaunchKernelGGL((scan_bin_gate_kernel<KC, RC>), dim3(2048), dim3(TPB), 0, c->stream, (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT),
                           *bins, open, c->d_hit_count);
    } else
        switch (c->scan_rows) {
        case 1: launch_filter_rows<KC, RC, 1>(c, d_hi, d_lo, d_cnt, n, open); break;
        case 4: launch_filter_rows<KC, RC, 4>(c, d_hi, d_lo, d_cnt, n, open); break;
        default: launch_filter_rows<KC, RC, 2>(c, d_hi, d_lo, d_cnt, n, open); break;
        }
    if (ev) hipEventRecord(ev[1], c->stream);
    // the list lengths live on the device; fixed grids walk them with a stride, so no host round trip
    const unsigned grid = (unsigned)std::min<u64>(nblocks(n), (u64)c->probe_grid);
    hipLaunchKernelGGL((scan_probe_kernel<KC, RC>), dim3(grid), dim3(TPB), 0, c->stream, (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT), view(c), open, hits, c->d_hit_count,
                       bins && !tickets ? (const u32 *)nullptr : d_cnt, tickets ? d_hi : (const u64 *)nullptr, tickets ? d_lo : (const u64 *)nullptr, (const u32 *)nullptr);
    if (ev) hipEventRecord(ev[2], c->stream);
    hipLaunchKernelGGL((scan_hits_kernel<KC, RC>), dim3(std::min(grid, (unsigned)c->hits_grid)), dim3(TPB), 0, c->stream, (int)c->k, (int)c->ref_k,
                       view(c, MG_BF_ALT), view(c, MG_BF_CTX), view(c), hits, c->d_hit_count);
    if (ev) hipEventRecord(ev[3], c->stream);
}
} // namespace

MG_EXPORT int mg_kmc_scan_device(mg_ctx *c, const void *d_hi, const void *d_lo, const void *d_cnt, size_t n)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    if (!c->bf[0].mode || !c->bf[1].mode) return fail(c, MG_ERR_STATE, "mg_kmc_scan needs both filters finalised");
    if (c->ref_k > MG_MAX_PACKED_K)
        return fail(c, MG_ERR_LIMIT, "packed scan supports k <= ref_k <= 64 (k=%u ref_k=%u)", c->k, c->ref_k);
    if (n == 0) return MG_OK;
    if (!d_hi || !d_lo || !d_cnt) return fail(c, MG_ERR_ARG, "NULL table pointer");
    if (!c->map.slots) TRY(map_reserve(c, 0));
    TRY(ctx_set_ready(c));
    TRY(records_current(c));
    // large index: tickets by gate slice (takes precedence over the row-moving partition below)
    const BFState &alt = c->bf[MG_BF_ALT];
    const u32 word_shift = (u32)(c->pregate_log2 - 1 - 6); // slices of half the L2-resident size: 2 MiB
    u32 row_bits = 27;
    const bool subs = sub_form(c, &row_bits);
    const bool tickets = !subs && ticket_form(c, &row_bits);
    const u64 TP = ticket_slices(c);
    const u64 chunk = 1ULL << std::min<u32>(tickets || subs ? row_bits : 27, (u32)c->chunk_log2); // rows per launch group (bounds the two lists' worst-case size; a ticket holds the row number)
    const u64 cap = n < chunk ? n : chunk; // worst case (gate disabled): every row is listed
    void *p[6];
    Scratch *sc[6] = {&c->s_open[0], &c->s_open[1], &c->s_open[2], &c->s_hit[0], &c->s_hit[1], &c->s_hit[2]};
    for (int i = 0; i < 6; ++i) TRY(scratch(c, *sc[i], cap * (i % 3 == 2 ? 4 : 8), &p[i]));
    RowList open{(u64 *)p[0], (u64 *)p[1], (u32 *)p[2]}, hits{(u64 *)p[3], (u64 *)p[4], (u32 *)p[5]};
    if (c->use_hit_entries && c->map.cap_log2 <= 30) { // (record * 2 + entry in 32 bits)
        void *pa;
        TRY(scratch(c, c->s_hit[3], cap * 8, &pa));
        hits.aux = (u64 *)pa;
    }
    c->stats_valid = false;
    // partitioned second level: two-level gate in use and the fine gate splits into 2..BIN_MAXP slices of half the coarse gate's size
    BinSet bins{};
    TicketSet tks{};
    SubPlan sbs{};
    if (tickets) TRY(ticket_layout(c, cap, row_bits, &tks));
    if (subs) TRY(sub_layout(c, cap, row_bits, &sbs));
    const u64 P = !tickets && !subs && alt.pregate && pregate_on(c) ? (((alt.n_gate_bits + 63) / 64 + (1ULL << word_shift) - 1) >> word_shift) : 0;
    const bool partition = c->use_summary && c->use_pregate && c->use_partition && P >= 2 && P <= BIN_MAXP;
    if (partition) {
        bins.nbins = (u32)P;
        bins.word_shift = word_shift;
        bins.nseg = (u32)std::min<u64>((cap + 2 * TPB - 1) / (2 * TPB), BIN_SEGS);
        // 1.5x an even share of the worst case (every row passes the coarse gate)
        bins.segcap = c->bin_cap ? c->bin_cap : ((cap / P / bins.nseg) * 3 / 2 + 256 + 63) / 64 * 64; // line-aligned segments
        bins.ring = 64;
        while (bins.ring < 256 && bins.ring * 2 * P <= BIN_LDS_ROWS) bins.ring *= 2;
        if (c->bin_ring && (u64)c->bin_ring * P <= BIN_LDS_ROWS) bins.ring = (u32)c->bin_ring;
        void *q[6];
        for (int i = 0; i < 3; ++i) TRY(scratch(c, c->s_bin[i], P * bins.nseg * bins.segcap * (i == 2 ? 4 : 8), &q[i]));
        for (int i = 0; i < 3; ++i) TRY(scratch(c, c->s_spill[i], cap * (i == 2 ? 4 : 8), &q[3 + i]));
        bins.rows = RowList{(u64 *)q[0], (u64 *)q[1], (u32 *)q[2]};
        bins.spill = RowList{(u64 *)q[3], (u64 *)q[4], (u32 *)q[5]};
        if (!c->d_bin_meta) HIP_TRY(c, hipMalloc(&c->d_bin_meta, 8 + (size_t)BIN_MAXP * BIN_SEGS * 4));
        bins.spill_count = c->d_bin_meta;
        bins.counts = (u32 *)(c->d_bin_meta + 1);
    }
    HIP_TRY(c, hipMemsetAsync(c->d_hit_count, 0, 32, c->stream));
    for (u64 r0 = 0; r0 < n; r0 += chunk) {
        const u64 nr = n - r0 < chunk ? n - r0 : chunk;
        const u64 *ph = (const u64 *)d_hi + r0, *pl = (const u64 *)d_lo + r0;
        const u32 *pc = (const u32 *)d_cnt + r0;
        if (r0) HIP_TRY(c, hipMemsetAsync(c->d_hit_count, 0, 16, c->stream));
        if (partition) HIP_TRY(c, hipMemsetAsync(c->d_bin_meta, 0, 8, c->stream));
        if (tickets) HIP_TRY(c, hipMemsetAsync(c->d_tk_meta, 0, TK_META_HEAD, c->stream));
        if (subs) HIP_TRY(c, hipMemsetAsync(sbs.ss.spill_count, 0, sub_meta_head(sbs.ss), c->stream));
        hipEvent_t *ev = scan_events(c, r0 / chunk, nr);
        // the reference's defaults (k35 r43, argument_parser.hpp:57-58) and config C5 (k35 r63) get fixed-length hashing
        if (c->k == 35 && c->ref_k == 43) launch_scan_chunk<35, 43>(c, ph, pl, pc, nr, open, hits, ev, partition ? &bins : nullptr, tickets ? &tks : nullptr, subs ? &sbs : nullptr);
        else if (c->k == 35 && c->ref_k == 63) launch_scan_chunk<35, 63>(c, ph, pl, pc, nr, open, hits, ev, partition ? &bins : nullptr, tickets ? &tks : nullptr, subs ? &sbs : nullptr);
        else launch_scan_chunk<0, 0>(c, ph, pl, pc, nr, open, hits, ev, partition ? &bins : nullptr, tickets ? &tks : nullptr, subs ? &sbs : nullptr);
        HIP_TRY(c, hipGetLastError());
    }
    c->ev_rows.resize(std::min<u64>((n + chunk - 1) / chunk, (u64)SCAN_EV_CHUNKS));
    c->last_bins = partition ? (int)P : 0;
    c->last_tickets = tickets ? (int)TP : 0;
    c->last_subs = subs ? (int)sbs.ss.nbins : 0;
    if (subs && view(c).lazy) c->vec_stale = true; // (the records' copies are ahead of the vectors now)
    else c->vec_zero = false;
    c->stats_valid = true;
    return MG_OK;
}

// ---- compact (12-byte) table rows --------------------------------------------------------------------------------
namespace {
template <int KC, int RC>
void launch_rows12_chunk(mg_ctx *c, const uint4 *rows, u64 n, RowList open, RowList hits, hipEvent_t *ev, const TicketSet *tickets, const SubPlan *subs)
{
    if (ev) hipEventRecord(ev[0], c->stream);
    if (subs) { // whole-genome index: tickets filed by LDS-sized sub-slice of the gate, each sub-slice then answered out of LDS; regions of tickets
        launch_sub_passes<KC, RC>(c, nullptr, nullptr, (const u32 *)rows, n, subs->ss);
        if (ev) hipEventRecord(ev[1], c->stream);
        launch_sub_tail<KC, RC>(c, subs, ev, nullptr, nullptr, nullptr, (const u32 *)rows);
        return;
    } else if (tickets) // the same with 2 MiB slices walked out of L2; the open list holds row numbers
        launch_ticket_passes<KC, RC>(c, nullptr, nullptr, (const u32 *)rows, n, *tickets, open);
    else {
        const unsigned fgrid = (unsigned)std::min<u64>(((n + 1) / 2 + TPB - 1) / TPB, (u64)c->scan_grid);
        hipLaunchKernelGGL((scan_filter12_kernel<KC, RC>), dim3(fgrid), dim3(TPB), 0, c->stream, (const u32 *)rows, n, (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT),
                           open, c->d_hit_count, c->scan_ablate);
    }
    if (ev) hipEventRecord(ev[1], c->stream);
    const unsigned grid = (unsigned)std::min<u64>(nblocks(n), (u64)c->probe_grid);
    hipLaunchKernelGGL((scan_probe_kernel<KC, RC>), dim3(grid), dim3(TPB), 0, c->stream, (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT), view(c), open, hits, c->d_hit_count,
                       (const u32 *)nullptr, (const u64 *)nullptr, (const u64 *)nullptr, tickets ? (const u32 *)rows : (const u32 *)nullptr);
    if (ev) hipEventRecord(ev[2], c->stream);
    hipLaunchKernelGGL((scan_hits_kernel<KC, RC>), dim3(std::min(grid, (unsigned)c->hits_grid)), dim3(TPB), 0, c->stream, (int)c->k, (int)c->ref_k,
                       view(c, MG_BF_ALT), view(c, MG_BF_CTX), view(c), hits, c->d_hit_count);
    if (ev) hipEventRecord(ev[3], c->stream);
}
int rows12_ok(mg_ctx *c)
{
    if (c->ref_k < 33 || c->ref_k > 44)
        return fail(c, MG_ERR_LIMIT, "packed 12-byte rows hold a ref_k-mer of 33..44 bases and a count of 96 - 2 ref_k bits (ref_k = %u): use the SoA table", c->ref_k);
    return MG_OK;
}
} // namespace

MG_EXPORT size_t mg_kmc_rows_bytes(size_t n) { return (n + 3) / 4 * 4 * 12; }

MG_EXPORT int mg_kmc_pack_rows_device(mg_ctx *c, const void *d_hi, const void *d_lo, const void *d_cnt, size_t n, void *d_rows_out)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    TRY(rows12_ok(c));
    if (n == 0) return MG_OK;
    if (!d_hi || !d_lo || !d_cnt || !d_rows_out) return fail(c, MG_ERR_ARG, "NULL pointer");
    int *d_bad = (int *)(c->d_hit_count + 3);
    HIP_TRY(c, hipMemsetAsync(d_bad, 0, 4, c->stream));
    hipLaunchKernelGGL(pack_rows12_kernel, dim3(nblocks((n + 3) / 4 * 4)), dim3(TPB), 0, c->stream, (const u64 *)d_hi, (const u64 *)d_lo, (const u32 *)d_cnt,
                       (u64)n, (int)c->ref_k, (u32 *)d_rows_out, d_bad);
    HIP_TRY(c, hipGetLastError());
    int bad = 0;
    HIP_TRY(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (bad) return fail(c, MG_ERR_LIMIT, "a count of 2^%u or more (or a k-mer wider than ref_k) does not fit a packed row: use the SoA table", 96 - 2 * c->ref_k);
    return MG_OK;
}

MG_EXPORT int mg_kmc_scan_rows_device(mg_ctx *c, const void *d_rows, size_t n)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    if (!c->bf[0].mode || !c->bf[1].mode) return fail(c, MG_ERR_STATE, "mg_kmc_scan needs both filters finalised");
    TRY(rows12_ok(c));
    if (n == 0) return MG_OK;
    if (!d_rows || ((uintptr_t)d_rows & 15)) return fail(c, MG_ERR_ARG, "packed rows must be 16-byte aligned");
    if (!c->map.slots) TRY(map_reserve(c, 0));
    TRY(ctx_set_ready(c));
    TRY(records_current(c));
    u32 row_bits = 27;
    const bool subs = sub_form(c, &row_bits);
    const bool tickets = !subs && ticket_form(c, &row_bits);
    const u64 chunk = std::max<u64>(4, 1ULL << std::min<u32>(tickets || subs ? row_bits : 27, (u32)c->chunk_log2)); // (a multiple of 4 rows: chunks start on whole quads)
    const u64 cap = n < chunk ? n : chunk;
    void *p[6];
    Scratch *sc[6] = {&c->s_open[0], &c->s_open[1], &c->s_open[2], &c->s_hit[0], &c->s_hit[1], &c->s_hit[2]};
    for (int i = 0; i < 6; ++i) TRY(scratch(c, *sc[i], cap * (i % 3 == 2 ? 4 : 8), &p[i]));
    RowList open{(u64 *)p[0], (u64 *)p[1], (u32 *)p[2]}, hits{(u64 *)p[3], (u64 *)p[4], (u32 *)p[5]};
    if (c->use_hit_entries && c->map.cap_log2 <= 30) { // (record * 2 + entry in 32 bits)
        void *pa;
        TRY(scratch(c, c->s_hit[3], cap * 8, &pa));
        hits.aux = (u64 *)pa;
    }
    TicketSet tks{};
    SubPlan sbs{};
    if (tickets) TRY(ticket_layout(c, cap, row_bits, &tks));
    if (subs) TRY(sub_layout(c, cap, row_bits, &sbs));
    c->stats_valid = false;
    HIP_TRY(c, hipMemsetAsync(c->d_hit_count, 0, 32, c->stream));
    for (u64 r0 = 0; r0 < n; r0 += chunk) {
        const u64 nr = n - r0 < chunk ? n - r0 : chunk;
        const uint4 *pr = (const uint4 *)d_rows + r0 / 4 * 3;
        if (r0) HIP_TRY(c, hipMemsetAsync(c->d_hit_count, 0, 16, c->stream));
        if (tickets) HIP_TRY(c, hipMemsetAsync(c->d_tk_meta, 0, TK_META_HEAD, c->stream));
        if (subs) HIP_TRY(c, hipMemsetAsync(sbs.ss.spill_count, 0, sub_meta_head(sbs.ss), c->stream));
        hipEvent_t *ev = scan_events(c, r0 / chunk, nr);
        if (c->k == 35 && c->ref_k == 43) launch_rows12_chunk<35, 43>(c, pr, nr, open, hits, ev, tickets ? &tks : nullptr, subs ? &sbs : nullptr);
        else launch_rows12_chunk<0, 0>(c, pr, nr, open, hits, ev, tickets ? &tks : nullptr, subs ? &sbs : nullptr);
        HIP_TRY(c, hipGetLastError());
    }
    c->ev_rows.resize(std::min<u64>((n + chunk - 1) / chunk, (u64)SCAN_EV_CHUNKS));
    c->last_bins = 0;
    c->last_subs = subs ? (int)sbs.ss.nbins : 0;
    if (subs && view(c).lazy) c->vec_stale = true; // (the records' copies are ahead of the vectors now)
    else c->vec_zero = false;
    c->last_tickets = tickets ? (int)ticket_slices(c) : 0;
    c->stats_valid = true;
    return MG_OK;
}

namespace {
// Host-fed scans stream the table through the device in pieces.  Two staging slots: while the scan kernels work
// on one, the copy stream fills the other (PCIe and HBM work overlap; with pinned host memory -- mg_host_alloc --
// the copies are truly asynchronous, with pageable memory the runtime stages them and the host blocks per copy,
// which still leaves the previous piece's kernels running underneath).
int pipeline_ready(mg_ctx *c)
{
    if (c->copy_stream) return MG_OK;
    HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_up[i], hipEventDisableTiming));
        HIP_TRY(c, hipEventCreateWithFlags(&c->ev_free[i], hipEventDisableTiming));
    }
    return MG_OK;
}
constexpr size_t HOST_PIECE = 1u << 24; // rows per staged piece: 320 MB of SoA rows per slot
} // namespace

namespace {
// Host-fed scans leave DMA from the caller's buffers in flight on copy_stream: whatever way the call ends, both streams are
// drained before the caller (who may free or unmap the source on an error) sees the result.
struct StreamsDrained {
    mg_ctx *c;
    ~StreamsDrained()
    {
        if (c->copy_stream) hipStreamSynchronize(c->copy_stream);
        hipStreamSynchronize(c->stream);
    }
};
} // namespace

MG_EXPORT int mg_kmc_scan(mg_ctx *c, const uint64_t *hi, const uint64_t *lo, const uint32_t *cnt, size_t n)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    if (n == 0) return MG_OK;
    if (!hi || !lo || !cnt) return fail(c, MG_ERR_ARG, "NULL table pointer");
    if (!c->bf[0].mode || !c->bf[1].mode) return fail(c, MG_ERR_STATE, "mg_kmc_scan needs both filters finalised");
    TRY(pipeline_ready(c));
    const StreamsDrained drained{c};
    const size_t cap = n < HOST_PIECE ? n : HOST_PIECE;
    void *d[2][3];
    for (int sl = 0; sl < 2; ++sl)
        for (int i = 0; i < 3; ++i) TRY(scratch(c, c->s_stage[sl][i], cap * (i == 2 ? 4 : 8), &d[sl][i]));
    size_t piece_no = 0;
    for (size_t r0 = 0; r0 < n; r0 += HOST_PIECE, ++piece_no) {
        const size_t nr = n - r0 < HOST_PIECE ? n - r0 : HOST_PIECE;
        const int sl = (int)(piece_no & 1);
        if (piece_no >= 2) HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->ev_free[sl], 0)); // the scan that used this slot is done
        HIP_TRY(c, hipMemcpyAsync(d[sl][0], hi + r0, nr * 8, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipMemcpyAsync(d[sl][1], lo + r0, nr * 8, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipMemcpyAsync(d[sl][2], cnt + r0, nr * 4, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipEventRecord(c->ev_up[sl], c->copy_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_up[sl], 0));
        TRY(mg_kmc_scan_device(c, d[sl][0], d[sl][1], d[sl][2], nr));
        HIP_TRY(c, hipEventRecord(c->ev_free[sl], c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // the caller may reuse its buffers
    return MG_OK;
}

// ---- KMC database feed -------------------------------------------------------------------------

MG_EXPORT int mg_host_alloc(void **out, size_t bytes)
{
    if (!out) return MG_ERR_ARG;
    *out = nullptr;
    return hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault) == hipSuccess ? MG_OK : MG_ERR_NOMEM;
}
MG_EXPORT int mg_host_free(void *p)
{
    if (p && hipHostFree(p) != hipSuccess) return MG_ERR_HIP;
    return MG_OK;
}

MG_EXPORT int mg_kmc_set_lut(mg_ctx *c, const uint64_t *lut, size_t n_lut, uint32_t lut_prefix_len, uint32_t suffix_bytes,
                             uint32_t counter_bytes, uint32_t min_count, uint64_t max_count, uint64_t total_records)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    if (!lut || n_lut == 0) return fail(c, MG_ERR_ARG, "mg_kmc_set_lut: empty prefix table");
    if (lut_prefix_len == 0 || lut_prefix_len > 15) return fail(c, MG_ERR_ARG, "mg_kmc_set_lut: lut_prefix_len %u (1..15)", lut_prefix_len);
    if (counter_bytes < 1 || counter_bytes > 4) return fail(c, MG_ERR_LIMIT, "KMC counters of %u bytes (1..4 supported: the reference reads them into a uint32)", counter_bytes);
    if (suffix_bytes < 1 || suffix_bytes > 15 || lut_prefix_len + 4 * suffix_bytes != c->ref_k)
        return fail(c, MG_ERR_ARG, "database k = %u (prefix %u + %u suffix bytes), context expects -r %u", lut_prefix_len + 4 * suffix_bytes,
                    lut_prefix_len, suffix_bytes, c->ref_k);
    if (c->ref_k > MG_MAX_PACKED_K) return fail(c, MG_ERR_LIMIT, "packed scan supports ref_k <= 64");
    if (n_lut % (1ULL << (2 * lut_prefix_len)) != 0) return fail(c, MG_ERR_ARG, "prefix table of %zu entries is not a whole number of 4^%u-entry bins", n_lut, lut_prefix_len);
    if (lut[0] != 0) return fail(c, MG_ERR_ARG, "prefix table does not start at record 0");
    for (size_t j = 1; j < n_lut; ++j)
        if (lut[j] < lut[j - 1] || lut[j] > total_records) return fail(c, MG_ERR_ARG, "prefix table is not ascending within the %llu records", (unsigned long long)total_records);
    hipFree(c->d_kmc_lut);
    c->d_kmc_lut = nullptr;
    HIP_TRY(c, hipMalloc(&c->d_kmc_lut, (n_lut + 1) * 8));
    HIP_TRY(c, hipMemcpy(c->d_kmc_lut, lut, n_lut * 8, hipMemcpyHostToDevice));
    const u64 guard = ~0ULL; // above every record index
    HIP_TRY(c, hipMemcpy(c->d_kmc_lut + n_lut, &guard, 8, hipMemcpyHostToDevice));
    c->kmc_n_lut = n_lut;
    c->kmc_prefix_len = lut_prefix_len;
    c->kmc_suffix_bytes = suffix_bytes;
    c->kmc_counter_bytes = counter_bytes;
    c->kmc_min_count = min_count;
    c->kmc_max_count = max_count;
    return MG_OK;
}

MG_EXPORT int mg_kmc_scan_records(mg_ctx *c, const void *records, size_t n, uint64_t first_record)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    if (n == 0) return MG_OK;
    if (!records) return fail(c, MG_ERR_ARG, "NULL records");
    if (!c->d_kmc_lut) return fail(c, MG_ERR_STATE, "mg_kmc_set_lut first");
    if (!c->bf[0].mode || !c->bf[1].mode) return fail(c, MG_ERR_STATE, "mg_kmc_scan needs both filters finalised");
    TRY(pipeline_ready(c));
    const StreamsDrained drained{c};
    const size_t rs = c->kmc_suffix_bytes + c->kmc_counter_bytes;
    const size_t cap = n < HOST_PIECE ? n : HOST_PIECE;
    void *d[2][3], *raw[2];
    for (int sl = 0; sl < 2; ++sl) {
        for (int i = 0; i < 3; ++i) TRY(scratch(c, c->s_stage[sl][i], cap * (i == 2 ? 4 : 8), &d[sl][i]));
        TRY(scratch(c, c->s_raw[sl], cap * rs + 64, &raw[sl]));
    }
    size_t piece_no = 0;
    for (size_t r0 = 0; r0 < n; r0 += HOST_PIECE, ++piece_no) {
        const size_t nr = n - r0 < HOST_PIECE ? n - r0 : HOST_PIECE;
        const int sl = (int)(piece_no & 1);
        if (piece_no >= 2) HIP_TRY(c, hipStreamWaitEvent(c->copy_stream, c->ev_free[sl], 0));
        HIP_TRY(c, hipMemcpyAsync(raw[sl], (const char *)records + r0 * rs, nr * rs, hipMemcpyHostToDevice, c->copy_stream));
        HIP_TRY(c, hipEventRecord(c->ev_up[sl], c->copy_stream));
        HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_up[sl], 0));
        hipLaunchKernelGGL(kmc_decode_kernel, dim3((unsigned)((nr + KMC_TILE - 1) / KMC_TILE)), dim3(TPB), 0, c->stream, (const u8 *)raw[sl], (u64)nr,
                           (u64)(first_record + r0), c->kmc_suffix_bytes, c->kmc_counter_bytes, c->kmc_prefix_len, (const u64 *)c->d_kmc_lut, c->kmc_n_lut,
                           c->kmc_min_count, c->kmc_max_count, (u64 *)d[sl][0], (u64 *)d[sl][1], (u32 *)d[sl][2]);
        HIP_TRY(c, hipGetLastError());
        TRY(mg_kmc_scan_device(c, d[sl][0], d[sl][1], d[sl][2], nr));
        HIP_TRY(c, hipEventRecord(c->ev_free[sl], c->stream));
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}

// the decoded rows of a run of records (tests, and callers that want the table itself)
MG_EXPORT int mg_kmc_decode_records(mg_ctx *c, const void *records, size_t n, uint64_t first_record, uint64_t *hi_out, uint64_t *lo_out,
                                    uint32_t *cnt_out)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    if (n == 0) return MG_OK;
    if (!records || !hi_out || !lo_out || !cnt_out) return fail(c, MG_ERR_ARG, "NULL argument");
    if (!c->d_kmc_lut) return fail(c, MG_ERR_STATE, "mg_kmc_set_lut first");
    const size_t rs = c->kmc_suffix_bytes + c->kmc_counter_bytes;
    for (size_t r0 = 0; r0 < n; r0 += HOST_PIECE) {
        const size_t nr = n - r0 < HOST_PIECE ? n - r0 : HOST_PIECE;
        void *d[3], *raw;
        for (int i = 0; i < 3; ++i) TRY(scratch(c, c->s_stage[0][i], nr * (i == 2 ? 4 : 8), &d[i]));
        TRY(scratch(c, c->s_raw[0], nr * rs + 64, &raw));
        HIP_TRY(c, hipMemcpyAsync(raw, (const char *)records + r0 * rs, nr * rs, hipMemcpyHostToDevice, c->stream));
        hipLaunchKernelGGL(kmc_decode_kernel, dim3((unsigned)((nr + KMC_TILE - 1) / KMC_TILE)), dim3(TPB), 0, c->stream, (const u8 *)raw, (u64)nr,
                           (u64)(first_record + r0), c->kmc_suffix_bytes, c->kmc_counter_bytes, c->kmc_prefix_len, (const u64 *)c->d_kmc_lut, c->kmc_n_lut,
                           c->kmc_min_count, c->kmc_max_count, (u64 *)d[0], (u64 *)d[1], (u32 *)d[2]);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(hi_out + r0, d[0], nr * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(lo_out + r0, d[1], nr * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipMemcpyAsync(cnt_out + r0, d[2], nr * 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return MG_OK;
}

MG_EXPORT int mg_scan_stats(mg_ctx *c, float *ms_out, uint64_t *n_hits)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    if (!c->stats_valid || c->ev_rows.empty()) return fail(c, MG_ERR_STATE, "no scan has run");
    // per launch group (chunk) of the most recent scan, AVERAGED over its chunks by rows: the sums of the three phases over all
    // timed chunks, scaled to the rows of the first (largest) one -- what a rocprofv3 --stats average of the same kernels gives
    // when every chunk is full, and the same rate when the last one is not
    const size_t nc = c->ev_rows.size();
    HIP_TRY(c, hipEventSynchronize(c->ev[4 * nc - 1]));
    if (ms_out) {
        double sum[3] = {0, 0, 0};
        u64 rows = 0;
        for (size_t q = 0; q < nc; ++q) {
            for (int i = 0; i < 3; ++i) {
                float ms = 0;
                HIP_TRY(c, hipEventElapsedTime(&ms, c->ev[4 * q + i], c->ev[4 * q + i + 1]));
                sum[i] += ms;
            }
            rows += c->ev_rows[q];
        }
        for (int i = 0; i < 3; ++i) ms_out[i] = (float)(sum[i] * (double)c->ev_rows[0] / (double)(rows ? rows : 1));
    }
    if (n_hits) {
        unsigned long long t[4] = {0, 0, 0, 0};
        HIP_TRY(c, hipMemcpy(t, c->d_hit_count, 32, hipMemcpyDeviceToHost));
        n_hits[0] = t[0]; // open rows of the last chunk
        n_hits[1] = t[2]; // bf hit rows of the whole call
    }
    return MG_OK;
}

MG_EXPORT int mg_debug_packed_index(mg_ctx *c, int which, const uint64_t *hi, const uint64_t *lo, size_t n, uint32_t klen,
                                    uint64_t *out)
{
    const DeviceGuard on_device(c, KEEP);
    TRY(check_which(c, which));
    if (klen < 1 || klen > MG_MAX_PACKED_K) return fail(c, MG_ERR_LIMIT, "packed k-mers: 1 <= k <= 64");
    if (n == 0) return MG_OK;
    void *dh, *dl, *dout;
    TRY(upload(c, c->s_misc[5], hi, n * 8, &dh));
    TRY(upload(c, c->s_misc[6], lo, n * 8, &dl));
    TRY(scratch(c, c->s_out, n * 8, &dout));
    hipLaunchKernelGGL(packed_index_kernel, dim3(nblocks(n)), dim3(TPB), 0, c->stream, (const u64 *)dh, (const u64 *)dl,
                       (u64)n, (int)klen, c->bf[which].mod, (u64 *)dout);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(out, dout, n * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}

// ---- counters exchange --------------------------------------------------------------------

MG_EXPORT int mg_counters_size(mg_ctx *c, uint64_t *n_bf, uint64_t *n_map)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    if (n_bf) *n_bf = c->bf[0].nset;
    if (n_map) *n_map = c->map.rows_total;
    return MG_OK;
}
namespace {
// the two counter arrays as ONE allocation [bf counters | map counters] (what an in-place all-reduce wants)
int ensure_joined(mg_ctx *c)
{
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    BFState &b = c->bf[MG_BF_ALT];
    MapState &m = c->map;
    if (!c->joined) {
        const u64 nb = b.nset, nm = m.rows_total;
        u32 *j = nullptr;
        HIP_TRY(c, hipMalloc(&j, (nb + nm ? nb + nm : 1) * 4));
        if (nb) HIP_TRY(c, hipMemcpyAsync(j, b.counts, nb * 4, hipMemcpyDeviceToDevice, c->stream));
        if (nm) HIP_TRY(c, hipMemcpyAsync(j + nb, m.vals, nm * 4, hipMemcpyDeviceToDevice, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        hipFree(b.counts);
        hipFree(m.vals);
        c->joined = j;
        b.counts = j;
        m.vals = j + nb;
        m.vals_cap = nm;
    }
    return MG_OK;
}
} // namespace

MG_EXPORT int mg_counters_view(mg_ctx *c, void **d_ptr, uint64_t *n_bf, uint64_t *n_map)
{
    const DeviceGuard on_device(c);
    if (!c || !d_ptr) return MG_ERR_ARG;
    TRY(ensure_joined(c));
    c->rec_off = true; // the caller may write the vector (an all-reduce by other means): from here on it alone holds the counters
    *d_ptr = c->joined;
    if (n_bf) *n_bf = c->bf[MG_BF_ALT].nset;
    if (n_map) *n_map = c->map.rows_total;
    return MG_OK;
}
MG_EXPORT int mg_counters_export_device(mg_ctx *c, void *d_out)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c || !d_out) return MG_ERR_ARG;
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    const u64 nb = c->bf[0].nset, nm = c->map.rows_total;
    if (nb) HIP_TRY(c, hipMemcpyAsync(d_out, c->bf[0].counts, nb * 4, hipMemcpyDeviceToDevice, c->stream));
    if (nm) HIP_TRY(c, hipMemcpyAsync((u32 *)d_out + nb, c->map.vals, nm * 4, hipMemcpyDeviceToDevice, c->stream));
    return MG_OK;
}
MG_EXPORT int mg_counters_import_device(mg_ctx *c, const void *d_in)
{
    const DeviceGuard on_device(c);
    if (!c || !d_in) return MG_ERR_ARG;
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    const u64 nb = c->bf[0].nset, nm = c->map.rows_total;
    if (nb) HIP_TRY(c, hipMemcpyAsync(c->bf[0].counts, d_in, nb * 4, hipMemcpyDeviceToDevice, c->stream));
    if (nm) HIP_TRY(c, hipMemcpyAsync(c->map.vals, (const u32 *)d_in + nb, nm * 4, hipMemcpyDeviceToDevice, c->stream));
    return MG_OK;
}
MG_EXPORT int mg_counters_reset(mg_ctx *c)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    if (!c->vec_zero) { // (vectors that nothing has written since the last reset -- lazy scans only -- are still all zero)
        if (c->bf[0].mode && c->bf[0].nset) HIP_TRY(c, hipMemsetAsync(c->bf[0].counts, 0, c->bf[0].nset * 4, c->stream));
        if (c->map.rows_total) HIP_TRY(c, hipMemsetAsync(c->map.vals, 0, c->map.rows_total * 4, c->stream));
    }
    c->vec_zero = !c->joined && !c->rec_off; // (a vector that has been handed out may be written behind the library's back)
    c->vec_stale = false;                    // (whatever the records were ahead by is gone with their epoch)
    for (auto &kv : c->map.irregular) kv.second = 0;
    // every record's copy is now of an older epoch: it reads as zero, like the vectors just cleared.  At the wrap of the 32-bit
    // epoch the copies written 2^32 resets ago would read as current again: the next scan republishes every record instead
    // (records_current rewrites all of them from the vectors, which are zero)
    if (++c->rec_epoch == 0) {
        c->rec_epoch = 1;
        c->rec_ok = false;
    } else
        c->rec_ok = records_wanted(c) && c->map.slots != nullptr;
    return MG_OK;
}

// ---- multi-GPU exchange: RCCL over xGMI -----------------------------------------------------
// The scan shards by table rows; what has to be combined afterwards is one vector of wrapping u32 sums
// (mg_counters_view).  One ncclAllReduce(sum, uint32) over it, in place, on the context's stream.
// librccl is opened on first use: a single-GPU run never maps its half gigabyte, and a process that has already
// mapped a librccl.so.1 (PyTorch-ROCm bundles one) gets that copy -- one RCCL, like one HIP runtime, per process.
namespace {
struct Rccl {
    void *h = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string err;
};
Rccl *rccl()
{
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.h) break;
        }
        if (!r.h) {
            const char *e = dlerror();
            r.err = std::string("cannot open librccl.so.1: ") + (e ? e : "?");
            return;
        }
        bool ok = true;
        auto sym = [&](const char *n) {
            void *p = dlsym(r.h, n);
            if (!p) {
                ok = false;
                r.err = std::string("librccl lacks ") + n;
            }
            return p;
        };
        r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
        r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
        r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
        r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
        r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
        r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
        r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
        if (!ok) r.h = nullptr;
    });
    return r.h ? &r : nullptr;
}
#define NCCL_TRY(c, R, expr)                                                                           \
    do {                                                                                               \
        ncclResult_t r_ = (expr);                                                                      \
        if (r_ != ncclSuccess) return fail(c, MG_ERR_COMM, "%s: %s", #expr, (R)->GetErrorString(r_)); \
    } while (0)

// sum of up to MG_MAX_LOCAL counter vectors that live on ONE device, written back to all of them
constexpr int MG_MAX_LOCAL = 16;
struct LocalPtrs {
    u32 *p[MG_MAX_LOCAL];
    int n;
};
__global__ void __launch_bounds__(TPB) local_sum_kernel(LocalPtrs v, u64 n)
{
    for (u64 i = (u64)blockIdx.x * TPB + threadIdx.x; i < n; i += (u64)gridDim.x * TPB) {
        u32 s = 0;
        for (int j = 0; j < v.n; ++j) s += v.p[j][i];
        for (int j = 0; j < v.n; ++j) v.p[j][i] = s;
    }
}
// ---- the 16-bit packed form of the exchange --------------------------------------------------------------------------------
// Two counters travel in one 32-bit word (low and high half).  That is exact iff no half can carry into its neighbour, i.e.
// iff every rank's every partial counter is <= 65535 / world: checked first (a max over the vector, then over the ranks), the
// plain 32-bit sum runs otherwise.  Partial counters are small in practice: KMC lists each distinct k-mer once, so a counter
// receives one count (<= 255) plus those of the few k-mers that share it.  Half the bytes on the links.
__global__ void __launch_bounds__(TPB) vec_max_kernel(const u32 *__restrict__ v, u64 n, u32 *out)
{
    u32 m = 0;
    for (u64 i = (u64)blockIdx.x * TPB + threadIdx.x; i < n; i += (u64)gridDim.x * TPB) m = max(m, v[i]);
    for (int o = 32; o > 0; o >>= 1) m = max(m, (u32)__shfl_down((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}
__global__ void __launch_bounds__(TPB) pack16_kernel(const u32 *__restrict__ v, u64 n, u32 *__restrict__ packed)
{
    const u64 m = n / 2;
    for (u64 i = (u64)blockIdx.x * TPB + threadIdx.x; i < m + (n & 1); i += (u64)gridDim.x * TPB)
        packed[i] = i < m ? (v[2 * i] | v[2 * i + 1] << 16) : v[n - 1]; // (an odd last counter travels whole)
}
__global__ void __launch_bounds__(TPB) unpack16_kernel(u32 *__restrict__ v, u64 n, const u32 *__restrict__ packed)
{
    const u64 m = n / 2;
    for (u64 i = (u64)blockIdx.x * TPB + threadIdx.x; i < m + (n & 1); i += (u64)gridDim.x * TPB) {
        const u32 w = packed[i];
        if (i < m) {
            v[2 * i] = w & 0xFFFFu;
            v[2 * i + 1] = w >> 16;
        } else
            v[n - 1] = w;
    }
}
bool pack_wanted(const mg_ctx *c, u64 nn, int world)
{
    if (!c->exchange_pack || nn < 2 || world > 16) return false;
    return c->exchange_pack == 2 || nn * 4 >= (u64)c->exchange_pack_min_mb << 20;
}
// this context's largest counter -> *c->d_xmax (device), on stream `st`
int local_max(mg_ctx *c, u64 nn, hipStream_t st)
{
    if (!c->d_xmax) HIP_TRY(c, hipMalloc(&c->d_xmax, 8));
    HIP_TRY(c, hipMemsetAsync(c->d_xmax, 0, 4, st));
    hipLaunchKernelGGL(vec_max_kernel, dim3((unsigned)std::min<u64>(nblocks(nn), 2048u)), dim3(TPB), 0, st, (const u32 *)c->joined, nn, c->d_xmax);
    HIP_TRY(c, hipGetLastError());
    return MG_OK;
}
int pack_into_scratch(mg_ctx *c, u64 nn, hipStream_t st, u32 **out)
{
    void *p;
    TRY(scratch(c, c->s_pack, (nn / 2 + 1) * 4, &p));
    hipLaunchKernelGGL(pack16_kernel, dim3((unsigned)std::min<u64>(nblocks(nn / 2 + 1), 4096u)), dim3(TPB), 0, st, (const u32 *)c->joined, nn, (u32 *)p);
    HIP_TRY(c, hipGetLastError());
    *out = (u32 *)p;
    return MG_OK;
}
int comm_drop(mg_ctx *c)
{
    if (c->comm) {
        Rccl *R = rccl();
        if (R) R->CommDestroy(c->comm);
        c->comm = nullptr;
    }
    for (mg_ctx *o : c->local_group) // a local group dissolves as a whole
        if (o != c) o->local_group.clear();
    c->local_group.clear();
    c->comm_world = 0;
    c->comm_rank = 0;
    return MG_OK;
}
} // namespace

MG_EXPORT int mg_comm_unique_id(void *id_out)
{
    if (!id_out) return MG_ERR_ARG;
    Rccl *R = rccl();
    if (!R) return MG_ERR_COMM;
    static_assert(sizeof(ncclUniqueId) == MG_COMM_ID_BYTES, "MG_COMM_ID_BYTES");
    ncclUniqueId id;
    if (R->GetUniqueId(&id) != ncclSuccess) return MG_ERR_COMM;
    memcpy(id_out, &id, sizeof id);
    return MG_OK;
}

MG_EXPORT int mg_comm_init(mg_ctx *c, int rank, int world, const void *id)
{
    const DeviceGuard on_device(c);
    if (!c) return MG_ERR_ARG;
    if (world < 1 || rank < 0 || rank >= world || !id) return fail(c, MG_ERR_ARG, "mg_comm_init: rank %d of %d", rank, world);
    Rccl *R = rccl();
    if (!R) return fail(c, MG_ERR_COMM, "RCCL unavailable");
    comm_drop(c);
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    NCCL_TRY(c, R, R->CommInitRank(&c->comm, world, uid, rank));
    c->comm_rank = rank;
    c->comm_world = world;
    return MG_OK;
}

MG_EXPORT int mg_comm_init_all(mg_ctx **ctxs, int n)
{
    if (!ctxs || n < 1 || n > 64) return MG_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return MG_ERR_ARG;
    bool distinct = true, same = true;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < i; ++j) {
            if (ctxs[i] == ctxs[j]) return fail(ctxs[0], MG_ERR_ARG, "mg_comm_init_all: context %d listed twice", i);
            if (ctxs[i]->device == ctxs[j]->device) distinct = false;
            else same = false;
        }
    for (int i = 0; i < n; ++i) comm_drop(ctxs[i]);
    if (n == 1 || distinct) { // one rank per device: RCCL (n == 1 too, so that a single-GPU run drives the same calls)
        Rccl *R = rccl();
        if (!R) return fail(ctxs[0], MG_ERR_COMM, "RCCL unavailable");
        ncclUniqueId uid;
        NCCL_TRY(ctxs[0], R, R->GetUniqueId(&uid));
        int prev = -1;
        hipGetDevice(&prev);
        NCCL_TRY(ctxs[0], R, R->GroupStart());
        ncclResult_t bad = ncclSuccess;
        for (int i = 0; i < n && bad == ncclSuccess; ++i) {
            hipSetDevice(ctxs[i]->device);
            bad = R->CommInitRank(&ctxs[i]->comm, n, uid, i);
        }
        const ncclResult_t ge = R->GroupEnd();
        if (prev >= 0) hipSetDevice(prev);
        if (bad != ncclSuccess || ge != ncclSuccess) {
            for (int i = 0; i < n; ++i) { // communicators the group did create are destroyed, not dropped
                if (ctxs[i]->comm) R->CommDestroy(ctxs[i]->comm);
                ctxs[i]->comm = nullptr;
            }
            return fail(ctxs[0], MG_ERR_COMM, "ncclCommInitRank x%d: %s", n, R->GetErrorString(bad != ncclSuccess ? bad : ge));
        }
        for (int i = 0; i < n; ++i) {
            ctxs[i]->comm_rank = i;
            ctxs[i]->comm_world = n;
        }
        return MG_OK;
    }
    if (!same) return fail(ctxs[0], MG_ERR_ARG, "mg_comm_init_all: contexts must sit on distinct devices, or all on one");
    if (n > MG_MAX_LOCAL) return fail(ctxs[0], MG_ERR_LIMIT, "at most %d contexts may share a device", MG_MAX_LOCAL);
    // all on one device (a rehearsal of the N-GPU layout on a one-GPU box: RCCL rejects duplicate devices)
    for (int i = 0; i < n; ++i) {
        ctxs[i]->local_group.assign(ctxs, ctxs + n);
        ctxs[i]->comm_rank = i;
        ctxs[i]->comm_world = n;
        if (!ctxs[i]->ev_x) {
            const DeviceGuard g(ctxs[i]);
            if (hipEventCreateWithFlags(&ctxs[i]->ev_x, hipEventDisableTiming) != hipSuccess) return fail(ctxs[i], MG_ERR_HIP, "hipEventCreate");
        }
    }
    return MG_OK;
}

MG_EXPORT int mg_comm_destroy(mg_ctx *c)
{
    const DeviceGuard on_device(c);
    if (!c) return MG_ERR_ARG;
    return comm_drop(c);
}

MG_EXPORT int mg_comm_info(mg_ctx *c, int *rank, int *world, int *backend)
{
    if (!c) return MG_ERR_ARG;
    if (rank) *rank = c->comm_rank;
    if (world) *world = c->comm_world;
    if (backend) *backend = c->comm ? MG_COMM_RCCL : !c->local_group.empty() ? MG_COMM_LOCAL : MG_COMM_NONE;
    return MG_OK;
}

namespace {
// The exchange of one context's counters over its RCCL communicator, enqueued on `st`.  With the packed form the call waits
// (the host has to see the ranks' largest partial counter to choose the form; every rank sees the same one).
int exchange_rccl(mg_ctx *c, hipStream_t st)
{
    Rccl *R = rccl();
    const u64 nn = c->bf[MG_BF_ALT].nset + c->map.rows_total;
    c->last_exchange_packed = 0;
    if (nn == 0) return MG_OK;
    bool packed = false;
    if (pack_wanted(c, nn, c->comm_world)) {
        TRY(local_max(c, nn, st));
        NCCL_TRY(c, R, R->AllReduce(c->d_xmax, c->d_xmax, 1, ncclUint32, ncclMax, c->comm, st));
        u32 mx = 0;
        HIP_TRY(c, hipMemcpyAsync(&mx, c->d_xmax, 4, hipMemcpyDeviceToHost, st));
        HIP_TRY(c, hipStreamSynchronize(st));
        packed = mx <= 65535u / (u32)c->comm_world;
    }
    if (c->x_timed) HIP_TRY(c, hipEventRecord(c->ev_xs[2], st));
    if (packed) {
        u32 *pk;
        TRY(pack_into_scratch(c, nn, st, &pk));
        NCCL_TRY(c, R, R->AllReduce(pk, pk, nn / 2 + (nn & 1), ncclUint32, ncclSum, c->comm, st));
        hipLaunchKernelGGL(unpack16_kernel, dim3((unsigned)std::min<u64>(nblocks(nn / 2 + 1), 4096u)), dim3(TPB), 0, st, c->joined, nn, (const u32 *)pk);
        HIP_TRY(c, hipGetLastError());
    } else
        NCCL_TRY(c, R, R->AllReduce(c->joined, c->joined, nn, ncclUint32, ncclSum, c->comm, st));
    if (c->x_timed) HIP_TRY(c, hipEventRecord(c->ev_xs[3], st));
    c->last_exchange_packed = packed ? 1 : 0;
    return MG_OK;
}
int exchange_ready(mg_ctx *c)
{
    if (!c->comm) return fail(c, MG_ERR_STATE, "no RCCL communicator (mg_comm_init first; contexts sharing a device use mg_counters_allreduce_all)");
    TRY(ensure_joined(c));
    for (int i = 0; i < 4; ++i)
        if (!c->ev_xs[i]) HIP_TRY(c, i < 2 ? hipEventCreateWithFlags(&c->ev_xs[i], hipEventDisableTiming) : hipEventCreate(&c->ev_xs[i]));
    c->x_timed = true;
    return MG_OK;
}
} // namespace

MG_EXPORT int mg_counters_allreduce(mg_ctx *c)
{
    const DeviceGuard on_device(c);
    if (!c) return MG_ERR_ARG;
    TRY(exchange_ready(c));
    return exchange_rccl(c, c->stream);
}
// The same exchange on a stream of its own: _begin orders it behind everything the context's stream holds so far (the scan),
// _end makes the context's stream wait for it.  In between the caller may enqueue what needs no counters -- the block cut of the
// record loop (mg_cut_blocks_device) -- which then runs beside the collective instead of behind it.
MG_EXPORT int mg_counters_allreduce_begin(mg_ctx *c)
{
    const DeviceGuard on_device(c);
    if (!c) return MG_ERR_ARG;
    if (c->x_pending) return fail(c, MG_ERR_STATE, "mg_counters_allreduce_begin twice without mg_counters_allreduce_end");
    TRY(exchange_ready(c));
    if (!c->xstream) HIP_TRY(c, hipStreamCreateWithFlags(&c->xstream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventRecord(c->ev_xs[0], c->stream));
    HIP_TRY(c, hipStreamWaitEvent(c->xstream, c->ev_xs[0], 0));
    TRY(exchange_rccl(c, c->xstream));
    HIP_TRY(c, hipEventRecord(c->ev_xs[1], c->xstream));
    c->x_pending = true;
    return MG_OK;
}
MG_EXPORT int mg_counters_allreduce_end(mg_ctx *c)
{
    const DeviceGuard on_device(c);
    if (!c) return MG_ERR_ARG;
    if (!c->x_pending) return fail(c, MG_ERR_STATE, "mg_counters_allreduce_end without mg_counters_allreduce_begin");
    HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_xs[1], 0));
    c->x_pending = false;
    return MG_OK;
}
// duration of the most recent exchange of this context (its collective with pack / unpack, without the guard's max), and its form
MG_EXPORT int mg_exchange_stats(mg_ctx *c, float *ms_out, int *packed_out)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    if (!c->x_timed || !c->ev_xs[3]) return fail(c, MG_ERR_STATE, "no exchange has run");
    HIP_TRY(c, hipEventSynchronize(c->ev_xs[3]));
    if (ms_out) HIP_TRY(c, hipEventElapsedTime(ms_out, c->ev_xs[2], c->ev_xs[3]));
    if (packed_out) *packed_out = c->last_exchange_packed;
    return MG_OK;
}

MG_EXPORT int mg_counters_allreduce_all(mg_ctx **ctxs, int n)
{
    if (!ctxs || n < 1) return MG_ERR_ARG;
    for (int i = 0; i < n; ++i)
        if (!ctxs[i]) return MG_ERR_ARG;
    mg_ctx *c0 = ctxs[0];
    u64 nn = 0;
    for (int i = 0; i < n; ++i) {
        const DeviceGuard g(ctxs[i]);
        TRY(ensure_joined(ctxs[i]));
        const u64 ni = ctxs[i]->bf[MG_BF_ALT].nset + ctxs[i]->map.rows_total;
        if (i && (ni != nn || ctxs[i]->bf[MG_BF_ALT].nset != c0->bf[MG_BF_ALT].nset))
            return fail(c0, MG_ERR_STATE, "context %d holds another index (%llu counters, context 0 %llu): every rank must load the same index",
                        i, (unsigned long long)ni, (unsigned long long)nn);
        nn = ni;
        if (ctxs[i]->comm_world != n) return fail(c0, MG_ERR_STATE, "context %d is not part of a %d-way group (mg_comm_init_all)", i, n);
        ctxs[i]->last_exchange_packed = 0;
    }
    if (nn == 0) return MG_OK;
    // the packed form's guard: every context's largest partial counter, seen by this one process (no collective needed)
    bool packed = false;
    if (pack_wanted(c0, nn, n)) {
        u32 mx = 0;
        for (int i = 0; i < n; ++i) {
            const DeviceGuard g(ctxs[i]);
            TRY(local_max(ctxs[i], nn, ctxs[i]->stream));
        }
        for (int i = 0; i < n; ++i) {
            const DeviceGuard g(ctxs[i]);
            u32 m1 = 0;
            HIP_TRY(ctxs[i], hipMemcpyAsync(&m1, ctxs[i]->d_xmax, 4, hipMemcpyDeviceToHost, ctxs[i]->stream));
            HIP_TRY(ctxs[i], hipStreamSynchronize(ctxs[i]->stream));
            mx = std::max(mx, m1);
        }
        packed = mx <= 65535u / (u32)n;
    }
    const u64 np = nn / 2 + (nn & 1);
    if (c0->comm) { // one rank per device, one process: a group of all-reduces, each on its context's stream
        Rccl *R = rccl();
        int prev = -1;
        hipGetDevice(&prev);
        std::vector<u32 *> pk((size_t)n, nullptr);
        if (packed)
            for (int i = 0; i < n; ++i) {
                hipSetDevice(ctxs[i]->device);
                TRY(pack_into_scratch(ctxs[i], nn, ctxs[i]->stream, &pk[(size_t)i]));
            }
        NCCL_TRY(c0, R, R->GroupStart());
        ncclResult_t bad = ncclSuccess;
        for (int i = 0; i < n && bad == ncclSuccess; ++i) {
            hipSetDevice(ctxs[i]->device);
            bad = packed ? R->AllReduce(pk[(size_t)i], pk[(size_t)i], np, ncclUint32, ncclSum, ctxs[i]->comm, ctxs[i]->stream)
                         : R->AllReduce(ctxs[i]->joined, ctxs[i]->joined, nn, ncclUint32, ncclSum, ctxs[i]->comm, ctxs[i]->stream);
        }
        const ncclResult_t ge = R->GroupEnd();
        if (packed && bad == ncclSuccess && ge == ncclSuccess)
            for (int i = 0; i < n; ++i) {
                hipSetDevice(ctxs[i]->device);
                hipLaunchKernelGGL(unpack16_kernel, dim3((unsigned)std::min<u64>(nblocks(np), 4096u)), dim3(TPB), 0, ctxs[i]->stream, ctxs[i]->joined, nn, (const u32 *)pk[(size_t)i]);
                ctxs[i]->last_exchange_packed = 1;
            }
        if (prev >= 0) hipSetDevice(prev);
        if (bad != ncclSuccess || ge != ncclSuccess) return fail(c0, MG_ERR_COMM, "ncclAllReduce x%d: %s", n, R->GetErrorString(bad != ncclSuccess ? bad : ge));
        return MG_OK;
    }
    if (c0->local_group.size() != (size_t)n) return fail(c0, MG_ERR_STATE, "mg_comm_init_all first");
    // contexts sharing one device: context 0's stream waits for every scan, sums (the packed words, where that form applies:
    // the same pack / unpack kernels as on the wire), and everyone waits for the sum
    const DeviceGuard g(c0);
    LocalPtrs v{};
    v.n = n;
    for (int i = 0; i < n; ++i) {
        if (i) {
            HIP_TRY(c0, hipEventRecord(ctxs[i]->ev_x, ctxs[i]->stream));
            HIP_TRY(c0, hipStreamWaitEvent(c0->stream, ctxs[i]->ev_x, 0));
        }
        if (packed) TRY(pack_into_scratch(ctxs[i], nn, c0->stream, &v.p[i]));
        else v.p[i] = ctxs[i]->joined;
    }
    hipLaunchKernelGGL(local_sum_kernel, dim3((unsigned)std::min<u64>(nblocks(packed ? np : nn), 4096u)), dim3(TPB), 0, c0->stream, v, packed ? np : nn);
    HIP_TRY(c0, hipGetLastError());
    if (packed)
        for (int i = 0; i < n; ++i) {
            hipLaunchKernelGGL(unpack16_kernel, dim3((unsigned)std::min<u64>(nblocks(np), 4096u)), dim3(TPB), 0, c0->stream, ctxs[i]->joined, nn, (const u32 *)v.p[i]);
            ctxs[i]->last_exchange_packed = 1;
        }
    HIP_TRY(c0, hipGetLastError());
    HIP_TRY(c0, hipEventRecord(c0->ev_x, c0->stream));
    for (int i = 1; i < n; ++i) HIP_TRY(c0, hipStreamWaitEvent(ctxs[i]->stream, c0->ev_x, 0));
    return MG_OK;
}

// ---- per-variant path -----------------------------------------------------------------------

MG_EXPORT int mg_lookup_cover(mg_ctx *c, const char *rows, size_t stride, size_t n_rows, const uint8_t *is_ref,
                              const uint64_t *sig_kmer_off, size_t n_sigs, const uint64_t *allele_sig_off, size_t n_alleles,
                              uint32_t *cov_out)
{
    const DeviceGuard on_device(c, KEEP);
    TRY(check_rows(c, rows, stride, n_rows));
    if (n_alleles == 0) return MG_OK;
    if (!sig_kmer_off || !allele_sig_off || !cov_out || (n_rows && !is_ref)) return fail(c, MG_ERR_ARG, "NULL descriptor");
    if (allele_sig_off[n_alleles] != n_sigs || sig_kmer_off[n_sigs] != n_rows)
        return fail(c, MG_ERR_ARG, "descriptor offsets do not close (sigs %zu rows %zu)", n_sigs, n_rows);
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    if (!c->map.slots) TRY(map_reserve(c, 0));
    void *d_rows, *d_isref, *d_w, *d_so, *d_ao, *d_cov;
    TRY(upload(c, c->s_rows, rows, stride * n_rows, &d_rows));
    TRY(upload(c, c->s_misc[0], is_ref, n_rows, &d_isref));
    TRY(scratch(c, c->s_out, 4 * (n_rows ? n_rows : 1), &d_w));
    TRY(upload(c, c->s_misc[2], sig_kmer_off, 8 * (n_sigs + 1), &d_so));
    TRY(upload(c, c->s_misc[3], allele_sig_off, 8 * (n_alleles + 1), &d_ao));
    TRY(scratch(c, c->s_misc[4], 4 * n_alleles, &d_cov));
    if (n_rows) {
        // exact-map keys the packed table cannot hold (every key, when k > MG_MAX_PACKED_K) live in the context's host list:
        // the kernel marks such rows and their values are filled in from there (KMAP::get_count, kmap.hpp:124-131)
        void *d_irr = nullptr;
        const bool host_keys = !c->map.irregular.empty();
        if (host_keys) {
            TRY(scratch(c, c->s_irr, n_rows, &d_irr));
            HIP_TRY(c, hipMemsetAsync(d_irr, 0, n_rows, c->stream));
        }
        hipLaunchKernelGGL(rows_kernel<OP_WEIGHT>, dim3(nblocks(n_rows)), dim3(TPB), 0, c->stream, (const u8 *)d_rows, stride,
                           n_rows, view(c, MG_BF_ALT), view(c), (const u32 *)nullptr, (const u8 *)d_isref, d_w,
                           (u8 *)d_irr);
        HIP_TRY(c, hipGetLastError());
        if (host_keys) {
            std::vector<u8> irr(n_rows);
            std::vector<i32> w(n_rows);
            HIP_TRY(c, hipMemcpyAsync(irr.data(), d_irr, n_rows, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipMemcpyAsync(w.data(), d_w, 4 * n_rows, hipMemcpyDeviceToHost, c->stream));
            HIP_TRY(c, hipStreamSynchronize(c->stream));
            bool any = false;
            for (size_t i = 0; i < n_rows; ++i)
                if (irr[i]) {
                    auto it = c->map.irregular.find(host_irregular_key(rows + i * stride, stride));
                    w[i] = it != c->map.irregular.end() ? it->second : 0;
                    any = true;
                }
            if (any) {
                HIP_TRY(c, hipMemcpyAsync(d_w, w.data(), 4 * n_rows, hipMemcpyHostToDevice, c->stream));
                HIP_TRY(c, hipStreamSynchronize(c->stream)); // (w leaves scope)
            }
        }
    }
    hipLaunchKernelGGL(cover_kernel, dim3(nblocks(n_alleles)), dim3(TPB), 0, c->stream, (const i32 *)d_w, (const u64 *)d_so,
                       (const u64 *)d_ao, (u64)n_alleles, (u32 *)d_cov);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(cov_out, d_cov, 4 * n_alleles, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}

MG_EXPORT int mg_genotype_device(mg_ctx *c, const void *d_cov, const void *d_freq, const void *d_var_allele_off, size_t n_vars, float error_rate,
                                 int max_cov, int haploid, void *d_gt1, void *d_gt2, void *d_gq, void *d_status, void *d_probs, const void *d_var_gt_off)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    if (n_vars == 0) return MG_OK;
    if (!d_cov || !d_freq || !d_var_allele_off || !d_gt1 || !d_gt2 || !d_gq || !d_status) return fail(c, MG_ERR_ARG, "NULL argument");
    if (d_probs && !d_var_gt_off) return fail(c, MG_ERR_ARG, "probs needs var_gt_off");
    GenoParams p;
    TRY(fill_geno_params(c, error_rate, max_cov, haploid, &p));
    hipLaunchKernelGGL(genotype_kernel, dim3(nblocks(n_vars)), dim3(TPB), 0, c->stream, (const u32 *)d_cov, (const float *)d_freq,
                       (const u32 *)d_var_allele_off, (u64)n_vars, p, (i32 *)d_gt1, (i32 *)d_gt2, (i32 *)d_gq, (u8 *)d_status, (double *)d_probs,
                       (const u64 *)d_var_gt_off);
    HIP_TRY(c, hipGetLastError());
    return MG_OK;
}

MG_EXPORT int mg_genotype(mg_ctx *c, const uint32_t *cov, const float *freq, const uint32_t *var_allele_off, size_t n_vars,
                          float error_rate, int max_cov, int haploid, int32_t *gt1, int32_t *gt2, int32_t *gq, uint8_t *status,
                          double *probs, const uint64_t *var_gt_off)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    if (n_vars == 0) return MG_OK;
    if (!cov || !freq || !var_allele_off || !gt1 || !gt2 || !gq || !status) return fail(c, MG_ERR_ARG, "NULL argument");
    if (probs && !var_gt_off) return fail(c, MG_ERR_ARG, "probs needs var_gt_off");
    const size_t na = var_allele_off[n_vars];
    const size_t ng = probs ? var_gt_off[n_vars] : 0;
    void *d_cov, *d_freq, *d_off, *d_g1, *d_g2, *d_gq, *d_st, *d_pr = nullptr, *d_go = nullptr;
    TRY(upload(c, c->s_rows, cov, 4 * na, &d_cov));
    TRY(upload(c, c->s_aux, freq, 4 * na, &d_freq));
    TRY(upload(c, c->s_misc[0], var_allele_off, 4 * (n_vars + 1), &d_off));
    TRY(scratch(c, c->s_misc[1], 4 * n_vars, &d_g1));
    TRY(scratch(c, c->s_misc[2], 4 * n_vars, &d_g2));
    TRY(scratch(c, c->s_misc[3], 4 * n_vars, &d_gq));
    TRY(scratch(c, c->s_misc[4], n_vars, &d_st));
    if (probs) {
        TRY(scratch(c, c->s_out, 8 * (ng ? ng : 1), &d_pr));
        TRY(upload(c, c->s_misc[5], var_gt_off, 8 * (n_vars + 1), &d_go));
    }
    TRY(mg_genotype_device(c, d_cov, d_freq, d_off, n_vars, error_rate, max_cov, haploid, d_g1, d_g2, d_gq, d_st, d_pr, d_go));
    HIP_TRY(c, hipMemcpyAsync(gt1, d_g1, 4 * n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(gt2, d_g2, 4 * n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(gq, d_gq, 4 * n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(status, d_st, n_vars, hipMemcpyDeviceToHost, c->stream));
    if (probs && ng) HIP_TRY(c, hipMemcpyAsync(probs, d_pr, 8 * ng, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}

namespace {
// ---- blocks on the device ------------------------------------------------------------------------------------------------
// flags (one byte per record, set at the first record of every block) -> blk_var_off / var_block / the block count:
// per-tile counts are already in `tile_sums` (cut_flags_kernel / flag_count_kernel wrote them)
int scan_flags(mg_ctx *c, u64 n, const u8 *d_flags, u32 *d_tile_sums, u32 *d_blk_var_off, u32 *d_var_block, unsigned long long *d_n_blocks)
{
    const u64 n_tiles = nblocks(n);
    TRY(launch_tile_scan(c, d_tile_sums, n_tiles, c->d_hit_count + 2)); // (the total is re-derived by the scatter)
    hipLaunchKernelGGL(flag_scatter_kernel, dim3((unsigned)n_tiles), dim3(TPB), 0, c->stream, n, d_flags, (const u32 *)d_tile_sums, d_blk_var_off, d_var_block,
                       d_n_blocks);
    HIP_TRY(c, hipGetLastError());
    return MG_OK;
}
int check_panel(mg_ctx *c, const mg_panel_dev *p, bool need_gt)
{
    if (!p) return fail(c, MG_ERR_ARG, "NULL panel");
    if (p->n_vars >= 0xFFFFFFFFull) return fail(c, MG_ERR_LIMIT, "fewer than 2^32 records per batch");
    if (p->n_vars == 0) return MG_OK;
    if (!p->contig_id || !p->pos || !p->ref_size || !p->min_size) return fail(c, MG_ERR_ARG, "NULL panel array");
    if (need_gt && (!p->contig_base || !p->contig_len || !p->present || !p->var_allele_off || !p->allele_off || !p->pool || !p->canon ||
                    (p->n_samples && !p->gt && !p->sp_off) || (p->sp_off && (!p->sp_sample || !p->sp_gt))))
        return fail(c, MG_ERR_ARG, "NULL panel array");
    return MG_OK;
}
PanelView panel_view(const mg_panel_dev *p)
{
    PanelView P{};
    P.contig_base = p->contig_base; P.contig_len = p->contig_len; P.contig_id = p->contig_id; P.pos = p->pos;
    P.ref_size = p->ref_size; P.min_size = p->min_size; P.present = p->present; P.var_allele_off = p->var_allele_off;
    P.allele_off = p->allele_off; P.canon = p->canon; P.gt = p->gt; P.n_samples = p->n_samples;
    P.sp_off = p->sp_off; P.sp_sample = p->sp_sample; P.sp_gt = p->sp_gt; P.sp_default = p->sp_default;
    return P;
}
// The tiers of block_pipeline.h over one resident panel.  Counts stay on the device; c->d_gen_count holds
//   [0] general records listed by tier 1   [1] insertion-row cursor (index time) / block count of a host-form batch
//   [2] signature k-mers of the lone records   [3] signature k-mers tiers 2 and 3 assembled   [4] records tier 3 took
struct BlocksRun {
    mg_ctx *c;
    const mg_panel_dev *p;
    BlockBatch B;
    PanelView P;
    u32 *gen_list, *fb_list;
    u8 *fb_flag;
    CombDesc *combs;
    PickItem *items, *slides;
    u32 *retry, *order;
    unsigned long long *round_counters;
    u64 round, n_rounds;
    int cus;
};
int blocks_setup(mg_ctx *c, const mg_panel_dev *p, const u32 *d_blk_var_off, const u32 *d_var_block, const unsigned long long *d_n_blocks, int haploid, BlocksRun *out)
{
    const u64 n = p->n_vars;
    if (!c->d_ref) return fail(c, MG_ERR_STATE, "mg_reference_upload first");
    BlocksRun R{};
    R.c = c;
    R.p = p;
    // general records per round of tier 2.  The list's length lives on the device, so ceil(n / round) rounds are launched and those
    // beyond its end find nothing to do -- at ~30 us of empty launches each: 2^24 records per round (13.5 GB of round buffers at most,
    // of 288) leaves a whole-genome panel 5 rounds instead of 20
    R.round = std::min<u64>(n, 1ULL << c->blocks_round_log2);
    R.n_rounds = (n + R.round - 1) / R.round;
    void *q[8];
    TRY(scratch(c, c->s_blk[0], 4 * n, &q[0]));                                                   // gen_list
    TRY(scratch(c, c->s_blk[1], 4 * n, &q[1]));                                                   // fb_list
    TRY(scratch(c, c->s_blk[2], n, &q[2]));                                                       // fb_flag
    TRY(scratch(c, c->s_blk[3], sizeof(CombDesc) * (R.round * FW_COMBS_PER_REC + 64), &q[3]));    // one round's descriptors
    TRY(scratch(c, c->s_blk[4], sizeof(PickItem) * (R.round * FW_ITEMS_PER_REC + FW_CHUNK * (u64)(1 << 14)), &q[4])); // one round's items (+ a chunk per wave)
    TRY(scratch(c, c->s_blk[5], 8 * FW_ROUND_COUNTERS * R.n_rounds, &q[5]));                      // per round: descriptors, items, sliding items reserved, chains per length
    TRY(scratch(c, c->s_blk[6], sizeof(PickItem) * (R.round / 4 + 4096), &q[7]));                 // one round's sliding items
    void *q_retry, *q_order;
    TRY(scratch(c, c->s_blk[7], 4 * (R.round * FW_COMBS_PER_REC + 64), &q_retry));                // one round's chains to retry
    TRY(scratch(c, c->s_blk[13], 4 * (R.round * FW_COMBS_PER_REC + 64), &q_order));               // one round's chains in order of their length
    void *q_class;
    TRY(scratch(c, c->s_blk[14], 2 * n, &q_class));                                               // per listed record: REC_* and the alleles' codes
    if (!d_var_block) { // derive it from the cut: heads -> scan
        void *fl, *ts;
        TRY(scratch(c, c->s_blk[8], 4 * n, &q[6]));
        TRY(scratch(c, c->s_blk[9], n, &fl));
        TRY(scratch(c, c->s_blk[10], 4 * (u64)nblocks(n) + 4, &ts));
        HIP_TRY(c, hipMemsetAsync(fl, 0, n, c->stream));
        hipLaunchKernelGGL(block_heads_kernel, dim3(nblocks(n)), dim3(TPB), 0, c->stream, d_blk_var_off, d_n_blocks, n, (u8 *)fl);
        hipLaunchKernelGGL(flag_count_kernel, dim3(nblocks(n)), dim3(TPB), 0, c->stream, n, (const u8 *)fl, (u32 *)ts);
        TRY(scan_flags(c, n, (const u8 *)fl, (u32 *)ts, nullptr, (u32 *)q[6], nullptr));
        d_var_block = (const u32 *)q[6];
    }
    if (!c->d_gen_count) HIP_TRY(c, hipMalloc(&c->d_gen_count, 64));
    BlockBatch B{};
    B.reference = c->d_ref;
    B.ref2 = c->d_ref2;
    B.refbad = c->d_refbad;
    B.contig_base = p->contig_base; B.contig_len = p->contig_len; B.contig_id = p->contig_id;
    B.blk_var_off = d_blk_var_off; B.var_block = d_var_block;
    B.pos = p->pos; B.ref_size = p->ref_size; B.min_size = p->min_size; B.present = p->present; B.var_allele_off = p->var_allele_off;
    B.allele_off = p->allele_off; B.pool = (const u8 *)p->pool; B.canon = p->canon; B.gt = p->gt;
    B.sp_off = p->sp_off; B.sp_sample = p->sp_sample; B.sp_gt = p->sp_gt; B.sp_default = p->sp_default;
    B.n_samples = p->n_samples; B.haploid = haploid; B.k = (int)c->k;
    B.set_limit = c->blocks_set_limit;
    B.snp_chains = c->use_snp_chains;
    B.rec_class = (const unsigned short *)q_class;
    // the alleles packed like the reference (every call: the panel is the caller's) -- unless the pool is too small to hold alleles worth
    // it (a SNP panel: two one-base alleles per record; fw_eval takes alleles of up to 4 bases from the bytes anyway)
    if (p->pool_bytes && p->pool_bytes * 2 > n * 5 && ((uintptr_t)p->pool & 3) == 0 && c->use_packed_pool) {
        void *p2, *pb;
        const u64 n_words = (p->pool_bytes + 31) / 32 + 4;
        TRY(scratch(c, c->s_blk[11], 8 * n_words, &p2));
        TRY(scratch(c, c->s_blk[12], 4 * n_words, &pb));
        TRY(reference_pack(c, (const u8 *)p->pool, p->pool_bytes, (u64 *)p2, (u32 *)pb));
        B.pool2 = (const u64 *)p2;
        B.poolbad = (const u32 *)pb;
    }
    R.B = B;
    R.P = panel_view(p);
    R.gen_list = (u32 *)q[0]; R.fb_list = (u32 *)q[1]; R.fb_flag = (u8 *)q[2];
    R.combs = (CombDesc *)q[3]; R.items = (PickItem *)q[4]; R.round_counters = (unsigned long long *)q[5];
    R.slides = (PickItem *)q[7];
    R.retry = (u32 *)q_retry;
    R.order = (u32 *)q_order;
    int dev = 0;
    hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&R.cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) R.cus = 256;
    *out = R;
    return MG_OK;
}
// tier 2 over the whole general list, round by round
template <int MODE> int blocks_tier2(BlocksRun &R, u32 *d_cov, u8 *d_overflow, unsigned long long *d_cursor, u32 row0, unsigned long long *d_evaluated)
{
    mg_ctx *c = R.c;
    if (c->use_flat_tier == 0) { // A/B and tests: everything general goes to the workgroup kernel
        hipLaunchKernelGGL(flag_all_kernel, dim3(R.cus * 8), dim3(TPB), 0, c->stream, (const u32 *)R.gen_list, (const unsigned long long *)c->d_gen_count, R.fb_flag,
                           MODE == 0 ? d_cov : (u32 *)nullptr, R.B.var_allele_off);
        HIP_TRY(c, hipGetLastError());
        return MG_OK;
    }
    HIP_TRY(c, hipMemsetAsync(R.round_counters, 0, 8 * FW_ROUND_COUNTERS * R.n_rounds, c->stream));
    c->last_rounds = R.n_rounds;
    for (u64 r = 0; r < R.n_rounds; ++r) {
        FlatWork W{};
        W.gen_list = R.gen_list;
        W.gen_count = c->d_gen_count;
        W.base = r * R.round;
        W.round_len = R.round;
        W.combs = R.combs;
        W.comb_cap = (u32)(R.round * FW_COMBS_PER_REC);
        W.items = R.items;
        W.item_cap = (u32)(R.round * FW_ITEMS_PER_REC + FW_CHUNK * (u64)(1 << 14));
        W.slides = R.slides;
        W.slide_cap = (u32)(R.round / 4 + 4096);
        W.retry = R.retry;
        W.counters = R.round_counters + FW_ROUND_COUNTERS * r;
        W.order = c->use_chain_order ? R.order : nullptr;
        W.fb_flag = R.fb_flag;
        hipLaunchKernelGGL(fw_walk_kernel<MODE>, dim3(nblocks(R.round)), dim3(TPB), 0, c->stream, R.B, W, d_cov, d_overflow);
        const u32 n_haps = R.B.haploid ? R.B.n_samples : 2 * R.B.n_samples;
        if (c->use_snp_kernel && R.B.snp_chains && n_haps <= FW_SNP_MAX_HAPS) { // chains of SNPs whole, before the picks kernel sees them
            int GH = 2;
            while ((u32)GH < n_haps) GH *= 2;
            hipLaunchKernelGGL(fw_snp_kernel<MODE>, dim3(R.cus * 8), dim3(TPB), 0, c->stream, R.B, W, GH, view(c, MG_BF_ALT), view(c), d_cov, d_cursor, row0, d_evaluated);
        }
        if (W.order) { // what is left, by length (grids sized for the round's worst case: workgroups beyond the chains written find nothing)
            const unsigned og = (unsigned)std::min<u64>(nblocks((u64)W.comb_cap), (u64)R.cus * 4);
            hipLaunchKernelGGL(fw_order_kernel<0>, dim3(og), dim3(TPB), 0, c->stream, W);
            hipLaunchKernelGGL(fw_order_kernel<1>, dim3(og), dim3(TPB), 0, c->stream, W);
        }
        int G = 2; // lanes per chain: the samples, rounded up to a power of two
        while (G < 64 && (u32)G < R.B.n_samples) G *= 2;
        // the counting pass of `index` leaves a margin in the item buffer: the insert pass packs its chunks in another order
        const u32 cap_eff = MODE == 1 ? W.item_cap / 8 * 7 : W.item_cap;
        if (c->use_chain_kernel && c->k >= 17 && c->k <= MG_MAX_PACKED_K) { // picks and evaluation in one kernel; what it lists, the pair below takes
            hipLaunchKernelGGL(fw_chain_kernel<MODE>, dim3(R.cus * 6), dim3(TPB), 0, c->stream, R.B, W, G, view(c, MG_BF_ALT), view(c), d_cov, d_overflow, d_cursor, row0,
                               d_evaluated);
            hipLaunchKernelGGL(fw_picks_kernel<true>, dim3(R.cus * 4), dim3(TPB), 0, c->stream, R.B, W, 64, cap_eff);
        } else {
            hipLaunchKernelGGL(fw_picks_kernel<false>, dim3(R.cus * 6), dim3(TPB), 0, c->stream, R.B, W, G, cap_eff);
            if (G < 64) hipLaunchKernelGGL(fw_picks_kernel<true>, dim3(R.cus * 4), dim3(TPB), 0, c->stream, R.B, W, 64, cap_eff);
        }
        hipLaunchKernelGGL(fw_eval_kernel<MODE>, dim3(R.cus * 8), dim3(TPB), 0, c->stream, R.B, W, view(c, MG_BF_ALT), view(c), d_cov, d_overflow, d_cursor, row0,
                           d_evaluated);
        hipLaunchKernelGGL(fw_slide_kernel<MODE>, dim3(R.cus * 8), dim3(TPB), 0, c->stream, R.B, W, view(c, MG_BF_ALT), view(c), d_cov, d_cursor, row0, d_evaluated);
        HIP_TRY(c, hipGetLastError());
    }
    return MG_OK;
}
// persistent grid of the workgroup kernel: as many workgroups as are resident together
template <int MODE> unsigned blocks_grid(mg_ctx *c)
{
    int &g = c->blocks_grid[MODE];
    if (!g) {
        int per_cu = 0, cus = 0, dev = 0;
        hipGetDevice(&dev);
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 256;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cover_blocks_kernel<MODE>, TPB, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        g = cus * per_cu;
    }
    return (unsigned)g;
}
// tier 3: what tier 2 handed on, compacted, through the workgroup kernel
template <int MODE> int blocks_tier3(BlocksRun &R, bool compact, u32 *d_cov, u8 *d_overflow, unsigned long long *d_cursor, u32 row0, unsigned long long *d_evaluated)
{
    mg_ctx *c = R.c;
    unsigned long long *fb_count = c->d_gen_count + 4;
    if (compact) { // (the insert pass of `index` walks the counting pass's list)
        HIP_TRY(c, hipMemsetAsync(fb_count, 0, 8, c->stream));
        hipLaunchKernelGGL(fb_compact_kernel, dim3(R.cus * 8), dim3(TPB), 0, c->stream, (const u32 *)R.gen_list, (const unsigned long long *)c->d_gen_count,
                           (const u8 *)R.fb_flag, R.fb_list, fb_count);
    }
    hipLaunchKernelGGL(cover_blocks_kernel<MODE>, dim3(blocks_grid<MODE>(c)), dim3(TPB), 0, c->stream, R.B, (const u32 *)R.fb_list, (const unsigned long long *)fb_count,
                       view(c, MG_BF_ALT), view(c), d_cov, d_overflow, IndexEmit{nullptr, d_cursor, row0}, d_evaluated);
    HIP_TRY(c, hipGetLastError());
    return MG_OK;
}
} // namespace

// block cutting for a batch of kept records in file order (main.cpp:341, 547; var_block.hpp:77-80), on the device
MG_EXPORT int mg_cut_blocks_device(mg_ctx *c, const mg_panel_dev *p, void *d_blk_var_off_out, void *d_var_block_out, void *d_n_blocks_out)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    TRY(check_panel(c, p, false));
    if (!d_n_blocks_out) return fail(c, MG_ERR_ARG, "NULL argument");
    if (p->n_vars == 0) {
        HIP_TRY(c, hipMemsetAsync(d_n_blocks_out, 0, 8, c->stream));
        return MG_OK;
    }
    if (!d_blk_var_off_out) return fail(c, MG_ERR_ARG, "NULL argument");
    const u64 n = p->n_vars;
    void *d_cut, *d_ts;
    TRY(scratch(c, c->s_blk[9], n, &d_cut));
    TRY(scratch(c, c->s_blk[10], 4 * (u64)nblocks(n) + 4, &d_ts));
    hipLaunchKernelGGL(cut_flags_kernel, dim3(nblocks(n)), dim3(TPB), 0, c->stream, n, (const int *)p->pos, (const u32 *)p->ref_size, (const u32 *)p->min_size,
                       (const u32 *)p->contig_id, (int)c->k, (u8 *)d_cut, (u32 *)d_ts);
    return scan_flags(c, n, (const u8 *)d_cut, (u32 *)d_ts, (u32 *)d_blk_var_off_out, (u32 *)d_var_block_out, (unsigned long long *)d_n_blocks_out);
}

MG_EXPORT int mg_cut_blocks(mg_ctx *c, size_t n_vars, const int32_t *pos, const uint32_t *ref_size, const uint32_t *min_size, const uint32_t *contig_id,
                            uint32_t *blk_var_off_out, size_t *n_blocks_out)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    if (!n_blocks_out) return fail(c, MG_ERR_ARG, "NULL argument");
    *n_blocks_out = 0;
    if (n_vars == 0) return MG_OK;
    if (!pos || !ref_size || !min_size || !contig_id || !blk_var_off_out) return fail(c, MG_ERR_ARG, "NULL argument");
    if (n_vars >= 0xFFFFFFFFull) return fail(c, MG_ERR_LIMIT, "mg_cut_blocks takes fewer than 2^32 records per batch");
    void *d_pos, *d_rs, *d_ms, *d_cid, *d_off;
    TRY(upload(c, c->s_misc[0], pos, 4 * n_vars, &d_pos));
    TRY(upload(c, c->s_misc[1], ref_size, 4 * n_vars, &d_rs));
    TRY(upload(c, c->s_misc[2], min_size, 4 * n_vars, &d_ms));
    TRY(upload(c, c->s_misc[3], contig_id, 4 * n_vars, &d_cid));
    TRY(scratch(c, c->s_out, 4 * (n_vars + 1), &d_off));
    if (!c->d_gen_count) HIP_TRY(c, hipMalloc(&c->d_gen_count, 64));
    mg_panel_dev p{};
    p.n_vars = n_vars;
    p.pos = (const int32_t *)d_pos; p.ref_size = (const uint32_t *)d_rs; p.min_size = (const uint32_t *)d_ms; p.contig_id = (const uint32_t *)d_cid;
    TRY(mg_cut_blocks_device(c, &p, d_off, nullptr, c->d_gen_count + 1));
    unsigned long long nb = 0;
    HIP_TRY(c, hipMemcpyAsync(&nb, c->d_gen_count + 1, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(blk_var_off_out, d_off, 4 * (nb + 1), hipMemcpyDeviceToHost));
    *n_blocks_out = (size_t)nb;
    return MG_OK;
}

// VB::extract_kmers + set_coverages (main.cpp:556-557) for every block of a resident panel
MG_EXPORT int mg_cover_blocks_device(mg_ctx *c, const mg_panel_dev *p, const void *d_blk_var_off, const void *d_var_block, const void *d_n_blocks, int haploid,
                                     void *d_cov_out, void *d_overflow_out)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c) return MG_ERR_ARG;
    TRY(check_panel(c, p, true));
    if (p->n_vars == 0) return MG_OK;
    if (!d_blk_var_off || !d_n_blocks || !d_cov_out || !d_overflow_out) return fail(c, MG_ERR_ARG, "NULL argument");
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    if (!c->map.slots) TRY(map_reserve(c, 0));
    for (auto &e : c->ev_b)
        if (!e) HIP_TRY(c, hipEventCreate(&e));
    c->blocks_stats_valid = false;
    HIP_TRY(c, hipEventRecord(c->ev_b[0], c->stream));
    BlocksRun R{};
    TRY(blocks_setup(c, p, (const u32 *)d_blk_var_off, (const u32 *)d_var_block, (const unsigned long long *)d_n_blocks, haploid, &R));
    const u64 n = p->n_vars;
    HIP_TRY(c, hipMemsetAsync(d_overflow_out, 0, n, c->stream));
    HIP_TRY(c, hipMemsetAsync(R.fb_flag, 0, n, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_gen_count, 0, 64, c->stream));
    // tier 1: lone and short records (classification fused in); everything else is listed
    u32 *need_slow = (u32 *)(c->d_hit_count + 3);
    if (++c->iso_call_no == 0) c->iso_call_no = 1;
    hipLaunchKernelGGL(panel_lone_kernel<false>, dim3((unsigned)((2 * n + (u64)LONE_TILES * TPB - 1) / ((u64)LONE_TILES * TPB))), dim3(TPB), 0, c->stream, R.P, n, R.B.blk_var_off, R.B.var_block, (const u8 *)c->d_ref,
                       (const u64 *)c->d_ref2, (const u32 *)c->d_refbad, (const u8 *)p->pool, (int)c->k, haploid, view(c, MG_BF_ALT), view(c), (u32 *)d_cov_out, need_slow,
                       c->iso_call_no, R.gen_list, c->d_gen_count, (unsigned short *)R.B.rec_class);
    hipLaunchKernelGGL(panel_lone_kernel<true>, dim3((unsigned)((2 * n + (u64)LONE_TILES * TPB - 1) / ((u64)LONE_TILES * TPB))), dim3(TPB), 0, c->stream, R.P, n, R.B.blk_var_off, R.B.var_block, (const u8 *)c->d_ref,
                       (const u64 *)c->d_ref2, (const u32 *)c->d_refbad, (const u8 *)p->pool, (int)c->k, haploid, view(c, MG_BF_ALT), view(c), (u32 *)d_cov_out, need_slow,
                       c->iso_call_no, R.gen_list, c->d_gen_count, (unsigned short *)R.B.rec_class);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_b[1], c->stream));
    TRY(blocks_tier2<0>(R, (u32 *)d_cov_out, (u8 *)d_overflow_out, nullptr, 0u, c->d_gen_count + 3));
    HIP_TRY(c, hipEventRecord(c->ev_b[2], c->stream));
    TRY(blocks_tier3<0>(R, true, (u32 *)d_cov_out, (u8 *)d_overflow_out, nullptr, 0u, c->d_gen_count + 3));
    hipLaunchKernelGGL(fw_finish_kernel, dim3(R.cus * 8), dim3(TPB), 0, c->stream, R.B, (const u32 *)R.gen_list, (const unsigned long long *)c->d_gen_count,
                       (const u8 *)d_overflow_out, (u32 *)d_cov_out);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_b[3], c->stream));
    c->blocks_stats_valid = true;
    return MG_OK;
}

// timing and counts of the most recent mg_cover_blocks_device (waits for it)
MG_EXPORT int mg_blocks_stats(mg_ctx *c, float *ms_out, uint64_t *counts_out)
{
    const DeviceGuard on_device(c, LAZY);
    if (!c || !ms_out || !counts_out) return MG_ERR_ARG;
    if (!c->blocks_stats_valid) return fail(c, MG_ERR_STATE, "no mg_cover_blocks_device yet");
    HIP_TRY(c, hipEventSynchronize(c->ev_b[3]));
    for (int i = 0; i < 3; ++i) HIP_TRY(c, hipEventElapsedTime(&ms_out[i], c->ev_b[i], c->ev_b[i + 1]));
    unsigned long long h[8];
    HIP_TRY(c, hipMemcpy(h, c->d_gen_count, 64, hipMemcpyDeviceToHost));
    counts_out[0] = h[0];
    counts_out[1] = h[2];
    counts_out[2] = h[3];
    counts_out[3] = h[4];
    return MG_OK;
}

namespace {
// the host forms' panel: every block its own "sequence" (blk_ref_base / blk_ref_len as the caller gives them)
int upload_blocks(mg_ctx *c, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len, const uint32_t *blk_var_off, size_t n_vars,
                  const int32_t *pos, const uint32_t *ref_size, const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                  const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon, const uint16_t *gt, uint32_t n_samples,
                  mg_panel_dev *out, void **d_blk_var_off, void **d_var_block, void **d_n_blocks, const uint32_t *sp_off = nullptr,
                  const uint32_t *sp_sample = nullptr, const uint16_t *sp_gt = nullptr)
{
    if (!blk_ref_base || !blk_ref_len || !blk_var_off || !pos || !ref_size || !min_size || !present || !var_allele_off || !allele_off || !pool || !canon ||
        (n_samples && !gt && !sp_off) || (sp_off && (!sp_sample || !sp_gt)))
        return fail(c, MG_ERR_ARG, "NULL argument");
    if (sp_off)
        for (size_t v = 0; v < n_vars; ++v) {
            if (sp_off[v + 1] < sp_off[v] || sp_off[v + 1] - sp_off[v] > n_samples) return fail(c, MG_ERR_ARG, "sparse genotype offsets of record %zu", v);
            for (u32 e = sp_off[v]; e < sp_off[v + 1]; ++e)
                if (sp_sample[e] >= n_samples || (e > sp_off[v] && sp_sample[e] <= sp_sample[e - 1]))
                    return fail(c, MG_ERR_ARG, "sparse genotypes of record %zu: samples must ascend below n_samples", v);
        }
    if (!c->d_ref) return fail(c, MG_ERR_STATE, "mg_reference_upload first");
    if (blk_var_off[n_blocks] != n_vars) return fail(c, MG_ERR_ARG, "block offsets do not close");
    const size_t na = var_allele_off[n_vars];
    if (allele_off[na] > pool_len) return fail(c, MG_ERR_ARG, "allele offsets exceed the pool");
    std::vector<u32> var_block(n_vars);
    for (size_t b = 0; b < n_blocks; ++b) {
        if (blk_ref_base[b] + blk_ref_len[b] > c->ref_len) return fail(c, MG_ERR_ARG, "block %zu lies outside the uploaded reference", b);
        for (u32 v = blk_var_off[b]; v < blk_var_off[b + 1]; ++v) var_block[v] = (u32)b;
    }
    void *d[14];
    TRY(upload(c, c->s_misc[0], blk_ref_base, 8 * n_blocks, &d[0]));
    TRY(upload(c, c->s_misc[1], blk_ref_len, 4 * n_blocks, &d[1]));
    TRY(upload(c, c->s_misc[2], blk_var_off, 4 * (n_blocks + 1), &d[2]));
    TRY(upload(c, c->s_misc[3], var_block.data(), 4 * n_vars, &d[3]));
    TRY(upload(c, c->s_misc[4], pos, 4 * n_vars, &d[4]));
    TRY(upload(c, c->s_misc[5], ref_size, 4 * n_vars, &d[5]));
    TRY(upload(c, c->s_misc[6], min_size, 4 * n_vars, &d[6]));
    TRY(upload(c, c->s_misc[7], present, n_vars, &d[7]));
    TRY(upload(c, c->s_rows, var_allele_off, 4 * (n_vars + 1), &d[8]));
    TRY(upload(c, c->s_aux, allele_off, 4 * (na + 1), &d[9]));
    TRY(upload(c, c->s_open[0], pool, pool_len, &d[10]));
    TRY(upload(c, c->s_open[1], canon, na, &d[11]));
    void *d_sp[3] = {nullptr, nullptr, nullptr};
    if (sp_off) {
        const size_t ne = sp_off[n_vars];
        TRY(upload(c, c->s_open[2], sp_off, 4 * (n_vars + 1), &d_sp[0]));
        TRY(upload(c, c->s_hit[0], sp_sample, 4 * ne, &d_sp[1]));
        TRY(upload(c, c->s_hit[1], sp_gt, 2 * ne, &d_sp[2]));
        d[12] = nullptr;
    } else
        TRY(upload(c, c->s_open[2], gt, 2 * (size_t)n_vars * n_samples, &d[12]));
    if (!c->d_gen_count) HIP_TRY(c, hipMalloc(&c->d_gen_count, 64));
    const unsigned long long nb = n_blocks;
    HIP_TRY(c, hipMemcpyAsync(c->d_gen_count + 1, &nb, 8, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // (`nb` and `var_block` live on this frame)
    mg_panel_dev p{};
    p.n_vars = n_vars;
    p.n_contigs = (uint32_t)n_blocks;
    p.contig_base = (const uint64_t *)d[0]; p.contig_len = (const uint32_t *)d[1]; p.contig_id = (const uint32_t *)d[3];
    p.pos = (const int32_t *)d[4]; p.ref_size = (const uint32_t *)d[5]; p.min_size = (const uint32_t *)d[6]; p.present = (const uint8_t *)d[7];
    p.var_allele_off = (const uint32_t *)d[8]; p.allele_off = (const uint32_t *)d[9]; p.pool = (const char *)d[10]; p.pool_bytes = pool_len; p.canon = (const uint8_t *)d[11];
    p.gt = (const uint16_t *)d[12]; p.n_samples = n_samples;
    p.sp_off = (const uint32_t *)d_sp[0]; p.sp_sample = (const uint32_t *)d_sp[1]; p.sp_gt = (const uint16_t *)d_sp[2];
    *out = p;
    *d_blk_var_off = d[2];
    *d_var_block = d[3];
    *d_n_blocks = c->d_gen_count + 1;
    return MG_OK;
}
} // namespace

namespace {
int cover_blocks_host(mg_ctx *c, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len, const uint32_t *blk_var_off, size_t n_vars,
                      const int32_t *pos, const uint32_t *ref_size, const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                      const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon, const uint16_t *gt, const uint32_t *sp_off,
                      const uint32_t *sp_sample, const uint16_t *sp_gt, uint16_t sp_default, uint32_t n_samples, int haploid, uint32_t *cov_out, uint8_t *overflow_out);
int index_blocks_host(mg_ctx *c, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len, const uint32_t *blk_var_off, size_t n_vars,
                      const int32_t *pos, const uint32_t *ref_size, const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                      const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon, const uint16_t *gt, const uint32_t *sp_off,
                      const uint32_t *sp_sample, const uint16_t *sp_gt, uint16_t sp_default, uint32_t n_samples, int haploid, uint8_t *overflow_out);
} // namespace

MG_EXPORT int mg_cover_blocks(mg_ctx *c, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len,
                              const uint32_t *blk_var_off, size_t n_vars, const int32_t *pos, const uint32_t *ref_size,
                              const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                              const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon, const uint16_t *gt,
                              uint32_t n_samples, int haploid, uint32_t *cov_out, uint8_t *overflow_out)
{
    return cover_blocks_host(c, n_blocks, blk_ref_base, blk_ref_len, blk_var_off, n_vars, pos, ref_size, min_size, present, var_allele_off, allele_off, pool, pool_len,
                             canon, gt, nullptr, nullptr, nullptr, 0, n_samples, haploid, cov_out, overflow_out);
}
MG_EXPORT int mg_cover_blocks_sparse(mg_ctx *c, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len, const uint32_t *blk_var_off,
                                     size_t n_vars, const int32_t *pos, const uint32_t *ref_size, const uint32_t *min_size, const uint8_t *present,
                                     const uint32_t *var_allele_off, const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon,
                                     const uint32_t *sp_off, const uint32_t *sp_sample, const uint16_t *sp_gt, uint16_t sp_default, uint32_t n_samples, int haploid,
                                     uint32_t *cov_out, uint8_t *overflow_out)
{
    if (c && !sp_off) return fail(c, MG_ERR_ARG, "NULL argument");
    return cover_blocks_host(c, n_blocks, blk_ref_base, blk_ref_len, blk_var_off, n_vars, pos, ref_size, min_size, present, var_allele_off, allele_off, pool, pool_len,
                             canon, nullptr, sp_off, sp_sample, sp_gt, sp_default, n_samples, haploid, cov_out, overflow_out);
}
MG_EXPORT int mg_index_blocks_sparse(mg_ctx *c, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len, const uint32_t *blk_var_off,
                                     size_t n_vars, const int32_t *pos, const uint32_t *ref_size, const uint32_t *min_size, const uint8_t *present,
                                     const uint32_t *var_allele_off, const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon,
                                     const uint32_t *sp_off, const uint32_t *sp_sample, const uint16_t *sp_gt, uint16_t sp_default, uint32_t n_samples, int haploid,
                                     uint8_t *overflow_out)
{
    if (c && !sp_off) return fail(c, MG_ERR_ARG, "NULL argument");
    return index_blocks_host(c, n_blocks, blk_ref_base, blk_ref_len, blk_var_off, n_vars, pos, ref_size, min_size, present, var_allele_off, allele_off, pool, pool_len,
                             canon, nullptr, sp_off, sp_sample, sp_gt, sp_default, n_samples, haploid, overflow_out);
}

namespace {
int cover_blocks_host(mg_ctx *c, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len, const uint32_t *blk_var_off, size_t n_vars,
                      const int32_t *pos, const uint32_t *ref_size, const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                      const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon, const uint16_t *gt, const uint32_t *sp_off,
                      const uint32_t *sp_sample, const uint16_t *sp_gt, uint16_t sp_default, uint32_t n_samples, int haploid, uint32_t *cov_out, uint8_t *overflow_out)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    if (n_vars == 0) return MG_OK;
    if (!cov_out || !overflow_out) return fail(c, MG_ERR_ARG, "NULL argument");
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    mg_panel_dev p{};
    void *d_bo, *d_vb, *d_nb;
    TRY(upload_blocks(c, n_blocks, blk_ref_base, blk_ref_len, blk_var_off, n_vars, pos, ref_size, min_size, present, var_allele_off, allele_off, pool, pool_len,
                      canon, gt, n_samples, &p, &d_bo, &d_vb, &d_nb, sp_off, sp_sample, sp_gt));
    p.sp_default = sp_default;
    const size_t na = var_allele_off[n_vars];
    void *d_cov, *d_ovf;
    TRY(scratch(c, c->s_out, 4 * na, &d_cov));
    TRY(scratch(c, c->s_irr, n_vars, &d_ovf));
    TRY(mg_cover_blocks_device(c, &p, d_bo, d_vb, d_nb, haploid, d_cov, d_ovf));
    HIP_TRY(c, hipMemcpyAsync(cov_out, d_cov, 4 * na, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(overflow_out, d_ovf, n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}
} // namespace

// index time: VB::extract_kmers + add_kmers_to_bf (main.cpp:349-350, 122-144) for every block of a resident panel.  Synchronises
// once on eight bytes: the exact map is sized from the counting pass before the insert pass runs.
MG_EXPORT int mg_index_blocks_device(mg_ctx *c, const mg_panel_dev *p, const void *d_blk_var_off, const void *d_var_block, const void *d_n_blocks, int haploid,
                                     void *d_overflow_out)
{
    const DeviceGuard on_device(c);
    if (!c) return MG_ERR_ARG;
    TRY(check_panel(c, p, true));
    if (p->n_vars == 0) return MG_OK;
    if (!d_blk_var_off || !d_n_blocks || !d_overflow_out) return fail(c, MG_ERR_ARG, "NULL argument");
    if (c->bf[MG_BF_ALT].mode) return fail(c, MG_ERR_STATE, "mg_index_blocks after mg_bf_finalize");
    const u64 n = p->n_vars;
    if (c->map.rows_total + n >= 0xFFFFFFFFULL) return fail(c, MG_ERR_LIMIT, "exact map: more than 2^32-1 insertion rows");
    TRY(map_reserve(c, n)); // the lone records' REF keys take insertion rows rows_total + v; may grow and re-hash the table: before the kernels, never under one
    BlocksRun R{};
    TRY(blocks_setup(c, p, (const u32 *)d_blk_var_off, (const u32 *)d_var_block, (const unsigned long long *)d_n_blocks, haploid, &R));
    c->gate_dirty = true;
    HIP_TRY(c, hipMemsetAsync(d_overflow_out, 0, n, c->stream));
    HIP_TRY(c, hipMemsetAsync(R.fb_flag, 0, n, c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_gen_count, 0, 64, c->stream));
    // tier 1: lone and short records are inserted at once; everything else is listed
    hipLaunchKernelGGL(panel_lone_index_kernel, dim3((unsigned)((n + (u64)LONE_TILES * TPB - 1) / ((u64)LONE_TILES * TPB))), dim3(TPB), 0, c->stream, R.P, n, R.B.blk_var_off, R.B.var_block, (const u8 *)c->d_ref, (const u8 *)p->pool,
                       (int)c->k, haploid, view(c, MG_BF_ALT), view(c), (u32)c->map.rows_total, (u8 *)d_overflow_out, R.gen_list, c->d_gen_count, (unsigned short *)R.B.rec_class);
    HIP_TRY(c, hipGetLastError());
    c->map.rows_total += n;
    unsigned long long *d_cursor = c->d_gen_count + 1;
    // pass 1: how many exact-map insertion rows the other records need; which of them go to tier 3, which to the host
    TRY(blocks_tier2<1>(R, nullptr, (u8 *)d_overflow_out, d_cursor, 0u, nullptr));
    TRY(blocks_tier3<1>(R, true, nullptr, (u8 *)d_overflow_out, d_cursor, 0u, nullptr));
    unsigned long long rows = 0;
    HIP_TRY(c, hipMemcpyAsync(&rows, d_cursor, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (rows) TRY(map_reserve(c, rows)); // between the passes
    // pass 2: insert.  Every REF k-mer takes the next insertion row of the batch (ids are then whatever order the
    // device reached them in: the index FILE, which every GPU of a call loads alike, is what fixes the counter layout).
    // The records keep the tier pass 1 gave them (its flags stand), so the rows counted are the rows used.
    HIP_TRY(c, hipMemsetAsync(d_cursor, 0, 8, c->stream));
    TRY(blocks_tier2<2>(R, nullptr, (u8 *)d_overflow_out, d_cursor, (u32)c->map.rows_total, nullptr));
    TRY(blocks_tier3<2>(R, false, nullptr, (u8 *)d_overflow_out, d_cursor, (u32)c->map.rows_total, nullptr));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->map.rows_total += rows;
    return MG_OK;
}

MG_EXPORT int mg_index_blocks(mg_ctx *c, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len,
                              const uint32_t *blk_var_off, size_t n_vars, const int32_t *pos, const uint32_t *ref_size,
                              const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                              const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon, const uint16_t *gt,
                              uint32_t n_samples, int haploid, uint8_t *overflow_out)
{
    return index_blocks_host(c, n_blocks, blk_ref_base, blk_ref_len, blk_var_off, n_vars, pos, ref_size, min_size, present, var_allele_off, allele_off, pool, pool_len,
                             canon, gt, nullptr, nullptr, nullptr, 0, n_samples, haploid, overflow_out);
}
namespace {
int index_blocks_host(mg_ctx *c, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len, const uint32_t *blk_var_off, size_t n_vars,
                      const int32_t *pos, const uint32_t *ref_size, const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                      const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon, const uint16_t *gt, const uint32_t *sp_off,
                      const uint32_t *sp_sample, const uint16_t *sp_gt, uint16_t sp_default, uint32_t n_samples, int haploid, uint8_t *overflow_out)
{
    const DeviceGuard on_device(c);
    if (!c) return MG_ERR_ARG;
    if (n_vars == 0) return MG_OK;
    if (!overflow_out) return fail(c, MG_ERR_ARG, "NULL argument");
    if (c->bf[MG_BF_ALT].mode) return fail(c, MG_ERR_STATE, "mg_index_blocks after mg_bf_finalize");
    mg_panel_dev p{};
    void *d_bo, *d_vb, *d_nb;
    TRY(upload_blocks(c, n_blocks, blk_ref_base, blk_ref_len, blk_var_off, n_vars, pos, ref_size, min_size, present, var_allele_off, allele_off, pool, pool_len,
                      canon, gt, n_samples, &p, &d_bo, &d_vb, &d_nb, sp_off, sp_sample, sp_gt));
    p.sp_default = sp_default;
    void *d_ovf;
    TRY(scratch(c, c->s_irr, n_vars, &d_ovf));
    HIP_TRY(c, hipMemsetAsync(d_ovf, 0, n_vars, c->stream));
    TRY(mg_index_blocks_device(c, &p, d_bo, d_vb, d_nb, haploid, d_ovf));
    HIP_TRY(c, hipMemcpy(overflow_out, d_ovf, n_vars, hipMemcpyDeviceToHost));
    return MG_OK;
}
} // namespace

// index time: blocks of one variant whose alleles are all shorter than k (nearly every block of a SNP panel), on the device
MG_EXPORT int mg_index_isolated(mg_ctx *c, size_t n_vars, const uint64_t *pos, const uint32_t *var_allele_off, const uint32_t *allele_off,
                                const char *allele_pool, size_t pool_len, const uint64_t *present_mask, const uint8_t *flags, uint8_t *overflow_out)
{
    const DeviceGuard on_device(c);
    if (!c) return MG_ERR_ARG;
    if (n_vars == 0) return MG_OK;
    if (!pos || !var_allele_off || !allele_off || !allele_pool || !present_mask || !flags || !overflow_out) return fail(c, MG_ERR_ARG, "NULL argument");
    if (!c->d_ref) return fail(c, MG_ERR_STATE, "mg_reference_upload first");
    if (c->bf[MG_BF_ALT].mode) return fail(c, MG_ERR_STATE, "mg_index_isolated after mg_bf_finalize");
    if (c->map.rows_total + n_vars >= 0xFFFFFFFFULL) return fail(c, MG_ERR_LIMIT, "exact map: more than 2^32-1 insertion rows");
    const size_t na = var_allele_off[n_vars];
    for (size_t v = 0; v < n_vars; ++v) // every window the kernel will read lies inside the uploaded reference (as mg_call_isolated checks)
        if (flags[v] & 1) {
            const u32 a0 = var_allele_off[v];
            const u64 rs = allele_off[a0 + 1] - allele_off[a0];
            if (pos[v] < c->k / 2 || pos[v] + rs + (c->k + 1) / 2 > c->ref_len)
                return fail(c, MG_ERR_ARG, "variant %zu flagged eligible but its flanks leave the uploaded reference", v);
        }
    if (allele_off[na] > pool_len) return fail(c, MG_ERR_ARG, "allele offsets exceed the pool");
    TRY(map_reserve(c, n_vars)); // may grow and re-hash the table: before the kernel, never under it
    void *d_pos, *d_vo, *d_ao, *d_pool, *d_pm, *d_fl, *d_ovf;
    TRY(upload(c, c->s_rows, pos, 8 * n_vars, &d_pos));
    TRY(upload(c, c->s_aux, var_allele_off, 4 * (n_vars + 1), &d_vo));
    TRY(upload(c, c->s_misc[0], allele_off, 4 * (na + 1), &d_ao));
    TRY(upload(c, c->s_misc[1], allele_pool, pool_len, &d_pool));
    TRY(upload(c, c->s_misc[3], present_mask, 8 * n_vars, &d_pm));
    TRY(upload(c, c->s_misc[4], flags, n_vars, &d_fl));
    TRY(scratch(c, c->s_irr, n_vars, &d_ovf));
    c->gate_dirty = true;
    hipLaunchKernelGGL(iso_index_kernel, dim3(nblocks(n_vars)), dim3(TPB), 0, c->stream, (const u8 *)c->d_ref, (u64)n_vars, (const u64 *)d_pos, (const u32 *)d_vo,
                       (const u32 *)d_ao, (const u8 *)d_pool, (const u64 *)d_pm, (const u8 *)d_fl, (int)c->k, view(c, MG_BF_ALT), view(c), (u32)c->map.rows_total,
                       (u8 *)d_ovf);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(overflow_out, d_ovf, n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->map.rows_total += n_vars;
    return MG_OK;
}

namespace {
int reference_alloc(mg_ctx *c, size_t len)
{
    for (void *q : {(void *)c->d_ref, (void *)c->d_ref2, (void *)c->d_refbad})
        if (q) hipFree(q);
    c->d_ref = nullptr;
    c->d_ref2 = nullptr;
    c->d_refbad = nullptr;
    c->ref_len = 0;
    const size_t padded = (len + 63) / 64 * 64 + 256; // the pack kernel reads whole 32-byte groups, pack_span whole dwords around a window
    const u64 n_words = (len + 31) / 32 + 4;
    HIP_TRY(c, hipMalloc(&c->d_ref, padded));
    HIP_TRY(c, hipMalloc(&c->d_ref2, n_words * 8));
    HIP_TRY(c, hipMalloc(&c->d_refbad, n_words * 4));
    HIP_TRY(c, hipMemsetAsync(c->d_ref, 0, padded, c->stream));
    return MG_OK;
}
} // namespace

MG_EXPORT int mg_reference_upload(mg_ctx *c, const char *ascii, size_t len)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c || (len && !ascii)) return MG_ERR_ARG;
    TRY(reference_alloc(c, len));
    if (len) HIP_TRY(c, hipMemcpyAsync(c->d_ref, ascii, len, hipMemcpyHostToDevice, c->stream));
    TRY(reference_pack(c, c->d_ref, len, c->d_ref2, c->d_refbad));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->ref_len = len;
    return MG_OK;
}
// the same from a buffer already on the device (a copy is kept: the caller's buffer may go)
MG_EXPORT int mg_reference_upload_device(mg_ctx *c, const void *d_ascii, size_t len)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c || (len && !d_ascii)) return MG_ERR_ARG;
    TRY(reference_alloc(c, len));
    if (len) HIP_TRY(c, hipMemcpyAsync(c->d_ref, d_ascii, len, hipMemcpyDeviceToDevice, c->stream));
    TRY(reference_pack(c, c->d_ref, len, c->d_ref2, c->d_refbad));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->ref_len = len;
    return MG_OK;
}

MG_EXPORT int mg_call_isolated_device(mg_ctx *c, size_t n_vars, const void *d_pos, const void *d_var_allele_off,
                                      const void *d_allele_off, const void *d_allele_pool, const void *d_freq,
                                      const void *d_present_mask, const void *d_flags, float error_rate, int max_cov, int haploid,
                                      void *d_cov_out, void *d_gt1, void *d_gt2, void *d_gq, void *d_status, void *d_probs,
                                      const void *d_var_gt_off)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    if (n_vars == 0) return MG_OK;
    if (!c->d_ref) return fail(c, MG_ERR_STATE, "mg_reference_upload first");
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    if (!c->map.slots) TRY(map_reserve(c, 0));
    GenoParams p;
    TRY(fill_geno_params(c, error_rate, max_cov, haploid, &p));
    u32 *need_slow = (u32 *)(c->d_hit_count + 3); // a spare word of the scan's counter block, compared with a call number
    if (++c->iso_call_no == 0) c->iso_call_no = 1;  // (never 0: the scan clears the block) so that nothing has to reset it
    hipLaunchKernelGGL(iso_cover_kernel<false>, dim3(nblocks(2 * (u64)n_vars)), dim3(TPB), 0, c->stream, (const u8 *)c->d_ref, (const u64 *)c->d_ref2, (const u32 *)c->d_refbad, (u64)n_vars,
                       (const u64 *)d_pos, (const u32 *)d_var_allele_off, (const u32 *)d_allele_off, (const u8 *)d_allele_pool,
                       (const u64 *)d_present_mask, (const u8 *)d_flags, (int)c->k, view(c, MG_BF_ALT), view(c), (u32 *)d_cov_out, need_slow,
                       c->iso_call_no);
    hipLaunchKernelGGL(iso_cover_kernel<true>, dim3(nblocks(2 * (u64)n_vars)), dim3(TPB), 0, c->stream, (const u8 *)c->d_ref, (const u64 *)c->d_ref2, (const u32 *)c->d_refbad, (u64)n_vars,
                       (const u64 *)d_pos, (const u32 *)d_var_allele_off, (const u32 *)d_allele_off, (const u8 *)d_allele_pool,
                       (const u64 *)d_present_mask, (const u8 *)d_flags, (int)c->k, view(c, MG_BF_ALT), view(c), (u32 *)d_cov_out, need_slow,
                       c->iso_call_no);
    hipLaunchKernelGGL(iso_genotype_kernel, dim3(nblocks(n_vars)), dim3(TPB), 0, c->stream, (u64)n_vars, (const u32 *)d_var_allele_off,
                       (const float *)d_freq, p, (const u32 *)d_cov_out, (i32 *)d_gt1, (i32 *)d_gt2, (i32 *)d_gq, (u8 *)d_status,
                       (double *)d_probs, (const u64 *)d_var_gt_off);
    HIP_TRY(c, hipGetLastError());
    return MG_OK;
}

MG_EXPORT int mg_call_isolated(mg_ctx *c, size_t n_vars, const uint64_t *pos, const uint32_t *var_allele_off,
                               const uint32_t *allele_off, const char *allele_pool, size_t pool_len, const float *freq,
                               const uint64_t *present_mask, const uint8_t *flags, float error_rate, int max_cov, int haploid,
                               uint32_t *cov_out, int32_t *gt1, int32_t *gt2, int32_t *gq, uint8_t *status, double *probs,
                               const uint64_t *var_gt_off)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    if (n_vars == 0) return MG_OK;
    if (!pos || !var_allele_off || !allele_off || !allele_pool || !freq || !present_mask || !flags || !cov_out || !gt1 || !gt2 ||
        !gq || !status)
        return fail(c, MG_ERR_ARG, "NULL argument");
    if (probs && !var_gt_off) return fail(c, MG_ERR_ARG, "probs needs var_gt_off");
    const size_t na = var_allele_off[n_vars];
    // host-side contract checks: every window the kernel will read lies inside the uploaded reference
    for (size_t v = 0; v < n_vars; ++v)
        if (flags[v] & 1) {
            const u32 a0 = var_allele_off[v];
            const u64 rs = allele_off[a0 + 1] - allele_off[a0];
            // what iso_cover_kernel reads: k/2 bases before the site, ceil(k/2) after the REF allele (the buffer's
            // 64-byte padding absorbs the aligned dwords around them)
            if (pos[v] < c->k / 2 || pos[v] + rs + (c->k + 1) / 2 > c->ref_len)
                return fail(c, MG_ERR_ARG, "variant %zu flagged eligible but its flanks leave the uploaded reference", v);
        }
    if (allele_off[na] > pool_len) return fail(c, MG_ERR_ARG, "allele offsets exceed the pool");
    void *d_pos, *d_vo, *d_ao, *d_pool, *d_fr, *d_pm, *d_fl, *d_cov, *d_g1, *d_g2, *d_gq, *d_st;
    TRY(upload(c, c->s_rows, pos, 8 * n_vars, &d_pos));
    TRY(upload(c, c->s_aux, var_allele_off, 4 * (n_vars + 1), &d_vo));
    TRY(upload(c, c->s_misc[0], allele_off, 4 * (na + 1), &d_ao));
    TRY(upload(c, c->s_misc[1], allele_pool, pool_len, &d_pool));
    TRY(upload(c, c->s_misc[2], freq, 4 * na, &d_fr));
    TRY(upload(c, c->s_misc[3], present_mask, 8 * n_vars, &d_pm));
    TRY(upload(c, c->s_misc[4], flags, n_vars, &d_fl));
    TRY(scratch(c, c->s_out, 4 * na, &d_cov));
    TRY(scratch(c, c->s_misc[5], 4 * n_vars, &d_g1));
    TRY(scratch(c, c->s_misc[6], 4 * n_vars, &d_g2));
    TRY(scratch(c, c->s_misc[7], 4 * n_vars, &d_gq));
    TRY(scratch(c, c->s_irr, n_vars, &d_st));
    // raw likelihoods are staged in a device workspace either way (one exp() per genotype instead of two)
    std::vector<u64> goff_tmp;
    if (!var_gt_off) {
        goff_tmp.resize(n_vars + 1);
        goff_tmp[0] = 0;
        for (size_t v = 0; v < n_vars; ++v) {
            const u64 A = var_allele_off[v + 1] - var_allele_off[v];
            goff_tmp[v + 1] = goff_tmp[v] + (haploid ? A : A * (A + 1) / 2);
        }
        var_gt_off = goff_tmp.data();
    }
    const size_t ng = var_gt_off[n_vars];
    void *d_pr, *d_go;
    TRY(scratch(c, c->s_open[0], 8 * (ng ? ng : 1), &d_pr));
    TRY(upload(c, c->s_open[1], var_gt_off, 8 * (n_vars + 1), &d_go));
    TRY(mg_call_isolated_device(c, n_vars, d_pos, d_vo, d_ao, d_pool, d_fr, d_pm, d_fl, error_rate, max_cov, haploid, d_cov, d_g1,
                                d_g2, d_gq, d_st, d_pr, d_go));
    if (probs && ng) HIP_TRY(c, hipMemcpyAsync(probs, d_pr, 8 * ng, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(cov_out, d_cov, 4 * na, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(gt1, d_g1, 4 * n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(gt2, d_g2, 4 * n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(gq, d_gq, 4 * n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(status, d_st, n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}

// ---- index payloads ---------------------------------------------------------------------------

// ---- panel genotypes from VCF text (gt_text_kernels.h) ------------------------------------------------------------------
MG_EXPORT int mg_decode_gt_text(mg_ctx *c, const char *text, size_t text_bytes, size_t n_records, const uint64_t *span_off, const uint32_t *span_len,
                                const int32_t *gt_index, uint32_t n_columns, const uint8_t *keep, int haploid, uint16_t *sp_default, uint32_t *sp_off,
                                uint64_t *raw_mask, uint32_t *max_allele, uint64_t *n_entries)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    c->gt_records = 0;
    c->gt_entries = 0;
    if (!sp_default || !sp_off || !n_entries) return fail(c, MG_ERR_ARG, "NULL argument");
    *n_entries = 0;
    *sp_default = (uint16_t)(1u << 14);
    sp_off[0] = 0;
    if (n_records == 0) return MG_OK;
    if (!span_off || !span_len || !gt_index || !raw_mask || !max_allele || (text_bytes && !text)) return fail(c, MG_ERR_ARG, "NULL argument");
    if (n_columns == 0) return fail(c, MG_ERR_ARG, "mg_decode_gt_text: no sample columns");
    std::vector<u32> rank(n_columns, 0xFFFFFFFFu);
    u32 n_keep = 0;
    for (u32 i = 0; i < n_columns; ++i)
        if (!keep || keep[i]) rank[i] = n_keep++;
    if (n_keep == 0) return fail(c, MG_ERR_ARG, "mg_decode_gt_text: no sample kept");
    if (n_records >= 0xFFFFFFFFull || (u64)n_records * n_keep > (1ull << 31)) return fail(c, MG_ERR_LIMIT, "mg_decode_gt_text takes at most 2^31 genotypes per batch");
    // the part of the text the spans lie in
    u64 lo = ~0ull, hi = 0;
    for (size_t r = 0; r < n_records; ++r) {
        if (span_off[r] > text_bytes || span_len[r] > text_bytes - span_off[r]) return fail(c, MG_ERR_ARG, "record %zu: sample columns outside the text", r);
        if (gt_index[r] < 0 || gt_index[r] > 1000) return fail(c, MG_ERR_ARG, "record %zu: GT index %d", r, gt_index[r]);
        if (!span_len[r]) continue;
        lo = std::min<u64>(lo, span_off[r]);
        hi = std::max<u64>(hi, span_off[r] + span_len[r]);
    }
    if (lo > hi) lo = hi = 0;
    std::vector<unsigned long long> off(n_records);
    for (size_t r = 0; r < n_records; ++r) off[r] = span_len[r] ? span_off[r] - lo : 0;
    void *d_text, *d_off, *d_len, *d_gi, *d_rank, *d_tok, *d_words, *d_stats;
    { // the text goes through a pinned staging buffer: one memcpy at memory speed, then one DMA
        const size_t nb = hi - lo;
        if (nb > c->h_gt_stage_cap) {
            if (c->h_gt_stage) hipHostFree(c->h_gt_stage);
            c->h_gt_stage = nullptr;
            c->h_gt_stage_cap = 0;
            HIP_TRY(c, hipHostMalloc(&c->h_gt_stage, nb + nb / 4 + 4096, hipHostMallocDefault));
            c->h_gt_stage_cap = nb + nb / 4 + 4096;
        }
        if (nb) memcpy(c->h_gt_stage, text + lo, nb);
        TRY(upload(c, c->s_gt[0], c->h_gt_stage, nb, &d_text));
    }
    TRY(upload(c, c->s_gt[1], off.data(), 8 * n_records, &d_off));
    TRY(upload(c, c->s_gt[2], span_len, 4 * n_records, &d_len));
    TRY(upload(c, c->s_gt[3], gt_index, 4 * n_records, &d_gi));
    TRY(upload(c, c->s_gt[4], rank.data(), 4 * (size_t)n_columns, &d_rank));
    const unsigned grid = (unsigned)std::min<u64>(n_records, 1024);
    TRY(scratch(c, c->s_gt[5], 8 * (size_t)grid * n_keep, &d_tok));
    TRY(scratch(c, c->s_gt[6], 2 * (size_t)n_records * n_keep, &d_words));
    TRY(scratch(c, c->s_gt[7], sizeof(GtStats) * n_records, &d_stats));
    hipLaunchKernelGGL(gt_decode_kernel, dim3(grid), dim3(GT_TPB), 0, c->stream, (const char *)d_text, (const unsigned long long *)d_off, (const u32 *)d_len,
                       (const i32 *)d_gi, (u32)n_records, n_columns, (const u32 *)d_rank, n_keep, haploid, (unsigned long long *)d_tok, (uint16_t *)d_words,
                       (GtStats *)d_stats);
    HIP_TRY(c, hipGetLastError());
    std::vector<GtStats> st(n_records);
    HIP_TRY(c, hipMemcpyAsync(st.data(), d_stats, sizeof(GtStats) * n_records, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream)); // (also: the host vectors uploaded above may go)
    // the default word: 0|0 phased or 0/0, whichever the batch holds more of
    u64 p0 = 0, u0 = 0;
    for (const GtStats &g : st) {
        p0 += g.n_phased0;
        u0 += g.n_unphased0;
    }
    const u32 dflt = u0 > p0 ? 0u : 1u << 14;
    u64 total = 0;
    for (size_t r = 0; r < n_records; ++r) {
        raw_mask[r] = st[r].raw_mask;
        max_allele[r] = st[r].max_allele;
        total += n_keep - (dflt ? st[r].n_phased0 : st[r].n_unphased0);
        if (total > 0xFFFFFFFFull) return fail(c, MG_ERR_LIMIT, "mg_decode_gt_text: 2^32 or more entries in one batch");
        sp_off[r + 1] = (u32)total;
    }
    *sp_default = (uint16_t)dflt;
    *n_entries = total;
    c->gt_records = (u32)n_records;
    c->gt_keep = n_keep;
    c->gt_default = dflt;
    c->gt_entries = total;
    void *d_spoff;
    TRY(upload(c, c->s_gt[8], sp_off, 4 * (n_records + 1), &d_spoff));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}
MG_EXPORT int mg_decode_gt_entries(mg_ctx *c, uint32_t *sp_sample, uint16_t *sp_gt)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    if (!c->gt_records) return fail(c, MG_ERR_STATE, "mg_decode_gt_entries follows a mg_decode_gt_text that decoded something");
    if (!c->gt_entries) return MG_OK;
    if (!sp_sample || !sp_gt) return fail(c, MG_ERR_ARG, "NULL argument");
    void *d_s, *d_g;
    TRY(scratch(c, c->s_gt[5], 4 * c->gt_entries, &d_s)); // (the token scratch is free again)
    TRY(scratch(c, c->s_gt[9], 2 * c->gt_entries, &d_g));
    hipLaunchKernelGGL(gt_compact_kernel, dim3(std::min<u32>(c->gt_records, 2048)), dim3(GT_TPB), 0, c->stream, (const uint16_t *)c->s_gt[6].p, c->gt_records, c->gt_keep,
                       c->gt_default, (const u32 *)c->s_gt[8].p, (u32 *)d_s, (uint16_t *)d_g);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(sp_sample, d_s, 4 * c->gt_entries, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(sp_gt, d_g, 2 * c->gt_entries, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}

MG_EXPORT int mg_bf_export(mg_ctx *c, int which, uint64_t *words_out, uint16_t *counts_out)
{
    const DeviceGuard on_device(c, KEEP);
    TRY(check_which(c, which));
    BFState &b = c->bf[which];
    if (words_out) HIP_TRY(c, hipMemcpy(words_out, b.words, b.nwords * 8, hipMemcpyDeviceToHost));
    if (counts_out && b.mode && b.nset) {
        void *d16;
        TRY(scratch(c, c->s_out, b.nset * 2, &d16));
        hipLaunchKernelGGL(mask_u16_kernel, dim3(nblocks(b.nset)), dim3(TPB), 0, c->stream, (const u32 *)b.counts, (uint16_t *)d16,
                           b.nset);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(counts_out, d16, b.nset * 2, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return MG_OK;
}
MG_EXPORT int mg_bf_import(mg_ctx *c, int which, int mode, uint64_t size_bits, const uint64_t *words, const uint16_t *counts,
                           uint64_t n_counts)
{
    if (c && (which == MG_BF_ALT || which == MG_BF_CTX)) c->bf[which].pos_set_valid = false;
    const DeviceGuard on_device(c);
    TRY(check_which(c, which));
    BFState &b = c->bf[which];
    if (size_bits != b.size) return fail(c, MG_ERR_ARG, "filter size %llu does not match the context (%llu)",
                                         (unsigned long long)size_bits, (unsigned long long)b.size);
    if (!words) return fail(c, MG_ERR_ARG, "words is NULL");
    HIP_TRY(c, hipMemcpy(b.words, words, b.nwords * 8, hipMemcpyHostToDevice));
    b.mode = 0;
    if (which == MG_BF_ALT) { // the gate follows the bits: rebuild it from them and from the map's keys
        TRY(alloc_gate(c));
        hipLaunchKernelGGL(gate_from_bits_kernel, dim3(nblocks(b.nwords)), dim3(TPB), 0, c->stream, view(c, MG_BF_ALT), b.nwords);
        if (c->map.slots)
            hipLaunchKernelGGL(map_gate_kernel, dim3(nblocks(1ULL << c->map.cap_log2)), dim3(TPB), 0, c->stream, view(c),
                               view(c, MG_BF_ALT));
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (mode) {
        TRY(mg_bf_finalize(c, which)); // rank is rebuilt on load, as bloom_filter.hpp:143 does
        if (n_counts != b.nset) return fail(c, MG_ERR_ARG, "counter count %llu != popcount %llu", (unsigned long long)n_counts,
                                            (unsigned long long)b.nset);
        if (n_counts) {
            void *d16;
            TRY(upload(c, c->s_out, counts, n_counts * 2, &d16));
            hipLaunchKernelGGL(widen_u16_kernel, dim3(nblocks(n_counts)), dim3(TPB), 0, c->stream, (const uint16_t *)d16, b.counts,
                               n_counts);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    return MG_OK;
}

// Sparse payloads: a filter is a few million set bits in 2^33..2^37, so the index file stores the
// ascending positions of the set bits (= counter order) instead of gigabytes of zeros.
MG_EXPORT int mg_bf_export_sparse(mg_ctx *c, int which, uint64_t *positions_out, uint16_t *counts_out)
{
    const DeviceGuard on_device(c, KEEP);
    TRY(check_which(c, which));
    BFState &b = c->bf[which];
    if (!b.mode) return fail(c, MG_ERR_STATE, "sparse export needs the filter finalised (rank directory)");
    if (positions_out && b.nset) {
        void *d;
        TRY(scratch(c, c->s_open[0], b.nset * 8, &d));
        hipLaunchKernelGGL(bit_positions_kernel, dim3(nblocks(b.nwords)), dim3(TPB), 0, c->stream, view(c, which), b.nwords, (u64 *)d);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipMemcpyAsync(positions_out, d, b.nset * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    if (counts_out) return mg_bf_export(c, which, nullptr, counts_out);
    return MG_OK;
}
MG_EXPORT int mg_bf_import_sparse(mg_ctx *c, int which, int mode, uint64_t size_bits, const uint64_t *positions,
                                  const uint16_t *counts, uint64_t n)
{
    if (c && (which == MG_BF_ALT || which == MG_BF_CTX)) c->bf[which].pos_set_valid = false;
    const DeviceGuard on_device(c);
    TRY(check_which(c, which));
    BFState &b = c->bf[which];
    if (size_bits != b.size) return fail(c, MG_ERR_ARG, "filter size %llu does not match the context (%llu)",
                                         (unsigned long long)size_bits, (unsigned long long)b.size);
    if (n && !positions) return fail(c, MG_ERR_ARG, "positions is NULL");
    HIP_TRY(c, hipMemsetAsync(b.words, 0, b.nwords * 8, c->stream));
    b.mode = 0;
    if (which == MG_BF_ALT) {
        TRY(alloc_gate(c));
        if (c->map.slots)
            hipLaunchKernelGGL(map_gate_kernel, dim3(nblocks(1ULL << c->map.cap_log2)), dim3(TPB), 0, c->stream, view(c),
                               view(c, MG_BF_ALT));
        c->gate_dirty = true;
    }
    if (n) {
        void *d;
        int *d_bad = (int *)(c->d_hit_count + 3);
        TRY(upload(c, c->s_open[0], positions, n * 8, &d));
        HIP_TRY(c, hipMemsetAsync(d_bad, 0, 4, c->stream));
        hipLaunchKernelGGL(set_bits_kernel, dim3(nblocks(n)), dim3(TPB), 0, c->stream, view(c, which), (const u64 *)d, (u64)n, b.size,
                           d_bad);
        HIP_TRY(c, hipGetLastError());
        int bad = 0;
        HIP_TRY(c, hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        if (bad) return fail(c, MG_ERR_ARG, "bit positions must be strictly ascending and below the filter size");
    }
    if (mode) {
        TRY(mg_bf_finalize(c, which));
        if (b.nset != n) return fail(c, MG_ERR_ARG, "popcount %llu != positions %llu", (unsigned long long)b.nset, (unsigned long long)n);
        if (n && counts) {
            void *d16;
            TRY(upload(c, c->s_out, counts, n * 2, &d16));
            hipLaunchKernelGGL(widen_u16_kernel, dim3(nblocks(n)), dim3(TPB), 0, c->stream, (const uint16_t *)d16, b.counts, (u64)n);
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    return MG_OK;
}

namespace {
void unpack_lform(u64 lo, u64 hi, u32 k, char *out)
{
    static const char L[4] = {'A', 'C', 'G', 'T'};
    for (u32 i = 0; i < k; ++i) out[i] = L[(i < 32 ? lo >> (2 * i) : hi >> (2 * (i - 32))) & 3];
    out[k] = 0;
}
} // namespace

// Regular keys all have length k (a shorter or longer pure-ACGT key cannot be
// told apart once packed, so mg_map_insert of a row whose length != k is irregular
// from the table's point of view only if it holds a non-ACGT byte; rows are
// expected to be k long, as every signature k-mer of the reference is).
MG_EXPORT int mg_map_export(mg_ctx *c, char *rows_out, size_t stride, int32_t *vals_out)
{
    const DeviceGuard on_device(c, KEEP);
    if (!c) return MG_ERR_ARG;
    std::vector<u64> lo, hi;
    std::vector<u32> ids;
    TRY(map_dump(c, &lo, &hi, &ids));
    if (!rows_out && !vals_out) return MG_OK;
    if (stride < c->k + 1) return fail(c, MG_ERR_ARG, "stride %zu < k+1", stride);
    std::vector<u32> vals(c->map.rows_total);
    if (c->map.rows_total) HIP_TRY(c, hipMemcpy(vals.data(), c->map.vals, c->map.rows_total * 4, hipMemcpyDeviceToHost));
    size_t j = 0;
    for (; j < lo.size(); ++j) {
        if (rows_out) {
            memset(rows_out + j * stride, 0, stride);
            unpack_lform(lo[j], hi[j], c->k, rows_out + j * stride);
        }
        if (vals_out) vals_out[j] = (int32_t)vals[ids[j]];
    }
    for (auto &kv : c->map.irregular) {
        if (rows_out) {
            memset(rows_out + j * stride, 0, stride);
            memcpy(rows_out + j * stride, kv.first.data(), kv.first.size() < stride - 1 ? kv.first.size() : stride - 1);
        }
        if (vals_out) vals_out[j] = kv.second;
        ++j;
    }
    return MG_OK;
}
MG_EXPORT int mg_map_import(mg_ctx *c, const char *rows, size_t stride, size_t n, const int32_t *vals)
{
    const DeviceGuard on_device(c);
    TRY(check_rows(c, rows, stride, n));
    if (n == 0) return MG_OK;
    TRY(mg_map_insert(c, rows, stride, n));
    if (vals) {
        // values go through a lookup of each key (a file may repeat a key, or name one that was already present:
        // the counter of a key is the one its first insertion row owns); irregular rows live in the host list
        // under the canonical form KMAP::canonical gives them (kmap.hpp:86-97)
        std::vector<u8> irr(n);
        TRY(run_rows<OP_MAP_SET>(c, 0, rows, stride, n, vals, nullptr, nullptr, 0, irr.data()));
        for (size_t i = 0; i < n; ++i)
            if (irr[i]) {
                auto it = c->map.irregular.find(host_irregular_key(rows + i * stride, stride));
                if (it != c->map.irregular.end()) it->second = vals[i];
            }
    }
    return MG_OK;
}
