/*
 * malva_hip.h -- C ABI of the MI355X-native malva-geno hot path.
 *
 * The reference (AlgoLab/malva v1.3.1) has no plugin/FFI seam: its hot path is
 * the header-only classes BF (bloom_filter.hpp:52-157), KMAP (kmap.hpp:46-132)
 * and VB (var_block.hpp:61-798), driven one k-mer / one block at a time from
 * main.cpp.  This header is that seam, cut at exactly the calls index_main and
 * call_main make, with every per-k-mer call turned into a batch call.  Each
 * entry point cites the reference interface it replaces; INTEGRATION.md shows
 * the thin BF/KMAP/VB wrappers a maintainer would put in front of it.
 *
 * Conventions
 *   - every function returns 0 (MG_OK) or a negative MG_ERR_*; the message for
 *     the last failure on a context is mg_last_error(ctx).  Nothing throws.
 *   - all device memory is owned by the opaque mg_ctx (one context = one GPU).
 *     Calls on one context must be serialised by the caller, exactly like the
 *     single-threaded reference; they may come from any host thread (each entry
 *     point makes the context's GPU current for the calling thread and puts the
 *     previous one back), and different contexts may be driven concurrently.
 *   - "rows" arguments are host buffers of n fixed-stride ASCII k-mers, each
 *     NUL-terminated inside its stride (the reference passes `const char*`
 *     and measures with strlen, bloom_filter.hpp:69); 1 <= strlen <= 128.
 *     They may be reused as soon as the call returns.
 *   - the *_device variants take device pointers valid on the context's GPU
 *     and are asynchronous on the context's stream (mg_set_stream); everything
 *     else synchronises before returning.
 *   - packed k-mer tables (the KMC stream) are SoA {hi[], lo[], cnt[]}: the
 *     ref_k-mer as a 2-bit string (A=0 C=1 G=2 T=3), MSB-first and right
 *     aligned in the 128-bit value hi:lo, so integer order == strcmp order.
 *   - there is no CPU fallback anywhere behind this interface: without a
 *     usable HIP device mg_create fails.
 */
#ifndef MALVA_HIP_H
#define MALVA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_OK 0
#define MG_ERR_ARG (-1)    /* bad argument */
#define MG_ERR_HIP (-2)    /* HIP runtime / launch failure */
#define MG_ERR_STATE (-3)  /* call not valid in the filter's current mode */
#define MG_ERR_NOMEM (-4)  /* allocation failed */
#define MG_ERR_LIMIT (-5)  /* size beyond what the implementation addresses */
#define MG_ERR_COMM (-6)   /* RCCL unavailable or a collective failed */

#define MG_BF_ALT 0 /* `bf`         main.cpp:300 -- ALT-allele signature k-mers        */
#define MG_BF_CTX 1 /* `context_bf` main.cpp:302 -- reference contexts (ref_k-mers)    */

#define MG_MAX_KMER 128 /* longest ASCII k-mer a row may hold                            */
#define MG_MAX_PACKED_K 64 /* longest k / ref_k of the packed (2-bit) paths              */

typedef struct mg_ctx mg_ctx;

/* ---- lifetime ----------------------------------------------------------- */

/* BF bf(opt::bf_size); KMAP ref_bf; BF context_bf(opt::bf_size)  (main.cpp:300-302)
 * plus VB vb(opt::k, opt::error_rate) (main.cpp:305,520).  bf_bits is the size
 * of EACH filter in bits (-b N => N * 2^33, argument_parser.hpp:119-123). */
int mg_create(mg_ctx **out, int device, uint32_t k, uint32_t ref_k, uint64_t bf_bits);
int mg_destroy(mg_ctx *ctx);
const char *mg_last_error(const mg_ctx *ctx);
/* Launch everything on `hip_stream` (a hipStream_t; NULL = the context's own
 * stream).  Lets a caller time the kernels with events on its own stream.
 * NULL never means HIP's default stream: a caller that works on the legacy
 * default stream (handle 0 in most bindings) passes MG_STREAM_DEFAULT. */
#define MG_STREAM_DEFAULT ((void *)1) /* == hipStreamLegacy */
int mg_set_stream(mg_ctx *ctx, void *hip_stream);
int mg_synchronize(mg_ctx *ctx);

/* ---- BF  (bloom_filter.hpp:52-157) -------------------------------------- */

/* void BF::add_key(const char*)                        bloom_filter.hpp:81   */
int mg_bf_insert(mg_ctx *ctx, int which, const char *rows, size_t stride, size_t n);
/* bool BF::test_key(const char*) const                 bloom_filter.hpp:87   */
int mg_bf_test(mg_ctx *ctx, int which, const char *rows, size_t stride, size_t n, uint8_t *out);
/* void BF::switch_mode()                               bloom_filter.hpp:93
 * builds the rank directory and one zeroed counter per set bit */
int mg_bf_finalize(mg_ctx *ctx, int which);
/* bool BF::increment(const char*, uint32)              bloom_filter.hpp:100
 * returns MG_ERR_STATE where the reference returns false (write mode) */
int mg_bf_increment(mg_ctx *ctx, int which, const char *rows, size_t stride, size_t n, const uint32_t *counters);
/* uint16_t BF::get_count(const char*) const            bloom_filter.hpp:115
 * (0 for every row while the filter is still in write mode) */
int mg_bf_get_count(mg_ctx *ctx, int which, const char *rows, size_t stride, size_t n, uint16_t *out);
/* _size, popcount (== _counts.size() once finalised), _mode */
int mg_bf_info(mg_ctx *ctx, int which, uint64_t *size_bits, uint64_t *n_set, int *mode);

/* ---- KMAP  (kmap.hpp:46-132) -------------------------------------------- */

/* void KMAP::add_key(const char*)                      kmap.hpp:108          */
int mg_map_insert(mg_ctx *ctx, const char *rows, size_t stride, size_t n);
/* bool KMAP::test_key(const char*)                     kmap.hpp:99           */
int mg_map_test(mg_ctx *ctx, const char *rows, size_t stride, size_t n, uint8_t *out);
/* void KMAP::increment(const char*, int)               kmap.hpp:114          */
int mg_map_increment(mg_ctx *ctx, const char *rows, size_t stride, size_t n, const int32_t *counters);
/* int KMAP::get_count(const char*)                     kmap.hpp:124          */
int mg_map_get_count(mg_ctx *ctx, const char *rows, size_t stride, size_t n, int32_t *out);
/* kmers.size() */
int mg_map_size(mg_ctx *ctx, uint64_t *n_keys);

/* ---- index-time reference scan  (main.cpp:383-401) ----------------------- */

/* One used contig, upper-cased ASCII: for every position, if the centre k-mer
 * of the ref_k window hits `bf`, add the window to `context_bf`. */
int mg_ref_scan(mg_ctx *ctx, const char *contig, size_t len);
/* the same for a contig that already lies at [offset, offset + len) of the buffer given to mg_reference_upload (the
 * reference's index_main holds every contig in memory too, main.cpp:283-295): nothing crosses PCIe, asynchronous */
int mg_ref_scan_resident(mg_ctx *ctx, uint64_t offset, size_t len);

/* ---- call-time KMC scan  (main.cpp:482-500) ------------------------------ */

/* for each (ref_k-mer, count): ref_bf.increment(centre, count);
 * if (!context_bf.test_key(ref_k-mer)) bf.increment(centre, count).
 * Both filters must be finalised.  ref_k <= MG_MAX_PACKED_K. */
int mg_kmc_scan(mg_ctx *ctx, const uint64_t *hi, const uint64_t *lo, const uint32_t *cnt, size_t n);
int mg_kmc_scan_device(mg_ctx *ctx, const void *d_hi, const void *d_lo, const void *d_cnt, size_t n);

/* The same scan over COMPACT rows: 12 bytes per row instead of 20 -- three little-endian dwords holding the 96-bit
 * value  count << (2 ref_k) | ref_k-mer (2-bit string as above).  For 33 <= ref_k <= 44 (the reference's default is 43)
 * and counts below 2^(96 - 2 ref_k) (1024 at ref_k 43; KMC caps counts at 255, MALVA:107).  mg_kmc_pack_rows_device
 * builds them from an SoA table (n rounded up to a multiple of four rows: mg_kmc_rows_bytes(n) bytes, zero padded)
 * and returns MG_ERR_LIMIT when a count does not fit; the scan is asynchronous like mg_kmc_scan_device. */
size_t mg_kmc_rows_bytes(size_t n);
int mg_kmc_pack_rows_device(mg_ctx *ctx, const void *d_hi, const void *d_lo, const void *d_cnt, size_t n, void *d_rows_out);
int mg_kmc_scan_rows_device(mg_ctx *ctx, const void *d_rows, size_t n);

/* KMC database feed: CKMCFile::OpenForListing / ReadNextKmer / CKmerAPI::to_string (main.cpp:444-449, 482-490; the KMC
 * API is a third-party library the reference links, absent from its checkout: format restated from KMC's published
 * database layout, "parity unpinned" -- DESIGN.md).  The host hands over what the two files hold and parses nothing:
 *   mg_kmc_set_lut       <db>.kmc_pre: the prefix table (every bin's 4^lut_prefix_len entries, concatenated: first
 *                        record index of each prefix), record geometry and the [min_count, max_count] listing filter
 *   mg_kmc_scan_records  a run of raw <db>.kmc_suf records (suffix bytes + little-endian counter), `first_record` =
 *                        index of the first one in the database; decoded to table rows ON THE DEVICE (prefix from the
 *                        table, suffix from the record), then scanned as mg_kmc_scan does.  10 bytes per 43-mer cross
 *                        PCIe instead of 20.  Upload and scan of consecutive pieces overlap (two staging slots);
 *                        buffers from mg_host_alloc (pinned) make the uploads asynchronous.
 *   mg_kmc_decode_records  the decoded rows themselves (tests / inspection). */
int mg_host_alloc(void **out, size_t bytes);
int mg_host_free(void *p);
int mg_kmc_set_lut(mg_ctx *ctx, const uint64_t *lut, size_t n_lut, uint32_t lut_prefix_len, uint32_t suffix_bytes,
                   uint32_t counter_bytes, uint32_t min_count, uint64_t max_count, uint64_t total_records);
int mg_kmc_scan_records(mg_ctx *ctx, const void *records, size_t n, uint64_t first_record);
int mg_kmc_decode_records(mg_ctx *ctx, const void *records, size_t n, uint64_t first_record, uint64_t *hi_out,
                          uint64_t *lo_out, uint32_t *cnt_out);

/* ---- multi-GPU exchange step --------------------------------------------- */

/* The scan's only state is two commutative wrapping-u32 sums:
 * [ bf counters (n_bf u32, in rank order) | map counters (n_map u32, in key
 * insertion order) ].  Ranks that built the same index agree on this layout,
 * so one sum all-reduce of this vector combines shard scans. */
int mg_counters_size(mg_ctx *ctx, uint64_t *n_bf, uint64_t *n_map);
int mg_counters_export_device(mg_ctx *ctx, void *d_u32_out);
int mg_counters_import_device(mg_ctx *ctx, const void *d_u32_in);
int mg_counters_reset(mg_ctx *ctx);
/* Zero-copy form: makes the two counter arrays one contiguous device allocation and returns
 * it (valid until the next insert / finalize / import).  An in-place all-reduce over
 * d_ptr[0 .. n_bf + n_map) replaces export + all-reduce + import. */
int mg_counters_view(mg_ctx *ctx, void **d_ptr, uint64_t *n_bf, uint64_t *n_map);

/* The exchange itself, inside the library: RCCL (librccl.so.1, opened on first use) over xGMI.  The reference has
 * no counterpart (single-threaded, CMakeLists.txt:46 links pthread and never uses it); SURVEY 8(b)/(e) define it.
 *
 *   one process per GPU   rank 0: mg_comm_unique_id(id); ship the 128 bytes to the other ranks (any channel);
 *                         every rank: mg_comm_init(ctx, rank, world, id); after its shard's mg_kmc_scan*:
 *                         mg_counters_allreduce(ctx) -- ncclAllReduce(sum, uint32), in place over the
 *                         mg_counters_view allocation, asynchronous on the context's stream.
 *   one process, N GPUs   mg_comm_init_all(ctxs, N) (one context per device) and, after the N shard scans,
 *                         mg_counters_allreduce_all(ctxs, N): the same all-reduces as one RCCL group.
 *                         Contexts that all sit on ONE device (rehearsing the N-way layout on a one-GPU box; RCCL
 *                         rejects duplicate devices) are summed by a kernel instead; mg_comm_info tells which.
 * Every rank must hold the same index (same inserts / same index file): the counter layout is then identical. */
#define MG_COMM_ID_BYTES 128
#define MG_COMM_NONE 0
#define MG_COMM_RCCL 1
#define MG_COMM_LOCAL 2
int mg_comm_unique_id(void *id_out /* MG_COMM_ID_BYTES */);
int mg_comm_init(mg_ctx *ctx, int rank, int world, const void *id);
int mg_comm_init_all(mg_ctx **ctxs, int n);
int mg_comm_destroy(mg_ctx *ctx);
int mg_comm_info(mg_ctx *ctx, int *rank, int *world, int *backend);
int mg_counters_allreduce(mg_ctx *ctx);
int mg_counters_allreduce_all(mg_ctx **ctxs, int n);
/* Two refinements of the exchange, both inside the library:
 *   the 16-bit packed form   two counters per 32-bit word on the links when that is exact, i.e. when no rank's partial counter
 *       exceeds 65535 / world (checked first: a max over the vector, then over the ranks; the plain 32-bit sum runs otherwise).
 *       Option exchange_pack: 0 never, 1 (default) for vectors of exchange_pack_min_mb (32) megabytes and more, 2 always.
 *       The call then waits for the guard's answer on the host; mg_get_option("exchange_packed") tells which form ran.
 *   _begin / _end   the same exchange on a stream of its own: _begin orders it behind what the context's stream holds (the
 *       scan), _end makes the context's stream wait for it; in between the caller may enqueue what needs no counters (the
 *       record loop's block cut, mg_cut_blocks_device), which then runs beside the collective.  main.cpp has neither: its
 *       loop B starts when loop A has ended (main.cpp:500-522).
 * mg_exchange_stats: duration of the context's most recent exchange (collective + pack / unpack, HIP events on its stream). */
int mg_counters_allreduce_begin(mg_ctx *ctx);
int mg_counters_allreduce_end(mg_ctx *ctx);
int mg_exchange_stats(mg_ctx *ctx, float *ms_out, int *packed_out);

/* ---- per-variant path ----------------------------------------------------- */

/* set_coverages (main.cpp:151-184) over flat signature descriptors of any
 * number of blocks: allele slot a owns signatures [allele_sig_off[a],
 * allele_sig_off[a+1]); signature s owns rows [sig_kmer_off[s], sig_kmer_off[s+1]);
 * is_ref[row] != 0 -> KMAP::get_count, else BF(bf)::get_count.
 * cov_out[a] = max over signatures of the truncating running mean. */
int mg_lookup_cover(mg_ctx *ctx, const char *rows, size_t stride, size_t n_rows, const uint8_t *is_ref,
                    const uint64_t *sig_kmer_off, size_t n_sigs, const uint64_t *allele_sig_off, size_t n_alleles,
                    uint32_t *cov_out);

/* Block cutting of the record loops (main.cpp:341, 547: `!vb.is_near_to_last(v) || last_seq_name != v.seq_name`, with
 * VB::is_near_to_last = are_near(last variant, v), var_block.hpp:77-80, 417-423 -- in the reference's float arithmetic) for
 * a batch of n_vars KEPT records in file order, on the device.  contig_id[i] identifies record i's sequence; give
 * contig_id[0] the id of `last_seq_name` as it stands when record 0 arrives (the name of the file's first record, kept or
 * not, for the first batch; the previous kept record's name afterwards) and put the previous batch's last record in front
 * as record 0 to continue a block across batches.  blk_var_off_out (n_vars + 1 entries) receives the first record of
 * every block and n_vars behind the last; these are the blk_var_off of mg_cover_blocks / mg_index_blocks. */
int mg_cut_blocks(mg_ctx *ctx, size_t n_vars, const int32_t *pos, const uint32_t *ref_size, const uint32_t *min_size,
                  const uint32_t *contig_id, uint32_t *blk_var_off_out, size_t *n_blocks_out);

/* VB::extract_kmers (var_block.hpp:95-219, chains :436-677, haplotype picks :709-786) fused with
 * set_coverages (main.cpp:151-184) for blocks of any shape, enumerated ON THE DEVICE.  Variants are flat
 * across blocks (blk_var_off); pos is the 0-based position in the block's contig, which starts at
 * blk_ref_base in the uploaded reference and is blk_ref_len long; canon[slot] = first allele index of
 * the variant with the same text (variant.hpp:228); gt[v * n_samples + s] = a1 | a2 << 7 | phased << 14
 * for the kept panel samples (variant.hpp:158-211); a1 and a2 are below the record's allele count (the reference
 * reads out of bounds otherwise; the library does not check: its caller's reader does).  cov_out as in mg_lookup_cover, one slot per
 * (variant, allele).  overflow_out[v] = 1 where a fixed device capacity (16 chains per side, 32 members per
 * chain side, 14 unphased members, 127 alleles, k <= 64) or a window clipped by a contig end was hit: redo
 * that variant's block through the host enumerator + mg_lookup_cover. */
int mg_cover_blocks(mg_ctx *ctx, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len,
                    const uint32_t *blk_var_off, size_t n_vars, const int32_t *pos, const uint32_t *ref_size,
                    const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                    const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon,
                    const uint16_t *gt, uint32_t n_samples, int haploid, uint32_t *cov_out, uint8_t *overflow_out);

/* The same enumeration at INDEX time: VB::extract_kmers + add_kmers_to_bf (main.cpp:349-350, 122-144) for blocks of any
 * shape -- every signature k-mer of allele 0 is added to the exact map (KMAP::add_key), every other one sets its bit of
 * `bf` (BF::add_key).  Arguments as mg_cover_blocks (the panel genotypes decide which alleles have signatures); the
 * blocks hold only the variants `index` keeps (has_alts and is_present, main.cpp:332).  overflow_out[v] = 1: nothing of
 * that variant was inserted (a device capacity, a window clipped by a contig end, or a REF k-mer the packed table cannot
 * hold) -- enumerate its block on the host and insert with mg_map_insert / mg_bf_insert.  Before mg_bf_finalize.
 * Counter ids of keys inserted here follow the order the device reached them in; ranks of a multi-GPU call agree on
 * the layout by loading the same index file. */
int mg_index_blocks(mg_ctx *ctx, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len,
                    const uint32_t *blk_var_off, size_t n_vars, const int32_t *pos, const uint32_t *ref_size,
                    const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                    const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon,
                    const uint16_t *gt, uint32_t n_samples, int haploid, uint8_t *overflow_out);

/* ---- the record loop on a RESIDENT panel (main.cpp:522-579 / 309-370 with every array already in HBM) --------------------
 * A panel is the kept records of a VCF (or of a batch of one) in file order, as flat arrays: every pointer below is a
 * DEVICE pointer on the context's GPU, nothing is copied or re-uploaded, and the calls are asynchronous on the
 * context's stream (mg_index_blocks_device excepted, see there).  Record v sits on sequence contig_id[v], which starts
 * at contig_base[] in the buffer of mg_reference_upload and is contig_len[] long; the other arrays are those of
 * mg_cover_blocks.  One step of `call` on a resident panel is
 *     mg_cut_blocks_device -> mg_cover_blocks_device -> mg_genotype_device
 * and of `index`:  mg_cut_blocks_device -> mg_index_blocks_device. */
typedef struct mg_panel_dev {
    uint64_t n_vars;
    uint32_t n_contigs;
    uint32_t n_samples;
    const uint64_t *contig_base;     /* [n_contigs] */
    const uint32_t *contig_len;      /* [n_contigs] */
    const uint32_t *contig_id;       /* [n_vars]; contig_id[0] as for mg_cut_blocks */
    const int32_t *pos;              /* [n_vars] Variant::ref_pos */
    const uint32_t *ref_size;        /* [n_vars] */
    const uint32_t *min_size;        /* [n_vars] */
    const uint8_t *present;          /* [n_vars] Variant::is_present */
    const uint32_t *var_allele_off;  /* [n_vars + 1] */
    const uint32_t *allele_off;      /* [slots + 1] */
    const char *pool;
    const uint8_t *canon;            /* [slots] */
    const uint16_t *gt;              /* [n_vars][n_samples] -- or NULL with the sparse form below */
    /* sparse genotypes (a panel of tens of thousands of samples is nearly all 0|0): record v's entries are
     * [sp_off[v], sp_off[v + 1]) of sp_sample (ascending sample numbers) / sp_gt (their genotype words); every sample
     * without an entry carries the word sp_default (1 << 14 = 0|0 phased for a phased panel, 0 = 0/0 for an unphased one:
     * the phase bit of a homozygous genotype still decides how its sample's OTHER genotypes along a chain combine,
     * var_block.hpp:758-782, so it is part of the word).  Given (non-NULL sp_off), `gt` is ignored. */
    const uint32_t *sp_off;          /* [n_vars + 1] or NULL */
    const uint32_t *sp_sample;
    const uint16_t *sp_gt;
    uint32_t sp_default;
    /* bytes of `pool` (= allele_off[last allele slot]).  Given (non-zero, and `pool` 4-byte aligned) the record loop packs the
     * alleles to 2 bits per base at the start of every call and assembles signature k-mers from the packed form; 0: from the bytes. */
    uint64_t pool_bytes;
} mg_panel_dev;
/* The cut (main.cpp:341, 547) of all n_vars records: d_blk_var_off_out ([n_vars + 1] u32), d_n_blocks_out (one u64), and
 * -- optional, NULL to skip -- d_var_block_out ([n_vars] u32: the block of every record, which the two calls below
 * otherwise derive again).  Needs contig_id, pos, ref_size, min_size only. */
int mg_cut_blocks_device(mg_ctx *ctx, const mg_panel_dev *panel, void *d_blk_var_off_out, void *d_var_block_out, void *d_n_blocks_out);
/* extract_kmers + set_coverages (main.cpp:556-557) for every block: d_cov_out ([slots] u32) and d_overflow_out ([n_vars]
 * u8) as mg_cover_blocks returns them.  A block is evaluated against the sequence of its first record (`last_seq_name` at
 * the flush, main.cpp:556).  Three tiers (csrc/block_pipeline.h): blocks of one variant whose alleles are all shorter than k
 * take the fused lone-variant lookup of mg_call_isolated; the other records a pipeline of flat kernels (one thread per record
 * for the chain walks, one wave per chain for the distinct haplotype picks, one thread per signature k-mer); what exceeds
 * that pipeline's capacities the workgroup-per-record kernel; what exceeds ITS capacities is flagged in d_overflow_out. */
int mg_cover_blocks_device(mg_ctx *ctx, const mg_panel_dev *panel, const void *d_blk_var_off, const void *d_var_block /* or NULL */,
                           const void *d_n_blocks, int haploid, void *d_cov_out, void *d_overflow_out);
/* extract_kmers + add_kmers_to_bf (main.cpp:349-350) for every block (the panel holds only what `index` keeps,
 * main.cpp:332); d_overflow_out as mg_index_blocks.  The arrays stay where they are, but the call waits for the device
 * twice on eight bytes: the exact map is sized from a counting pass before the insert pass runs. */
int mg_index_blocks_device(mg_ctx *ctx, const mg_panel_dev *panel, const void *d_blk_var_off, const void *d_var_block /* or NULL */,
                           const void *d_n_blocks, int haploid, void *d_overflow_out);

/* Panel genotypes straight from VCF text: Variant::extract_genotypes (variant.hpp:158-211) over the sample columns of a batch
 * of records, decoded on the device into the sparse form of mg_panel_dev.  The host finds each record's ninth tab and the
 * position of GT in its FORMAT column and parses nothing else of the sample columns.
 *   text, text_bytes      host buffer holding the records' lines (only the range the spans cover is uploaded)
 *   span_off/span_len[r]  record r's sample columns: from the byte after the FORMAT column's tab to the end of the line
 *                         (terminator excluded); length 0 = the record has no sample columns (every kept sample reads ".")
 *   gt_index[r]           position of GT among the ':'-separated FORMAT keys (the caller has checked that it is there)
 *   n_columns, keep       sample columns of the header; keep[i] != 0: column i is one of the kept samples (-s), NULL = all
 * Out, per record: sp_off[n_records + 1] (entries of record r at [sp_off[r], sp_off[r + 1])), raw_mask (bit a: raw allele
 * number a, mod 64, occurs in a kept sample -- both alleles unless haploid), max_allele (the largest allele number: the
 * caller checks it against the record's allele count as variant.hpp would crash on it, and decodes a record with a number
 * above 127 itself, whose words here are truncated).  *sp_default = the word of the samples without an entry (0|0 phased or
 * 0/0, whichever the batch holds more of), *n_entries = sp_off[n_records].  The entries themselves stay on the device until
 * mg_decode_gt_entries copies them out (sample numbers count KEPT samples, ascending inside a record).  Synchronous. */
int mg_decode_gt_text(mg_ctx *ctx, const char *text, size_t text_bytes, size_t n_records, const uint64_t *span_off, const uint32_t *span_len,
                      const int32_t *gt_index, uint32_t n_columns, const uint8_t *keep, int haploid, uint16_t *sp_default, uint32_t *sp_off,
                      uint64_t *raw_mask, uint32_t *max_allele, uint64_t *n_entries);
int mg_decode_gt_entries(mg_ctx *ctx, uint32_t *sp_sample, uint16_t *sp_gt);

/* mg_cover_blocks / mg_index_blocks with the panel's genotypes in the sparse form of mg_panel_dev (sp_off, sp_sample, sp_gt:
 * only the samples whose genotype word is not sp_default).  What crosses PCIe for a 27,934-sample panel drops from 56 KB per
 * record to a few bytes per non-reference genotype. */
int mg_cover_blocks_sparse(mg_ctx *ctx, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len,
                           const uint32_t *blk_var_off, size_t n_vars, const int32_t *pos, const uint32_t *ref_size,
                           const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                           const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon,
                           const uint32_t *sp_off, const uint32_t *sp_sample, const uint16_t *sp_gt, uint16_t sp_default,
                           uint32_t n_samples, int haploid, uint32_t *cov_out, uint8_t *overflow_out);
int mg_index_blocks_sparse(mg_ctx *ctx, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len,
                           const uint32_t *blk_var_off, size_t n_vars, const int32_t *pos, const uint32_t *ref_size,
                           const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                           const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon,
                           const uint32_t *sp_off, const uint32_t *sp_sample, const uint16_t *sp_gt, uint16_t sp_default,
                           uint32_t n_samples, int haploid, uint8_t *overflow_out);

/* Result codes of mg_genotype / mg_call_isolated per variant */
#define MG_GT_NORMAL 0   /* likelihood list computed                              */
#define MG_GT_OVERCOV 1  /* some allele > max_cov: (best_geno,0) per such allele  */
#define MG_GT_SINGLE 2   /* one allele only: (best_geno,1)                        */
#define MG_GT_NOCOV 3    /* all coverages 0: (best_geno,0)                        */

/* Index-time counterpart of mg_call_isolated for blocks of ONE variant whose alleles are all shorter than k: VB::extract_kmers
 * with comb = {v} (var_block.hpp:95-219) + add_kmers_to_bf (main.cpp:122-144) on the device -- allele 0's signature k-mer goes
 * into the exact map (KMAP::add_key), the signature of every other allele some panel haplotype carries (present_mask) sets its
 * bit of `bf` (BF::add_key).  pos, offsets, pool, present_mask and flags as for mg_call_isolated (flags bit0: is_present and
 * not within k of a contig end, var_block.hpp:104; flanks inside the contig).  overflow_out[v] = 1: nothing of that variant
 * was inserted (a base outside ACGT in its window or alleles, more than 64 alleles, k outside 17..64) -- enumerate it on the
 * host and insert with mg_map_insert / mg_bf_insert.  Before mg_bf_finalize; after mg_reference_upload.  Variant v's REF key
 * takes insertion row (rows so far) + v; the index FILE fixes the counter layout for every GPU of a call. */
int mg_index_isolated(mg_ctx *ctx, size_t n_vars, const uint64_t *pos, const uint32_t *var_allele_off, const uint32_t *allele_off,
                      const char *allele_pool, size_t pool_len, const uint64_t *present_mask, const uint8_t *flags,
                      uint8_t *overflow_out);

/* VB::genotype (var_block.hpp:224-330) + the normalise / first-strict-max / GQ
 * part of VB::output_variants (:366-394).  Variant v owns allele slots
 * [var_allele_off[v], var_allele_off[v+1]).  gt2 = -1 in haploid mode.
 * probs (optional, may be NULL): normalised list in the reference's order,
 * variant v at var_gt_off[v] (caller-provided offsets, A or A(A+1)/2 each). */
int mg_genotype(mg_ctx *ctx, const uint32_t *cov, const float *freq, const uint32_t *var_allele_off, size_t n_vars,
                float error_rate, int max_cov, int haploid, int32_t *gt1, int32_t *gt2, int32_t *gq,
                uint8_t *status, double *probs, const uint64_t *var_gt_off);
/* same, every array already resident on the device (asynchronous) */
int mg_genotype_device(mg_ctx *ctx, const void *d_cov, const void *d_freq, const void *d_var_allele_off, size_t n_vars,
                       float error_rate, int max_cov, int haploid, void *d_gt1, void *d_gt2, void *d_gq, void *d_status,
                       void *d_probs, const void *d_var_gt_off);

/* Fused device path for blocks that hold ONE variant whose alleles are all
 * shorter than k (the isolated-SNP/indel case): signature enumeration
 * (var_block.hpp:95-219 with comb = {v}), lookup + coverage, likelihoods and
 * GT/GQ in one launch.  `reference` is the concatenation of the upper-cased
 * contigs already uploaded with mg_reference_upload; pos[v] is the variant's
 * offset in that buffer.  flags bit0: eligible (is_present and not within k of
 * a contig end, var_block.hpp:104).  An eligible variant's flanks -- k/2 bases
 * before pos, ceil(k/2) after the REF allele -- must lie inside its contig: a
 * right flank clipped by the contig end makes a shorter k-mer in the reference
 * (var_block.hpp:187), which is mg_cover_blocks' / the host enumerator's case.  present_mask bit a: some panel haplotype
 * carries allele a (build_alleles_combs, var_block.hpp:734-786).  probs (optional):
 * normalised likelihood lists at caller-provided var_gt_off, as in mg_genotype. */
int mg_reference_upload(mg_ctx *ctx, const char *ascii, size_t len);
int mg_reference_upload_device(mg_ctx *ctx, const void *d_ascii, size_t len); /* from a device buffer (copied) */
int mg_call_isolated(mg_ctx *ctx, size_t n_vars, const uint64_t *pos, const uint32_t *var_allele_off,
                     const uint32_t *allele_off, const char *allele_pool, size_t pool_len, const float *freq,
                     const uint64_t *present_mask, const uint8_t *flags, float error_rate, int max_cov, int haploid,
                     uint32_t *cov_out, int32_t *gt1, int32_t *gt2, int32_t *gq, uint8_t *status, double *probs,
                     const uint64_t *var_gt_off);
/* same, every array already resident on the device (asynchronous).  d_probs / d_var_gt_off
 * (both or neither; as in mg_genotype) double as the workspace that saves recomputing
 * each likelihood for the normalisation pass. */
int mg_call_isolated_device(mg_ctx *ctx, size_t n_vars, const void *d_pos, const void *d_var_allele_off,
                            const void *d_allele_off, const void *d_allele_pool, const void *d_freq,
                            const void *d_present_mask, const void *d_flags, float error_rate, int max_cov,
                            int haploid, void *d_cov_out, void *d_gt1, void *d_gt2, void *d_gq, void *d_status,
                            void *d_probs, const void *d_var_gt_off);

/* ---- index payloads  (bloom_filter.hpp:127-146, kmap.hpp:52-82) ----------- */

/* BF: _mode, _size, bit words (ceil(size/64) u64), counters (n_set u16).
 * Call with NULL buffers to query sizes first. */
int mg_bf_export(mg_ctx *ctx, int which, uint64_t *words_out, uint16_t *counts_out);
int mg_bf_import(mg_ctx *ctx, int which, int mode, uint64_t size_bits, const uint64_t *words,
                 const uint16_t *counts, uint64_t n_counts);
/* The same payload in sparse form (what the index file holds): the n_set ascending
 * bit positions (= counter order) and the counters.  Export needs a finalised filter. */
int mg_bf_export_sparse(mg_ctx *ctx, int which, uint64_t *positions_out, uint16_t *counts_out);
int mg_bf_import_sparse(mg_ctx *ctx, int which, int mode, uint64_t size_bits, const uint64_t *positions,
                        const uint16_t *counts, uint64_t n);
/* KMAP: n keys as NUL-terminated rows of `stride` bytes + values */
int mg_map_export(mg_ctx *ctx, char *rows_out, size_t stride, int32_t *vals_out);
int mg_map_import(mg_ctx *ctx, const char *rows, size_t stride, size_t n, const int32_t *vals);

/* ---- introspection for tests / profiling ---------------------------------- */

/* hash % size of BF::_get_hash for each row (bloom_filter.hpp:67-74,84) */
int mg_debug_bf_index(mg_ctx *ctx, int which, const char *rows, size_t stride, size_t n, uint64_t *idx_out);
/* same from packed k-mers of length klen (1..64), MSB-first right-aligned */
int mg_debug_packed_index(mg_ctx *ctx, int which, const uint64_t *hi, const uint64_t *lo, size_t n, uint32_t klen,
                          uint64_t *idx_out);
/* timing of the first chunk of the most recent mg_kmc_scan* in milliseconds (HIP
 * events on the context's stream): ms_out[0] filter kernel, [1] probe kernel,
 * [2] hit kernel (with the ticket form [0] is its two passes together); rows_out[0] = rows that passed the gate (last chunk),
 * rows_out[1] = rows whose bf bit was set (whole call) */
int mg_scan_stats(mg_ctx *ctx, float *ms_out, uint64_t *rows_out);
/* timing and counts of the most recent mg_cover_blocks_device (waits for it): ms_out[0] tier 1 (classification + the
 * lone-variant lookups), [1] tier 2 (the flat pipeline: chain walks, distinct picks, one thread per signature k-mer), [2] tier 3
 * (the workgroup-per-record kernel on what tier 2 handed on) + the final pass; counts_out[0] records beyond tier 1, [1] signature
 * k-mers of the lone records, [2] signature k-mers tiers 2 and 3 assembled, [3] records tier 3 took */
int mg_blocks_stats(mg_ctx *ctx, float *ms_out, uint64_t *counts_out);
/* 0 disables the cache-resident summary bitmaps (A/B switch; results identical) */
int mg_set_option(mg_ctx *ctx, const char *name, int64_t value);
/* reads an option back, plus "pregate_k" (0: single-level gate), "scan_bins" (slices the
 * most recent scan partitioned its second level into; 0: direct form), "scan_tickets" (gate slices the most
 * recent scan filed tickets under; 0: it did not) and "scan_spilled" */
int mg_get_option(mg_ctx *ctx, const char *name, int64_t *value);

#ifdef __cplusplus
}
#endif
#endif /* MALVA_HIP_H */
