// variant_kernels.h -- coverage reduction, likelihoods, the fused lone-variant kernel and the device-side block enumerator
// Part of the malva_hip translation unit: included by malva_hip.hip inside its anonymous namespace, after
// geno_dev.h (which brings xxh3_dev.h and kmer_dev.h).  See DESIGN.md section 4 for the kernels' rooflines.
#pragma once

// ---- V1: coverage reduction (main.cpp:159-181) -----------------------------------
__global__ void __launch_bounds__(TPB) cover_kernel(const i32 *w, const u64 *sig_kmer_off, const u64 *allele_sig_off,
                                                    u64 n_alleles, u32 *cov)
{
    const u64 a = (u64)blockIdx.x * TPB + threadIdx.x;
    if (a >= n_alleles) return;
    u32 allele_cov = 0;
    for (u64 s = allele_sig_off[a]; s < allele_sig_off[a + 1]; ++s) {
        u32 curr = 0;
        i32 n = 0;
        for (u64 j = sig_kmer_off[s]; j < sig_kmer_off[s + 1]; ++j) {
            const i32 wt = w[j];
            if (wt > 0) {
                curr = (curr * (u32)n + (u32)wt) / (u32)(n + 1);
                ++n;
            }
        }
        if (curr > allele_cov) allele_cov = curr;
    }
    cov[a] = (u32)(float)allele_cov; // through the float parameter of set_variant_coverage (var_block.hpp:84)
}

// ---- G1-G3 ---------------------------------------------------------------------------
__global__ void __launch_bounds__(TPB) genotype_kernel(const u32 *cov, const float *freq, const u32 *var_allele_off,
                                                       u64 n_vars, GenoParams p, i32 *gt1, i32 *gt2, i32 *gq, u8 *status,
                                                       double *probs, const u64 *var_gt_off)
{
    const u64 v = (u64)blockIdx.x * TPB + threadIdx.x;
    if (v >= n_vars) return;
    const u32 a0 = var_allele_off[v], A = var_allele_off[v + 1] - a0;
    genotype_one(cov + a0, freq + a0, (int)A, p, gt1 + v, gt2 + v, gq + v, status + v,
                 probs ? probs + var_gt_off[v] : nullptr);
}

// ---- fused isolated-variant path ---------------------------------------------------
// The signature k-mer of allele a of a lone variant (var_block.hpp:145-200 with
// comb = {v}):  ref[pos-mp, pos) + allele + ref[pos+ref_size, +ms),
// mp = k/2 - len/2,  ms = ceil(k/2) - (len - len/2).
struct SigIn {
    const u8 *ref_left;  // reference + pos - mp
    const u8 *allele;
    const u8 *ref_right; // reference + pos + ref_size
    int mp, alen;
    __device__ __forceinline__ u32 operator()(int i) const
    {
        return i < mp ? ref_left[i] : (i < mp + alen ? allele[i - mp] : ref_right[i - mp - alen]);
    }
};
// weight of one signature k-mer given as bytes: KMAP::get_count (allele 0) or BF::get_count
template <class IN> __device__ __forceinline__ i32 weight_bytes(const IN &in, int k, bool is_ref, const BFView &bf, const MapView &map)
{
    CanonBytes<IN> can(in, k);
    if (is_ref) {
        U128 key;
        if (pack_regular(can, k, (int)map.klen, &key)) {
            const u64 h = xxh3_bytes(can, k);
            return map_value(map, key, h, mod_size(h, bf.mod));
        }
        return 0;
    }
    const u64 idx = mod_size(xxh3_bytes(can, k), bf.mod);
    return (i32)(uint16_t)bf_count_at(bf, idx);
}
// 2-bit code of an upper-case ACGT byte without a table: (b >> 1) & 3 gives A0 C1 G3 T2
__device__ __forceinline__ u32 acgt_code(u32 b, bool *ok)
{
    *ok = b == 'A' || b == 'C' || b == 'G' || b == 'T';
    const u32 c = (b >> 1) & 3;
    return c ^ (c >> 1);
}
// n <= 32 bases starting at an arbitrary byte address -> 2-bit L-form, four bases per aligned
// dword load.  *bad gets bit 4j set when dword j holds a byte outside ACGT (coarse on purpose:
// a flagged span sends the allele down the exact byte-wise path).  Reads whole aligned dwords,
// i.e. up to 3 bytes either side of the span: the reference buffer is padded for that.
// The aligned dwords under [p, p + n), n <= 32: all requested at once (at most nine, predicated), then consumed --
// a loop that loads the next dword as it goes waits for memory once per dword (measured on the isolated-variant
// kernel: 0.228 -> 0.197 ms per 1e6 SNPs).
struct SpanWords {
    u32 w[10];
    u32 sh;
};
__device__ __forceinline__ SpanWords span_load(const u8 *p, int n)
{
    const u64 addr = (u64)p;
    const u32 *q = (const u32 *)(addr & ~3ULL);
    SpanWords s;
    s.sh = (u32)(addr & 3);
    const int nd = (n + 3) / 4 + 1; // dwords j and j + 1 for every 4 * j < n
#pragma unroll
    for (int j = 0; j < 10; ++j) s.w[j] = j < nd ? q[j] : 0u;
    return s;
}
__device__ __forceinline__ void span_pack(const SpanWords &s, int n, u64 *codes, u64 *bad)
{
    u64 c = 0, b = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (4 * j >= n) break;
        const u32 prev = s.w[j], next = s.w[j + 1];
        const u32 d = s.sh ? __builtin_amdgcn_alignbyte(next, prev, s.sh) : prev;
        u32 t = (d >> 1) & 0x03030303u; // per byte: A0 C1 G3 T2
        t ^= (t >> 1) & 0x01010101u;    //           A0 C1 G2 T3
        const u32 c8 = (t * 0x01041040u) >> 24;
        const int left = n - 4 * j;
        const u32 m = left >= 4 ? 0xFFFFFFFFu : ((1u << (8 * left)) - 1);
        if ((expand4(c8) ^ d) & m) b |= 0xFULL << (4 * j);
        c |= (u64)(left >= 4 ? c8 : (c8 & ((1u << (2 * left)) - 1))) << (8 * j);
    }
    *codes = c;
    *bad = b;
}
// Fast path: both flanks and the allele are pure ACGT, so the signature is assembled in
// 2-bit form from flanks packed once per variant (shared by its alleles), canonicalised
// with integer compares and hashed with the register-resident XXH3 -- the same code the
// scan uses.  Anything else (N / IUPAC in the window, k outside 17..64) takes weight_bytes.
//
// Two kernels.  The lookups are chains of dependent random reads (map slot -> value; filter word -> rank block ->
// counter) and want many waves in flight; the likelihoods are f64 arithmetic over a dozen live doubles and want
// registers.  Fused in one kernel (the first form) the likelihoods' 193 VGPRs left 2 waves per SIMD to hide the
// lookups' latency.  The coverage words written by the first kernel and read by the second are 8 B per SNP.
//   iso_cover_kernel<false>: TWO threads per variant, alleles split by parity (a biallelic SNP: REF on one, ALT on
//       the other); a signature with a non-ACGT base is only marked (ISO_SLOW) and `*need_slow` set to this call's number
//   iso_cover_kernel<true>: the same grid, returns at once unless `*need_slow` holds this call's number; redoes the marked alleles byte-wise
//       (kept out of the first kernel because its generic XXH3 alone needs 190 VGPRs)
//   iso_genotype_kernel: one thread per variant
constexpr u32 ISO_SLOW = 0xFFFFFFFFu; // never a coverage: those are float-rounded counts below 2^31
// one thread's share (alleles of parity `par`) of a lone variant: `site` = its offset in the uploaded reference,
// pm = mask of the alleles some panel haplotype carries, live = eligible (var_block.hpp:104)
// ref2 / refbad: the packed reference (scan_kernels.h); the flanks come out of it in two loads and a shift each, and a
// 128-byte line of it holds 512 bases (five SNPs of a whole-genome panel share one, against one line of text each)
template <bool SLOW>
__device__ __forceinline__ void iso_cover_body(const u8 *reference, const u64 *__restrict__ ref2, const u32 *__restrict__ refbad, u64 site_off, u32 a0, u32 A, bool live,
                                               u64 pm_in, u32 par, const u32 *allele_off, const u8 *pool, int k, const BFView &bf, const MapView &map, u32 *cov_out,
                                               u32 *need_slow, u32 call_no)
{
    const u32 ref_size = allele_off[a0 + 1] - allele_off[a0];
    u32 *cov = cov_out + a0;
    const u64 pm = live ? pm_in : 0;
    const u8 *site = reference + site_off;
    const int lmax = k / 2, rmax = (k + 1) / 2;
    const bool packed_ok = k >= 17 && k <= MG_MAX_PACKED_K;
    // flanks as L-forms: left = ref[pos-lmax, pos), right = ref[pos+ref_size, +rmax)  (<= 32 bases each)
    u64 lf = 0, rf = 0, lbad = 0, rbad = 0;
    if (!SLOW && packed_ok && live) {
        if (ref2) { // (bad masks: one bit per base here, four per dword below; both are used as "any bad base in the part taken")
            lf = ref_codes(ref2, site_off - lmax, lmax);
            rf = ref_codes(ref2, site_off + ref_size, rmax);
            lbad = ref_badbits(refbad, site_off - lmax, lmax);
            rbad = ref_badbits(refbad, site_off + ref_size, rmax);
        } else { // both flanks' dwords requested before either is consumed
            const SpanWords ls = span_load(site - lmax, lmax), rs = span_load(site + ref_size, rmax);
            span_pack(ls, lmax, &lf, &lbad);
            span_pack(rs, rmax, &rf, &rbad);
        }
    }
    for (u32 a = par; a < A; a += 2) {
        if (SLOW) {
            if (cov[a] != ISO_SLOW) continue;
            const int alen = (int)(allele_off[a0 + a + 1] - allele_off[a0 + a]);
            const int mp = k / 2 - alen / 2;
            const i32 w = weight_bytes(SigIn{site - mp, pool + allele_off[a0 + a], site + ref_size, mp, alen}, k, a == 0, bf, map);
            cov[a] = w > 0 ? (u32)(float)(u32)w : 0;
            continue;
        }
        u32 out = 0;
        const int alen = (int)(allele_off[a0 + a + 1] - allele_off[a0 + a]);
        const int mp = k / 2 - alen / 2, ms = (k + 1) / 2 - (alen - alen / 2);
        // alleles >= k take the general path (host contract); alleles past the 64-bit presence mask are not looked up
        if (a < 64 && ((pm >> a) & 1) && mp >= 0 && ms >= 0) {
            const u8 *al = pool + allele_off[a0 + a];
            bool fast = packed_ok && (lbad >> (lmax - mp)) == 0 && (ms == 0 || (rbad & ((1ULL << ms) - 1)) == 0);
            U128 L{0, 0};
            if (fast) {
                for (int i = 0; i < alen; ++i) {
                    bool ok;
                    const u64 code = acgt_code(al[i], &ok);
                    fast &= ok;
                    if (i < 32) L.lo |= code << (2 * i);
                    else L.hi |= code << (2 * (i - 32));
                }
            }
            i32 w = 0;
            if (!fast) {
                out = ISO_SLOW;
                *need_slow = call_no;
            } else {
                L = shl128(L, 2 * mp);
                if (mp) L.lo |= lf >> (2 * (lmax - mp));                         // last mp bases of the left flank
                if (ms) {
                    const U128 r = shl128(U128{ms >= 32 ? rf : rf & ((1ULL << (2 * ms)) - 1), 0}, 2 * (mp + alen));
                    L.lo |= r.lo;
                    L.hi |= r.hi;
                }
                const U128 mk = mask128(2 * k);
                const U128 mform = shr128(U128{pairrev64(L.hi), pairrev64(L.lo)}, 2 * (64 - k)); // M-form of the k-mer
                const U128 rc{~mform.lo & mk.lo, ~mform.hi & mk.hi};                               // L-form of its reverse complement
                const U128 key = lt128(L, rc) ? L : rc;
                const u64 h = k == 35 ? xxh3_packed_fixed<35>(key.lo, key.hi) : xxh3_packed(key, k);
                const u64 idx = mod_size(h, bf.mod);
                if (a == 0) w = map_value(map, key, h, idx);
                else w = (i32)bucket_count(map, bf.counts, idx); // the filter's directory entry sits in the exact map's record of the same slot
                if (w > 0) out = (u32)(float)(u32)w;
            }
        }
        cov[a] = out;
    }
}
template <bool SLOW>
__global__ void __launch_bounds__(TPB, SLOW ? 1 : 8) iso_cover_kernel(const u8 *reference, const u64 *__restrict__ ref2, const u32 *__restrict__ refbad, u64 n_vars, const u64 *pos,
                                                        const u32 *var_allele_off, const u32 *allele_off, const u8 *pool, const u64 *present_mask,
                                                        const u8 *flags, int k, BFView bf, MapView map, u32 *cov_out, u32 *need_slow,
                                                        u32 call_no)
{
    if (SLOW && *need_slow != call_no) return;
    const u64 t = (u64)blockIdx.x * TPB + threadIdx.x;
    const u64 v = t >> 1;
    if (v >= n_vars) return;
    const u32 a0 = var_allele_off[v], A = var_allele_off[v + 1] - a0;
    const bool live = flags[v] & 1;
    iso_cover_body<SLOW>(reference, ref2, refbad, pos[v], a0, A, live, live ? present_mask[v] : 0, (u32)(t & 1), allele_off, pool, k, bf, map, cov_out, need_slow, call_no);
}

__global__ void __launch_bounds__(TPB) iso_genotype_kernel(u64 n_vars, const u32 *var_allele_off, const float *freq, GenoParams p,
                                                           const u32 *cov, i32 *gt1, i32 *gt2, i32 *gq, u8 *status, double *probs,
                                                           const u64 *var_gt_off)
{
    const u64 v = (u64)blockIdx.x * TPB + threadIdx.x;
    if (v >= n_vars) return;
    const u32 a0 = var_allele_off[v], A = var_allele_off[v + 1] - a0;
    genotype_one(cov + a0, freq + a0, (int)A, p, gt1 + v, gt2 + v, gq + v, status + v, probs ? probs + var_gt_off[v] : nullptr);
}

// ---- isolated variants at INDEX time ------------------------------------------------------------------------------------
// VB::extract_kmers for a block of one variant whose alleles are all shorter than k (var_block.hpp:95-219 with comb = {v})
// + add_kmers_to_bf (main.cpp:122-144): the signature k-mer of allele 0 goes into the exact map (KMAP::add_key), that of
// every other allele some panel haplotype carries sets its bit of `bf` (BF::add_key).  One thread per variant; the
// signatures are assembled exactly as iso_cover_kernel's fast path assembles them.  A variant with a base outside ACGT
// in its window or alleles, more than 64 alleles, or k outside 17..64 is only flagged (nothing of it is inserted): the
// host enumerates it (extract_lone) and inserts through the batch calls.  Variant v's REF key takes insertion row row0 + v.
// returns false when nothing of the variant was inserted (the host enumerates it)
__device__ __forceinline__ bool iso_index_body(const u8 *reference, u64 site_off, u32 a0, u32 A, u64 pm, const u32 *allele_off, const u8 *pool, int k,
                                               const BFView &bf, const MapView &map, u32 my_row, u32 row0)
{
    if (pm == 0) return true;
    const u32 ref_size = allele_off[a0 + 1] - allele_off[a0];
    const u8 *site = reference + site_off;
    const int lmax = k / 2, rmax = (k + 1) / 2;
    if (k < 17 || k > MG_MAX_PACKED_K || A > 64 || k != (int)map.klen) return false;
    u64 lf = 0, rf = 0, lbad = 0, rbad = 0;
    {
        const SpanWords ls = span_load(site - lmax, lmax), rs = span_load(site + ref_size, rmax);
        span_pack(ls, lmax, &lf, &lbad);
        span_pack(rs, rmax, &rf, &rbad);
    }
    // the canonical signature of allele a (L-form), or false when something in it is not ACGT / it does not fit the lone-variant form
    auto signature = [&](u32 a, U128 *key) -> bool {
        const int alen = (int)(allele_off[a0 + a + 1] - allele_off[a0 + a]);
        const int mp = k / 2 - alen / 2, ms = (k + 1) / 2 - (alen - alen / 2);
        if (mp < 0 || ms < 0) return false; // an allele of k bases or more: not this kernel's case
        if ((lbad >> (lmax - mp)) != 0 || (ms != 0 && (rbad & ((1ULL << ms) - 1)) != 0)) return false;
        const u8 *al = pool + allele_off[a0 + a];
        U128 L{0, 0};
        bool fast = true;
        for (int i = 0; i < alen; ++i) {
            bool ok;
            const u64 code = acgt_code(al[i], &ok);
            fast &= ok;
            if (i < 32) L.lo |= code << (2 * i);
            else L.hi |= code << (2 * (i - 32));
        }
        if (!fast) return false;
        L = shl128(L, 2 * mp);
        if (mp) L.lo |= lf >> (2 * (lmax - mp)); // last mp bases of the left flank
        if (ms) {
            const U128 r = shl128(U128{ms >= 32 ? rf : rf & ((1ULL << (2 * ms)) - 1), 0}, 2 * (mp + alen));
            L.lo |= r.lo;
            L.hi |= r.hi;
        }
        const U128 mk = mask128(2 * k);
        const U128 mform = shr128(U128{pairrev64(L.hi), pairrev64(L.lo)}, 2 * (64 - k)); // M-form of the k-mer
        const U128 rc{~mform.lo & mk.lo, ~mform.hi & mk.hi};                               // L-form of its reverse complement
        *key = lt128(L, rc) ? L : rc;
        return true;
    };
    U128 key;
    for (u32 a = 0; a < A; ++a) // first: can every carried allele be done here?
        if (((pm >> a) & 1) && !signature(a, &key)) return false;
    for (u32 a = 0; a < A; ++a) {
        if (!((pm >> a) & 1)) continue;
        signature(a, &key);
        const u64 h = xxh3_packed(key, k);
        if (a == 0)
            map_insert_key(map, bf, key, h, my_row, row0);
        else {
            const u64 idx = mod_size(h, bf.mod);
            atomicOr((unsigned long long *)&bf.words[idx >> 6], 1ULL << (idx & 63)); // BF::add_key, bloom_filter.hpp:81-85
            gate_set(bf, idx);
        }
    }
    return true;
}
__global__ void __launch_bounds__(TPB) iso_index_kernel(const u8 *reference, u64 n_vars, const u64 *pos, const u32 *var_allele_off, const u32 *allele_off,
                                                        const u8 *pool, const u64 *present_mask, const u8 *flags, int k, BFView bf, MapView map, u32 row0,
                                                        u8 *overflow)
{
    const u64 v = (u64)blockIdx.x * TPB + threadIdx.x;
    if (v >= n_vars) return;
    overflow[v] = 0;
    if (!(flags[v] & 1)) return; // not present, or within k of a contig end: no k-mers (var_block.hpp:104)
    const u32 a0 = var_allele_off[v], A = var_allele_off[v + 1] - a0;
    if (!iso_index_body(reference, pos[v], a0, A, present_mask[v], allele_off, pool, k, bf, map, row0 + (u32)v, row0)) overflow[v] = 1;
}

// ---- general blocks on the device: chains, haplotype picks, signature assembly, lookup, coverage -----------
// VB::extract_kmers (var_block.hpp:95-219) with get_combs_on_the_right/left (:436-624), combine_combs (:630-677),
// get_ref_subs (:682-702) and build_alleles_combs / combine_haplotypes (:709-786), fused with set_coverages
// (main.cpp:151-184), or at index time with add_kmers_to_bf (main.cpp:122-144).  One workgroup per variant.  Like the
// reference (build_alleles_combs' unordered_set), the kernel reduces the picks of all panel samples along a chain to the
// DISTINCT ones first -- a code of a few bits per member in an LDS hash set -- and evaluates each once; only a chain
// whose code exceeds 63 bits or whose set exceeds its LDS capacity has every sample's pick evaluated directly (a max,
// and a set insert, do not care about duplicates).
// Fixed capacities (chains per side, chain length, unphased fan-out): a variant that exceeds one is flagged in
// `overflow` and its block is redone by the host enumerator + mg_lookup_cover / mg_*_insert, so results never depend on them.
struct BlockBatch {
    const u8 *reference;      // concatenated contigs (mg_reference_upload)
    const u64 *ref2;          // the same as 2-bit codes, and one bit per base: not ACGT (ref_pack_kernel)
    const u32 *refbad;
    const u64 *contig_base;   // per sequence: its offset in `reference`
    const u32 *contig_len;    //               and its length
    const u32 *contig_id;     // [n_vars] sequence of each record; a block is evaluated against the sequence of its FIRST record
                              // (`last_seq_name` at the flush, main.cpp:556)
    const u32 *blk_var_off;   // [n_blocks + 1]
    const u32 *var_block;     // [n_vars] block of each variant
    const i32 *pos;           // 0-based position in the contig
    const u32 *ref_size, *min_size;
    const u8 *present;
    const u32 *var_allele_off; // [n_vars + 1] allele slots
    const u32 *allele_off;     // [n_slots + 1] into pool
    const u8 *pool;
    const u8 *canon;           // [n_slots] first allele index of the variant with the same text
    const uint16_t *gt;        // [n_vars][n_samples]: a1 | a2 << 7 | phased << 14 -- or, when sp_off is given, the SPARSE form:
    const u32 *sp_off;         // [n_vars + 1] record v's entries are [sp_off[v], sp_off[v + 1]) of ...
    const u32 *sp_sample;      // ... the samples (ascending) whose genotype is anything but 0|0 phased ...
    const uint16_t *sp_gt;     // ... and that genotype.  A panel of tens of thousands of samples is nearly all 0|0.
    u32 sp_default;            // the genotype word of every sample WITHOUT an entry (0|0 phased, or 0/0 for an unphased panel)
    int snp_chains;            // fw_eval_kernel's fixed-geometry assembly for chains of SNPs (option use_snp_chains)
    const unsigned short *rec_class; // [n_vars] per record outside tier 1: REC_* flags and the alleles' codes (written by the tier-1 kernels of the same call)
    const u64 *pool2;          // the allele pool packed like the reference (2 bits per base at the pool's own offsets) + its not-ACGT bits,
    const u32 *poolbad;        // built at the start of the call when the panel states pool_bytes; else NULL: alleles are read byte by byte
    u32 n_samples;
    int haploid, k;
    u32 set_limit; // distinct picks per chain held in the LDS set (<= BK_SET_CAP / 2; tests shrink it to reach the direct form)
};
// entry of sample s in record v's sparse genotypes, or -1
__device__ __forceinline__ long long sp_find(const u32 *sp_off, const u32 *sp_sample, u32 v, u32 s)
{
    u32 lo = sp_off[v];
    const u32 end = sp_off[v + 1];
    u32 hi = end;
    while (lo < hi) {
        const u32 mid = lo + (hi - lo) / 2;
        if (sp_sample[mid] < s) lo = mid + 1;
        else hi = mid;
    }
    return lo < end && sp_sample[lo] == s ? (long long)lo : -1;
}
__device__ __forceinline__ u32 gt_at(const BlockBatch &B, u32 v, u32 s)
{
    if (!B.sp_off) return B.gt[(u64)v * B.n_samples + s];
    const long long e = sp_find(B.sp_off, B.sp_sample, v, s);
    return e >= 0 ? (u32)B.sp_gt[e] : B.sp_default;
}
constexpr int BK_MAXC = 16;  // chains per side
constexpr int BK_MAXL = 32;  // members per chain side (a combined chain: left + the variant + right <= 65 members)
constexpr int BK_MAXU = 14;  // unphased chain length (2^14 haplotype mixes per sample)
constexpr int BK_CODE_BITS = 63; // a chain's pick as a code: bits per member = ceil(log2(alleles))
constexpr int BK_SET_CAP = 2048; // distinct picks of one chain held in the LDS set (16 KB); more: every pick evaluated directly

struct BkChains {
    int n;
    int len[BK_MAXC];
    int sum[BK_MAXC];
    int mem[BK_MAXC][BK_MAXL];
};

// get_combs_on_the_right (step +1) / _left (step -1); indices are batch-global variant indices inside [b0, b1)
// VB::are_near (var_block.hpp:417-423) in the reference's arithmetic.  It writes  int + ... + ceil((float)k / 2) >= int
// under `using namespace std`: ceil is the float overload, so the int sum is CONVERTED TO FLOAT, the addition rounds to
// float and the right side is compared as a float.  Exact below 2^24; beyond (most of a human chromosome) positions are
// rounded to multiples of 2..16 and the answer differs from the exact one now and then, in both directions.
__device__ __forceinline__ bool near_f32(int lhs_sum, int k, int rhs_pos) { return (float)lhs_sum + ceilf((float)k / 2) >= (float)rhs_pos; }

// ---- block cutting on the device (main.cpp:341, 547 with var_block.hpp:77-80) -------------------------------------------
// The record loops close a block when the next kept record is not near the block's last one (VB::is_near_to_last: only the
// LAST variant counts, sum_to_add = 0) or sits on another sequence than `last_seq_name`.  `last_seq_name` is the name of
// the file's first record until the first flush and the previous kept record's name from then on (it is refreshed at every
// flush that sees a new name, and a new name always flushes), so with contig[0] = that first name the test is local:
//   cut[i] = contig[i] != contig[i-1]  ||  !are_near(record i-1, record i)          (cut[0] = 1)
// and the blocks are the runs between cuts.  One thread per record.
__global__ void __launch_bounds__(TPB) cut_flags_kernel(u64 n, const int *__restrict__ pos, const u32 *__restrict__ ref_size, const u32 *__restrict__ min_size,
                                                        const u32 *__restrict__ contig, int k, u8 *__restrict__ cut, u32 *__restrict__ tile_sums)
{
    __shared__ u32 sh_n;
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    const u64 i = (u64)blockIdx.x * TPB + threadIdx.x;
    bool f = false;
    if (i < n) {
        f = i == 0 || contig[i] != contig[i - 1] || !near_f32(pos[i - 1] + (int)ref_size[i - 1] - (int)min_size[i - 1] - 1, k, pos[i]);
        cut[i] = f;
    }
    const u64 mask = __ballot(f);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(&sh_n, (u32)__popcll(mask));
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = sh_n;
}
// flags -> per-tile counts (for flags that did not come from cut_flags_kernel)
__global__ void __launch_bounds__(TPB) flag_count_kernel(u64 n, const u8 *__restrict__ flags, u32 *__restrict__ tile_sums)
{
    __shared__ u32 sh_n;
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    const u64 i = (u64)blockIdx.x * TPB + threadIdx.x;
    const u64 mask = __ballot(i < n && flags[i]);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(&sh_n, (u32)__popcll(mask));
    __syncthreads();
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = sh_n;
}
// heads of the blocks of an existing cut: flags[blk_var_off[b]] = 1
__global__ void __launch_bounds__(TPB) block_heads_kernel(const u32 *__restrict__ blk_var_off, const unsigned long long *__restrict__ n_blocks, u64 n, u8 *__restrict__ flags)
{
    const u64 b = (u64)blockIdx.x * TPB + threadIdx.x;
    if (b < *n_blocks && blk_var_off[b] < n) flags[blk_var_off[b]] = 1;
}
// With the tiles' counts scanned (launch_tile_scan): blk_var_off[b] = first record of block b, blk_var_off[n_blocks] = n,
// var_block[i] = block of record i.  Any of the three outputs may be NULL.
__global__ void __launch_bounds__(TPB) flag_scatter_kernel(u64 n, const u8 *__restrict__ flags, const u32 *__restrict__ tile_base, u32 *__restrict__ blk_var_off,
                                                           u32 *__restrict__ var_block, unsigned long long *n_blocks_out)
{
    __shared__ u32 sh_wave[TPB / 64];
    const u64 i = (u64)blockIdx.x * TPB + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool f = i < n && flags[i];
    const u64 mask = __ballot(f);
    if (lane == 0) sh_wave[wave] = (u32)__popcll(mask);
    __syncthreads();
    u32 before = tile_base[blockIdx.x];
    for (int w = 0; w < wave; ++w) before += sh_wave[w];
    const u32 incl = before + (u32)__popcll(mask & ((2ULL << lane) - 1)); // flags in [tile start, i]
    if (i < n) {
        if (f && blk_var_off) blk_var_off[incl - 1] = (u32)i;
        if (var_block) var_block[i] = incl - 1;
        if (i == n - 1) {
            if (blk_var_off) blk_var_off[incl] = (u32)n;
            if (n_blocks_out) *n_blocks_out = incl;
        }
    }
}

// a resident panel as the kernels see it (include/malva_hip.h: mg_panel_dev)
struct PanelView {
    const u64 *contig_base;
    const u32 *contig_len, *contig_id;
    const i32 *pos;
    const u32 *ref_size, *min_size;
    const u8 *present;
    const u32 *var_allele_off, *allele_off;
    const u8 *canon;
    const uint16_t *gt;
    const u32 *sp_off, *sp_sample; // sparse genotypes (see BlockBatch), or NULL
    const uint16_t *sp_gt;
    u32 sp_default;
    u32 n_samples;
};

__device__ bool bk_chains(const BlockBatch &B, int b0, int b1, int i, int step, bool sorted, int max_gain, BkChains *out)
{
    const int k = B.k;
    auto ov = [&](int x, int y) { // overlapping(left, right) with (x, y) given in scan order
        const int l = step > 0 ? x : y, r = step > 0 ? y : x;
        return B.pos[l] <= B.pos[r] && B.pos[r] < B.pos[l] + (int)B.ref_size[l];
    };
    auto nr = [&](int x, int y, int extra) {
        const int l = step > 0 ? x : y, r = step > 0 ? y : x;
        return near_f32(B.pos[l] + (int)B.ref_size[l] - (int)B.min_size[l] - 1 + extra, k, B.pos[r]);
    };
    out->n = 0;
    bool halt = false;
    // The reference walks to the end of the block whatever happens (var_block.hpp:436-525: O(B) per variant, O(B^2) per
    // block).  Nothing can join a chain once the walk is beyond the reach of every chain -- positions only move away and
    // a chain's reach grows only when something joins -- so with sorted positions the walk stops there: same chains.
    // (`sorted`: positions never decrease along the block, `max_gain`: its largest ref_size - min_size -- the workgroup looks)
    for (int j = i + step; j >= b0 && j < b1 && !halt; j += step) {
        if (sorted) {
            int max_sum = 0;
            for (int c = 0; c < out->n; ++c) max_sum = max(max_sum, out->sum[c]);
            // (the same float test as `nr`, on the largest left side any chain can still present: it is monotone in both
            // arguments, so beyond the first position it rejects it rejects everything)
            if (step > 0 ? !near_f32(B.pos[i] + (int)B.ref_size[i] - (int)B.min_size[i] - 1 + max_sum, k, B.pos[j])
                         : !near_f32(B.pos[j] + max_gain - 1 + max_sum, k, B.pos[i]))
                break;
        }
        if (!B.present[j]) continue;
        if (ov(i, j)) continue;
        const int gain = (int)B.ref_size[j] - (int)B.min_size[j];
        if (out->n == 0) {
            if (nr(i, j, 0)) {
                out->mem[0][0] = j;
                out->len[0] = 1;
                out->sum[0] = gain;
                out->n = 1;
            }
            continue;
        }
        bool added = false;
        const int n0 = out->n;
        for (int c = 0; c < n0; ++c) {
            if (!ov(out->mem[c][out->len[c] - 1], j)) {
                added = true;
                if (nr(i, j, out->sum[c])) {
                    if (out->len[c] >= BK_MAXL) return false;
                    out->mem[c][out->len[c]++] = j;
                    out->sum[c] += gain;
                }
            }
        }
        if (!added) {
            for (int c = 0; c < n0; ++c) {
                int len = out->len[c], ns = out->sum[c];
                while (len > 0 && ov(out->mem[c][len - 1], j)) {
                    const int m = out->mem[c][len - 1];
                    ns -= (int)B.ref_size[m] - (int)B.min_size[m];
                    --len;
                }
                if (nr(i, j, ns)) {
                    added = true;
                    if (out->n >= BK_MAXC || len + 1 > BK_MAXL) return false;
                    const int d = out->n++;
                    for (int q = 0; q < len; ++q) out->mem[d][q] = out->mem[c][q];
                    out->mem[d][len] = j;
                    out->len[d] = len + 1;
                    out->sum[d] = ns + gain;
                }
            }
            if (!added) halt = true;
        }
    }
    return true;
}

struct LdsBytes {
    const u8 *p;
    __device__ __forceinline__ u32 operator()(int i) const { return p[i]; }
};
// weight of the k-mer in buf[0, len): packed fast path when it is k pure-ACGT bases, byte-wise otherwise
__device__ __forceinline__ i32 bk_weight(const u8 *buf, int len, bool is_ref, const BFView &bf, const MapView &map)
{
    if (len >= 17 && len <= MG_MAX_PACKED_K) {
        U128 L{0, 0};
        bool ok = true;
        for (int i = 0; i < len; ++i) {
            bool o;
            const u64 code = acgt_code(buf[i], &o);
            ok &= o;
            if (i < 32) L.lo |= code << (2 * i);
            else L.hi |= code << (2 * (i - 32));
        }
        if (ok) {
            const U128 mk = mask128(2 * len);
            const U128 mform = shr128(U128{pairrev64(L.hi), pairrev64(L.lo)}, 2 * (64 - len));
            const U128 rc{~mform.lo & mk.lo, ~mform.hi & mk.hi};
            const U128 key = lt128(L, rc) ? L : rc;
            const u64 h = xxh3_packed(key, len);
            if (is_ref) {
                if (len != (int)map.klen) return 0;
                return map_value(map, key, h, mod_size(h, bf.mod));
            }
            const u64 idx = mod_size(h, bf.mod);
            return (i32)(uint16_t)bf_count_at(bf, idx);
        }
    }
    return weight_bytes(LdsBytes{buf}, len, is_ref, bf, map);
}

// index time (add_kmers_to_bf, main.cpp:122-144): the k-mer in buf[0, len) goes into the exact map (REF allele) or sets
// its bit of `bf` (any other allele).  MODE 1 only counts the REF k-mers (the host sizes the map before MODE 2 inserts).
// false: a REF k-mer the packed table cannot hold (a non-ACGT base, or a clipped window: KMAP::canonical's NUL
// truncation, kmap.hpp:95) -- the variant's block goes to the host enumerator.
struct IndexEmit {
    u32 *sh_emit;               // REF k-mers of this variant (MODE 1)
    unsigned long long *cursor; // insertion rows handed out so far in this batch (MODE 2)
    u32 row0;
};
template <int MODE> __device__ __forceinline__ bool bk_index_emit(const u8 *buf, int len, bool is_ref, const BFView &bf, const MapView &map, const IndexEmit &e)
{
    U128 L{0, 0};
    bool ok = len <= MG_MAX_PACKED_K;
    if (ok)
        for (int i = 0; i < len; ++i) {
            bool o;
            const u64 code = acgt_code(buf[i], &o);
            ok &= o;
            if (i < 32) L.lo |= code << (2 * i);
            else L.hi |= code << (2 * (i - 32));
        }
    U128 key{0, 0};
    if (ok) {
        const U128 mk = mask128(2 * len);
        const U128 mform = shr128(U128{pairrev64(L.hi), pairrev64(L.lo)}, 2 * (64 - len));
        const U128 rc{~mform.lo & mk.lo, ~mform.hi & mk.hi};
        key = lt128(L, rc) ? L : rc;
    }
    if (is_ref) {
        if (!ok || len != (int)map.klen) return false;
        if (MODE == 1) {
            atomicAdd(e.sh_emit, 1u);
            return true;
        }
        const u32 my_id = e.row0 + (u32)atomicAdd(e.cursor, 1ULL);
        map_insert_key(map, bf, key, xxh3_lform(key, len), my_id, e.row0);
        return true;
    }
    if (MODE == 1) return true;
    const u64 idx = mod_size(ok ? xxh3_lform(key, len) : xxh3_bytes(CanonBytes<LdsBytes>(LdsBytes{buf}, len), len), bf.mod);
    atomicOr((unsigned long long *)&bf.words[idx >> 6], 1ULL << (idx & 63)); // BF::add_key, bloom_filter.hpp:81-85
    gate_set(bf, idx);
    return true;
}

// MODE 0: call time, coverage of every allele (set_coverages).  MODE 1 / 2: index time, see bk_index_emit; `overflow` then
// carries MODE 1's verdict into MODE 2 (a variant flagged there is skipped here and left to the host).
// The grid is persistent: workgroup w takes the records list[w], list[w + gridDim.x], ... of `list` (*list_n entries:
// the records the faster tiers handed on), so that neither the list's length nor the launch of millions of workgroups
// needs the host.
template <int MODE>
__global__ void __launch_bounds__(TPB) cover_blocks_kernel(BlockBatch B, const u32 *__restrict__ list, const unsigned long long *__restrict__ list_n, BFView bf,
                                                           MapView map, u32 *cov_out, u8 *overflow, IndexEmit emit, unsigned long long *n_evaluated)
{
    __shared__ BkChains sh_left, sh_right;
    __shared__ int sh_bad, sh_eligible, sh_unsorted, sh_max_gain;
    __shared__ u32 sh_cov[128];
    __shared__ u32 sh_slide[4]; // alleles (bit mask, 128 bits) that some sample carries alone and whole (len >= k)
    __shared__ u8 sh_buf[TPB][MG_MAX_PACKED_K];
    __shared__ unsigned long long sh_set[BK_SET_CAP]; // distinct picks of the chain in hand: code + 1, 0 = free
    __shared__ u32 sh_set_n;
    __shared__ u32 sh_emit;
    emit.sh_emit = &sh_emit;
    u32 evaluated = 0; // signature k-mers this thread assembled and looked up (or inserted)
    const u64 n_list = *list_n;
    for (u64 item = blockIdx.x; item < n_list; item += gridDim.x) {
    __syncthreads(); // the previous record's shared state is done with
    const int g = (int)list[item];
    if (MODE == 2 && overflow[g]) continue; // flagged by the counting pass: the host enumerates this variant's block
    if (threadIdx.x == 0) sh_emit = 0;
    const u32 a0 = B.var_allele_off[g], A = B.var_allele_off[g + 1] - a0;
    const u32 blk = B.var_block[g];
    const int b0 = (int)B.blk_var_off[blk], b1 = (int)B.blk_var_off[blk + 1];
    const u32 cid = B.contig_id[b0];
    const u8 *ref = B.reference + B.contig_base[cid];
    const i32 ref_len = (i32)B.contig_len[cid];
    const int k = B.k;
    for (u32 a = threadIdx.x; a < 128; a += TPB) sh_cov[a] = 0;
    if (threadIdx.x < 4) sh_slide[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        sh_bad = (A > 127 || k > MG_MAX_PACKED_K) ? 1 : 0;
        sh_eligible = B.present[g] && B.pos[g] >= k && B.pos[g] <= ref_len - k; // var_block.hpp:104
        sh_left.n = sh_right.n = 0;
        sh_unsorted = 0;
        sh_max_gain = 0;
    }
    __syncthreads();
    // what bounds the walks: is the block sorted, and its largest ref_size - min_size (all threads look)
    if (sh_eligible && !sh_bad) {
        int mg_ = 0, un = 0;
        for (int j = b0 + (int)threadIdx.x; j < b1; j += TPB) {
            if (B.ref_size[j] > B.min_size[j]) mg_ = max(mg_, (int)(B.ref_size[j] - B.min_size[j]));
            if (j > b0 && B.pos[j] < B.pos[j - 1]) un = 1;
        }
        if (mg_) atomicMax(&sh_max_gain, mg_);
        if (un) sh_unsorted = 1;
    }
    __syncthreads();
    // the two walks (get_combs_on_the_left / _right) are independent: one lane of wave 0 and one of wave 1 take one each
    if (sh_eligible && !sh_bad && (threadIdx.x == 0 || threadIdx.x == 64))
        if (!bk_chains(B, b0, b1, g, threadIdx.x == 0 ? -1 : +1, !sh_unsorted, sh_max_gain, threadIdx.x == 0 ? &sh_left : &sh_right)) atomicOr(&sh_bad, 1);
    __syncthreads();
    if (sh_bad) {
        if (threadIdx.x == 0) overflow[g] = 1;
        if (MODE == 0)
            for (u32 a = threadIdx.x; a < A; a += TPB) cov_out[a0 + a] = 0;
        continue;
    }
    u8 *buf = sh_buf[threadIdx.x];
    bool bad = false;
    // combine_combs (var_block.hpp:630-677): every left chain (reversed) + the variant + every right chain; a chain is
    // read member by member straight from the two walks' results, nothing is materialised
    const int nl = sh_eligible ? (sh_left.n ? sh_left.n : 1) : 0, nrr = sh_right.n ? sh_right.n : 1;
    for (int c = 0; c < nl * nrr; ++c) {
        const int cl = c / nrr, cr = c % nrr;
        const int len_l = sh_left.n ? sh_left.len[cl] : 0, len_r = sh_right.n ? sh_right.len[cr] : 0;
        const int m = len_l + 1 + len_r, jm = len_l;
        struct {
            const BkChains *L, *R;
            int cl, cr, len_l, g;
            __device__ __forceinline__ int operator[](int j) const
            {
                return j < len_l ? L->mem[cl][len_l - 1 - j] : j == len_l ? g : R->mem[cr][j - len_l - 1];
            }
        } comb{&sh_left, &sh_right, cl, cr, len_l, g};
        const int first_pos = B.pos[comb[0]];
        const int last_end = B.pos[comb[m - 1]] + (int)B.ref_size[comb[m - 1]];
        // one haplotype pick along the chain -> its signature k-mer -> weight -> max into the mid allele's coverage;
        // allele_of(j) = allele of member j under the pick
        auto evaluate = [&](auto allele_of) {
            // lengths: virtual string V = A_0 R_0 A_1 ... A_{m-1}
            int len_v = 0, mid_pos = 0, mid_len = 0;
            u32 mid_allele = 0;
            for (int j = 0; j < m; ++j) {
                const u32 slot = B.var_allele_off[comb[j]] + allele_of(j);
                const int al = (int)(B.allele_off[slot + 1] - B.allele_off[slot]);
                if (j == jm) {
                    mid_pos = len_v;
                    mid_len = al;
                    mid_allele = allele_of(j);
                }
                len_v += al;
                if (j + 1 < m) len_v += B.pos[comb[j + 1]] - (B.pos[comb[j]] + (int)B.ref_size[comb[j]]);
            }
            const u32 mid_canon = B.canon[a0 + mid_allele];
            if (m == 1 && mid_len >= k) { // the whole allele is the signature: sliding k-mers, done below
                atomicOr(&sh_slide[mid_canon >> 5], 1u << (mid_canon & 31));
                return;
            }
            const int first_part = mid_pos + mid_len / 2;
            const int mp = k / 2 - first_part;                  // missing_prefix (negative: cut)
            const int ms = (k + 1) / 2 - (len_v - first_part);  // missing_suffix
            if (first_pos - (mp > 0 ? mp : 0) < 0 || last_end + (ms > 0 ? ms : 0) > ref_len) {
                bad = true; // the reference clips or throws here: leave it to the host path
                return;
            }
            // W[x] = Vext[x - mp] for x in [0, k), where Vext is V with the reference continuing on both sides:
            // a piece that covers v in [vs, vs + L) lands at x in [vs + mp, vs + L + mp), clipped to the window
            for (int x = 0; x < mp && x < k; ++x) buf[x] = ref[first_pos - mp + x];
            int vs = 0;
            for (int j = 0; j < m; ++j) {
                const u32 slot = B.var_allele_off[comb[j]] + allele_of(j);
                const u8 *ap = B.pool + B.allele_off[slot];
                const int al = (int)(B.allele_off[slot + 1] - B.allele_off[slot]);
                for (int x = max(0, vs + mp), xe = min(k, vs + al + mp); x < xe; ++x) buf[x] = ap[x - mp - vs];
                vs += al;
                if (j + 1 < m) {
                    const int gs = B.pos[comb[j]] + (int)B.ref_size[comb[j]];
                    const int gl = B.pos[comb[j + 1]] - gs;
                    for (int x = max(0, vs + mp), xe = min(k, vs + gl + mp); x < xe; ++x) buf[x] = ref[gs + (x - mp - vs)];
                    vs += gl;
                }
            }
            for (int x = max(0, len_v + mp); x < k; ++x) buf[x] = ref[last_end + (x - mp - len_v)];
            ++evaluated;
            if (MODE == 0) {
                const i32 w = bk_weight(buf, k, mid_canon == 0, bf, map);
                if (w > 0) atomicMax(&sh_cov[mid_canon], (u32)w);
            } else if (!bk_index_emit<MODE>(buf, k, mid_canon == 0, bf, map, emit))
                bad = true;
        };
        // The picks of all samples are first reduced to the DISTINCT ones (the reference's unordered_set, var_block.hpp
        // :734-786): a pick is a code of a few bits per member, collected in an LDS hash set by the pass over the
        // samples and evaluated once by the pass over the set.  A panel has thousands of samples and a chain a handful
        // of distinct picks.  A chain whose code does not fit 63 bits, or with more than BK_SET_CAP / 2 distinct picks,
        // has every sample's picks evaluated directly.
        auto bits_of = [&](int j) -> int { // width of member j's field in the code
            const u32 A_j = B.var_allele_off[comb[j] + 1] - B.var_allele_off[comb[j]];
            return A_j <= 2 ? 1 : 32 - __clz((int)A_j - 1);
        };
        int code_bits = 0;
        for (int j = 0; j < m; ++j) code_bits += bits_of(j);
        const bool coded = code_bits <= BK_CODE_BITS;
        // every sample's picks: into the set (collect) or straight to evaluate().  Dense genotypes: the threads stride over the
        // samples.  Sparse genotypes: over the ENTRIES of the chain's members -- a sample is taken at the first member it has
        // an entry in -- plus once the sample with no entry anywhere (all 0|0; there is one whenever the entries are fewer
        // than the samples); a chain with as many entries as samples walks the samples and looks each genotype up.
        auto walk_samples = [&](bool collect) {
            auto one_sample = [&](auto gt_of) -> bool { // gt_of(j): member j's genotype word; false: the set overflowed, stop
                bool phased = true;
                if (!B.haploid)
                    for (int j = 0; j < m; ++j) phased = phased && ((gt_of(j) >> 14) & 1);
                const u32 npick = B.haploid ? 1u : phased ? 2u : (1u << m);
                if (!B.haploid && !phased && m > BK_MAXU) {
                    bad = true;
                    return true;
                }
                for (u32 pick = 0; pick < npick; ++pick) {
                    auto allele_of = [&](int j) -> u32 { // allele of member j under this pick
                        const u32 gt = gt_of(j);
                        const u32 a1 = gt & 127, a2 = (gt >> 7) & 127;
                        if (B.haploid) return a1;
                        if (phased) return pick ? a2 : a1;
                        return (pick >> j) & 1 ? a2 : a1;
                    };
                    if (!collect) {
                        evaluate(allele_of);
                        continue;
                    }
                    unsigned long long code = 0;
                    int sh = 0;
                    for (int j = 0; j < m; ++j) {
                        code |= (unsigned long long)allele_of(j) << sh;
                        sh += bits_of(j);
                    }
                    u32 at = (u32)((code * 0x9E3779B97F4A7C15ULL) >> 40) & (BK_SET_CAP - 1);
                    for (int tries = 0; tries < BK_SET_CAP; ++tries) {
                        const unsigned long long seen = atomicCAS(&sh_set[at], 0ULL, code + 1);
                        if (seen == 0ULL) {
                            atomicAdd(&sh_set_n, 1u);
                            break;
                        }
                        if (seen == code + 1) break;
                        at = (at + 1) & (BK_SET_CAP - 1);
                    }
                    if (sh_set_n > B.set_limit) return false; // too many distinct picks: this chain is redone directly
                }
                return true;
            };
            u32 entries = 0;
            if (B.sp_off)
                for (int j = 0; j < m; ++j) entries += B.sp_off[comb[j] + 1] - B.sp_off[comb[j]];
            if (!B.sp_off || entries >= B.n_samples) {
                for (u32 s = threadIdx.x; s < B.n_samples; s += TPB)
                    if (!one_sample([&](int j) -> u32 { return gt_at(B, (u32)comb[j], s); })) return;
                return;
            }
            for (u32 idx = threadIdx.x; idx < entries; idx += TPB) {
                u32 rem = idx;
                int j = 0;
                for (; j < m; ++j) {
                    const u32 cnt = B.sp_off[comb[j] + 1] - B.sp_off[comb[j]];
                    if (rem < cnt) break;
                    rem -= cnt;
                }
                const u32 e = B.sp_off[comb[j]] + rem, smp = B.sp_sample[e];
                bool first = true;
                for (int q = 0; q < j && first; ++q) first = sp_find(B.sp_off, B.sp_sample, (u32)comb[q], smp) < 0;
                if (!first) continue;
                if (!one_sample([&](int q) -> u32 { return q == j ? (u32)B.sp_gt[e] : gt_at(B, (u32)comb[q], smp); })) return;
            }
            // (every member default: one pick if the default is homozygous, whatever its phase bit says -- the mixes of equal alleles are equal)
            if (threadIdx.x == 0) one_sample([&](int) -> u32 { return (B.sp_default & 127) == ((B.sp_default >> 7) & 127) ? B.sp_default | (1u << 14) : B.sp_default; });
        };
        if (coded) {
            for (u32 w = threadIdx.x; w < BK_SET_CAP; w += TPB) sh_set[w] = 0;
            if (threadIdx.x == 0) sh_set_n = 0;
            __syncthreads();
            walk_samples(true);
            __syncthreads();
            if (sh_set_n <= B.set_limit) {
                for (u32 w = threadIdx.x; w < BK_SET_CAP; w += TPB) {
                    if (!sh_set[w]) continue;
                    const unsigned long long code = sh_set[w] - 1;
                    evaluate([&](int j) -> u32 {
                        int sh = 0;
                        for (int q = 0; q < j; ++q) sh += bits_of(q);
                        return (u32)(code >> sh) & ((1u << bits_of(j)) - 1);
                    });
                }
            } else
                walk_samples(false);
            __syncthreads(); // the set is reused by the next chain
        } else
            walk_samples(false);
    }
    if (bad) sh_bad = 1;
    __syncthreads();
    // sliding signatures of lone long alleles (var_block.hpp:130-144): truncating running mean over the allele's k-mers
    for (u32 a = threadIdx.x; a < A; a += TPB) {
        if (!((sh_slide[a >> 5] >> (a & 31)) & 1)) continue;
        const u8 *ap = B.pool + B.allele_off[a0 + a];
        const int al = (int)(B.allele_off[a0 + a + 1] - B.allele_off[a0 + a]);
        u32 curr = 0;
        i32 n = 0;
        for (int p = 0; p + k <= al; ++p) {
            for (int x = 0; x < k; ++x) buf[x] = ap[p + x];
            ++evaluated;
            if (MODE != 0) {
                if (!bk_index_emit<MODE>(buf, k, a == 0, bf, map, emit)) sh_bad = 1;
                continue;
            }
            const i32 w = bk_weight(buf, k, a == 0, bf, map);
            if (w > 0) {
                curr = (curr * (u32)n + (u32)w) / (u32)(n + 1);
                ++n;
            }
        }
        if (MODE == 0) atomicMax(&sh_cov[a], curr);
    }
    __syncthreads();
    if (MODE != 2 && threadIdx.x == 0) overflow[g] = sh_bad ? 1 : 0;
    if (MODE == 1 && threadIdx.x == 0 && !sh_bad && sh_emit) atomicAdd(emit.cursor, (unsigned long long)sh_emit); // rows the insert pass will need
    if (MODE == 0)
        for (u32 a = threadIdx.x; a < A; a += TPB) cov_out[a0 + a] = sh_bad ? 0 : (u32)(float)sh_cov[a];
    } // next record of the list
    if (n_evaluated) {
        for (int d = 32; d; d >>= 1) evaluated += __shfl_xor(evaluated, d, 64);
        if ((threadIdx.x & 63) == 0 && evaluated) atomicAdd(n_evaluated, (unsigned long long)evaluated);
    }
}

