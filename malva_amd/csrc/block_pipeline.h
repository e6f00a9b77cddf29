// block_pipeline.h -- the record loop's per-block work on a resident panel, as tiers of flat kernels.
// Part of the malva_hip translation unit: included by malva_hip.hip inside its anonymous namespace, after variant_kernels.h.
//
// VB::extract_kmers (var_block.hpp:95-219) with its helpers (:436-786) + set_coverages (main.cpp:151-184) / add_kmers_to_bf
// (main.cpp:122-144) for every record of a panel whose blocks are cut.  cover_blocks_kernel (variant_kernels.h) gives a
// whole workgroup to one record and walks its chains on one lane: right for a panel of tens of thousands of samples and
// blocks of thousands of records, 70-130 us per record for a cluster of four SNPs and two samples.  Most records of a
// real panel are of the second kind, so the work is dealt out by what it is made of instead:
//
//   tier 1  panel_lone_kernel         a block of ONE variant with short alleles (nine records in ten): classification fused
//                                     with the lone-variant lookup of mg_call_isolated, two threads per record; every
//                                     other record is appended to the general list
//   tier 2  fw_walk_kernel            one THREAD per general record: the two chain walks (get_combs_on_the_left / _right)
//                                     in private memory, every (left, right) pair written out as a 32-byte descriptor
//           fw_picks_kernel           one WAVE per descriptor, lanes over the panel's samples: the distinct haplotype picks
//                                     along the chain (build_alleles_combs' unordered_set) in a per-wave LDS set, written
//                                     out as (descriptor, pick) items
//           fw_eval_kernel            one THREAD per item: assemble the signature k-mer in 2-bit form, canonical, XXH3,
//                                     lookup (or insert), atomicMax into the allele's coverage
//   tier 3  cover_blocks_kernel       whatever exceeds a capacity of tier 2 (chains, reach, code width, distinct picks,
//                                     a base outside ACGT, an allele of k bases or more on its own, more than FW_MAX_SAMPLES
//                                     samples), from a compacted list; what exceeds ITS capacities is flagged for the host
//
// Every count lives on the device: the general list is processed in rounds of a fixed number of records whose buffers
// are sized by the round, rounds beyond the list's end find nothing to do, and no launch waits for the host.
#pragma once

constexpr int FW_MAXC = 6;           // chains per side
constexpr int FW_MAXM = 10;          // members per chain side
constexpr int FW_REACH = 120;        // records a walk may move away from its variant (members are stored as int8 offsets)
constexpr int FW_SET = 512;          // slots of a wave's LDS set of distinct picks (at most 3/4 used)
constexpr int FW_MAXU = 12;          // unphased chain length (2^12 mixes per sample)
constexpr u32 FW_MAX_SAMPLES = 512;  // larger panels take the workgroup kernel (its 256 threads stride over the samples)
constexpr int FW_COMBS_PER_REC = 8;  // descriptors a round's buffer holds per record of the round (average; a record may use 36)
constexpr int FW_ITEMS_PER_REC = 32; // items likewise
constexpr int FW_ORD_BINS = 24;      // chain lengths 0 (a reservation that did not fit) .. 21, rounded up
constexpr int FW_ORD_HIST = 8, FW_ORD_CUR = FW_ORD_HIST + FW_ORD_BINS; // places in a round's counter block
constexpr int FW_ROUND_COUNTERS = FW_ORD_CUR + FW_ORD_BINS;           // u64 per round

struct FwSide {
    int n;
    int len[FW_MAXC];
    int sum[FW_MAXC];
    signed char mem[FW_MAXC][FW_MAXM]; // offsets from the variant
};
struct __attribute__((aligned(16))) CombDesc {
    u32 g;   // the variant the chain is built around
    u32 cid; // sequence its block is evaluated against
    u8 m, jm; // members, index of g among them
    signed char rel[22]; // members as offsets from g, left to right
};
static_assert(sizeof(CombDesc) == 32, "one descriptor per 32 bytes");
struct __attribute__((aligned(16))) PickItem {
    u32 comb;
    u32 pad;
    u64 code; // allele of member j at bits [shift_j, shift_j + bits_j), bits_j = ceil(log2(alleles of member j)) (1 for two)
};

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// One insertion row for every lane that calls (any subset of the wave, together): one returning atomic per wave.  A lane each
// on ONE counter serialises at ~3 ns a row -- 14 ms of `index`'s insert pass per 4.4e6 REF k-mers (C5).
__device__ __forceinline__ u32 wave_take(unsigned long long *cursor)
{
    const u64 callers = __ballot(true);
    const int lane = threadIdx.x & 63, leader = __ffsll((unsigned long long)callers) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(cursor, (unsigned long long)__popcll(callers));
    base = __shfl(base, leader, 64);
    return (u32)base + (u32)__popcll(callers & ((1ULL << lane) - 1));
}
__device__ __forceinline__ int fw_bits(u32 n_alleles) { return n_alleles <= 2 ? 1 : 32 - __clz((int)n_alleles - 1); }

// ---- tier 1 ------------------------------------------------------------------------------------------------------------------
// counters: [0] general records listed, [2] signature k-mers of the lone records
struct LoneClass {
    bool lone, eligible;
    u64 site, mask;
    u32 a0, A;
    unsigned short cls; // what fw_walk_kernel and fw_snp_kernel want to know of a record outside tier 1, see REC_*
};
// REC_SNP: one base for one base, every allele; REC_PHASED: every sample's genotype phased (or the run is haploid); REC_CODES: at
// most four alleles, all of them A, C, G or T, and their 2-bit codes in the high byte (allele a at bits 8 + 2a)
constexpr unsigned short REC_SNP = 1, REC_PHASED = 2, REC_CODES = 4;
// The loads are arranged in three LEVELS of mutually independent requests (what a thread of this kernel does is wait for
// loads: the first form, a chain of fifteen dependent ones -- block, block ends, its sequence, the sequence's base; genotype
// word, canonical allele, next genotype word ... -- ran at a quarter of the rate its 2.5 lines of HBM traffic per record allow):
//   1  addressed by v alone: block number, sequence, position, sizes, allele range, presence flag, genotype words / entry range
//   2  addressed by those: the block's ends, the sequence's base and length, allele offsets, canonical alleles, genotype entries
//   3  the flanks and allele bytes (iso_cover_body), then the records
// The genotype words give a mask of RAW allele numbers; the canonical numbers (variant.hpp:228-240) are applied to the mask
// afterwards, so their loads wait for nothing but the allele range.
__device__ __forceinline__ LoneClass classify_lone(const PanelView &P, const u32 *blk_var_off, const u32 *var_block, u64 v, int k, int haploid, const u8 *pool)
{
    LoneClass c{};
    // level 1
    const u32 blk = var_block[v];
    const u32 cid = P.contig_id[v]; // (of the block's first record in general; a lone record IS its block's first)
    const i32 p = P.pos[v];
    const u32 rs = P.ref_size[v];
    c.a0 = P.var_allele_off[v];
    c.A = P.var_allele_off[v + 1] - c.a0;
    const bool present = P.present[v] != 0;
    u64 raw = 0; // raw allele numbers some panel haplotype carries (numbers of 64 and more fold back: such a record is not lone)
    u32 e0 = 0, e1 = 0;
    bool phased = true;
    if (P.sp_off) {
        e0 = P.sp_off[v];
        e1 = P.sp_off[v + 1];
    } else {
        const uint16_t *g = P.gt + v * P.n_samples;
        for (u32 s = 0; s < P.n_samples; s += 4) { // four words requested together
            u32 w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = s + j < P.n_samples ? (u32)g[s + j] : 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (w[j] != 0xFFFFFFFFu) {
                    raw |= 1ULL << (w[j] & 63);
                    if (!haploid) raw |= 1ULL << ((w[j] >> 7) & 63);
                    phased = phased && ((w[j] >> 14) & 1);
                }
        }
    }
    // level 2
    const u32 b0 = blk_var_off[blk], b1 = blk_var_off[blk + 1];
    const u64 cbase = P.contig_base[cid];
    const u32 clen = P.contig_len[cid];
    u32 ao[5];
    u8 cn[4];
#pragma unroll
    for (int a = 0; a < 5; ++a) ao[a] = a <= (int)c.A ? P.allele_off[c.a0 + a] : 0u;
#pragma unroll
    for (int a = 0; a < 4; ++a) cn[a] = a < (int)c.A ? P.canon[c.a0 + a] : (u8)0;
    if (P.sp_off) { // sparse genotypes: the entries, and the default word for the samples that have none
        for (u32 e = e0; e < e1; ++e) {
            const u32 g = P.sp_gt[e];
            raw |= 1ULL << (g & 63);
            if (!haploid) raw |= 1ULL << ((g >> 7) & 63);
            phased = phased && ((g >> 14) & 1);
        }
        if (e1 - e0 < P.n_samples) {
            raw |= 1ULL << (P.sp_default & 63);
            if (!haploid) raw |= 1ULL << ((P.sp_default >> 7) & 63);
            phased = phased && ((P.sp_default >> 14) & 1);
        }
    }
    // what fw_walk_kernel wants to know of a chain's members before it calls the chain one of SNPs (fw_snp_kernel takes those whole)
    bool snp = rs == 1;
#pragma unroll
    for (int a = 0; a < 4; ++a)
        if (a < (int)c.A) snp = snp && ao[a + 1] - ao[a] == 1;
    for (u32 a = 4; a < c.A && snp; ++a) snp = snp && P.allele_off[c.a0 + a + 1] - P.allele_off[c.a0 + a] == 1;
    c.cls = (unsigned short)((snp ? REC_SNP : 0) | (phased || haploid ? REC_PHASED : 0));
    // a block of one variant, alleles all shorter than k (at most 64 of them: the presence mask), flanks inside the sequence
    bool lone = b1 - b0 == 1 && c.A <= 64 && p >= k / 2 && (long long)p + rs + (k + 1) / 2 <= (long long)clen;
    if (lone) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a < (int)c.A) lone = lone && (int)(ao[a + 1] - ao[a]) < k;
        for (u32 a = 4; a < c.A; ++a) lone = lone && (int)(P.allele_off[c.a0 + a + 1] - P.allele_off[c.a0 + a]) < k;
    }
    c.lone = lone;
    if (!lone) {
        if (snp && c.A <= 4) { // the alleles' bases as codes, for fw_snp_kernel (which then reads one byte per member instead of
                               // allele range -> allele offset -> base)
            u32 codes = 0;
            bool acgt = true;
#pragma unroll
            for (int a = 0; a < 4; ++a)
                if (a < (int)c.A) {
                    bool o;
                    codes |= acgt_code(pool[ao[a]], &o) << (2 * a);
                    acgt = acgt && o;
                }
            if (acgt) c.cls = (unsigned short)(c.cls | REC_CODES | (codes << 8));
        }
        return c;
    }
    c.eligible = present && p >= k && (long long)p <= (long long)clen - k; // var_block.hpp:104
    u64 mask = 0;
    if (c.eligible) { // build_alleles_combs on a chain of one (var_block.hpp:734-786): the alleles some panel haplotype carries
#pragma unroll
        for (int a = 0; a < 4; ++a)
            if (a < (int)c.A && ((raw >> a) & 1)) mask |= 1ULL << cn[a];
        for (u32 a = 4; a < c.A; ++a)
            if ((raw >> a) & 1) mask |= 1ULL << P.canon[c.a0 + a];
    }
    c.mask = mask;
    c.site = cbase + (u64)p;
    return c;
}
// every lane of the wave calls this
__device__ __forceinline__ void list_append(bool take, u32 value, u32 *list, unsigned long long *count)
{
    const u64 m = __ballot(take);
    if (!m) return;
    const int lane = threadIdx.x & 63, leader = __ffsll((unsigned long long)m) - 1;
    unsigned long long base = 0;
    if (lane == leader) base = atomicAdd(count, (unsigned long long)__popcll(m));
    base = __shfl(base, leader, 64);
    if (take) list[base + __popcll(m & ((1ULL << lane) - 1))] = value;
}
constexpr int LONE_TILES = 8; // tiles of TPB / 2 records a workgroup of panel_lone_kernel takes (one list reservation per workgroup:
                              // a returning atomic per wave on ONE counter serialises at ~10 ns each -- 3 ms per 5e6 records)
template <bool SLOW>
__global__ void __launch_bounds__(TPB, SLOW ? 1 : 8) panel_lone_kernel(PanelView P, u64 n_vars, const u32 *__restrict__ blk_var_off, const u32 *__restrict__ var_block,
                                                         const u8 *reference, const u64 *__restrict__ ref2, const u32 *__restrict__ refbad, const u8 *pool, int k,
                                                         int haploid, BFView bf, MapView map, u32 *cov_out, u32 *need_slow, u32 call_no, u32 *gen_list,
                                                         unsigned long long *counters, unsigned short *rec_class)
{
    __shared__ u32 sh_gen[LONE_TILES * TPB / 2];
    __shared__ u32 sh_n, sh_sigs;
    __shared__ unsigned long long sh_base;
    if (SLOW && *need_slow != call_no) return;
    if (threadIdx.x == 0) sh_n = sh_sigs = 0;
    __syncthreads();
    u32 sigs = 0;
    for (int tile = 0; tile < LONE_TILES; ++tile) {
        const u64 t = ((u64)blockIdx.x * LONE_TILES + tile) * TPB + threadIdx.x;
        const u64 v = t >> 1;
        if (v >= n_vars) break;
        const LoneClass c = classify_lone(P, blk_var_off, var_block, v, k, haploid, pool);
        if (!SLOW && !(t & 1)) {
            if (!c.lone) {
                sh_gen[atomicAdd(&sh_n, 1u)] = (u32)v;
                rec_class[v] = c.cls; // (every member of a chain is a record of a block of two or more: listed here)
            } else sigs += (u32)__popcll(c.mask);
        }
        if (c.lone) iso_cover_body<SLOW>(reference, ref2, refbad, c.site, c.a0, c.A, c.eligible, c.mask, (u32)(t & 1), P.allele_off, pool, k, bf, map, cov_out, need_slow, call_no);
    }
    if (SLOW) return;
    for (int d = 32; d; d >>= 1) sigs += __shfl_xor(sigs, d, 64);
    if ((threadIdx.x & 63) == 0 && sigs) atomicAdd(&sh_sigs, sigs);
    __syncthreads();
    if (threadIdx.x == 0) {
        sh_base = sh_n ? atomicAdd(counters, (unsigned long long)sh_n) : 0ULL;
        if (sh_sigs) atomicAdd(counters + 2, (unsigned long long)sh_sigs);
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < sh_n; i += TPB) gen_list[sh_base + i] = sh_gen[i];
}
// index time: lone records are inserted here (REF key of record v takes insertion row row0 + v), the others listed
__global__ void __launch_bounds__(TPB, 8) panel_lone_index_kernel(PanelView P, u64 n_vars, const u32 *__restrict__ blk_var_off, const u32 *__restrict__ var_block,
                                                               const u8 *reference, const u8 *pool, int k, int haploid, BFView bf, MapView map, u32 row0,
                                                               u8 *overflow, u32 *gen_list, unsigned long long *counters, unsigned short *rec_class)
{
    __shared__ u32 sh_gen[LONE_TILES * TPB];
    __shared__ u32 sh_n;
    __shared__ unsigned long long sh_base;
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    for (int tile = 0; tile < LONE_TILES; ++tile) {
        const u64 v = ((u64)blockIdx.x * LONE_TILES + tile) * TPB + threadIdx.x;
        if (v >= n_vars) break;
        const LoneClass c = classify_lone(P, blk_var_off, var_block, v, k, haploid, pool);
        if (!c.lone) {
            sh_gen[atomicAdd(&sh_n, 1u)] = (u32)v;
            rec_class[v] = c.cls;
        } else if (c.eligible && !iso_index_body(reference, c.site, c.a0, c.A, c.mask, P.allele_off, pool, k, bf, map, row0 + (u32)v, row0)) overflow[v] = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) sh_base = sh_n ? atomicAdd(counters, (unsigned long long)sh_n) : 0ULL;
    __syncthreads();
    for (u32 i = threadIdx.x; i < sh_n; i += TPB) gen_list[sh_base + i] = sh_gen[i];
}

// ---- tier 2 ------------------------------------------------------------------------------------------------------------------
struct FlatWork {
    const u32 *gen_list;
    const unsigned long long *gen_count;
    u64 base, round_len; // this round: entries [base, base + round_len) of the list
    CombDesc *combs;
    u32 comb_cap;
    PickItem *items;
    u32 item_cap;
    PickItem *slides; // items of the sliding kind (a lone allele of k bases or more): a wave each, fw_slide_kernel
    u32 slide_cap;
    u32 *retry;       // chains whose picks outgrew their share of a wave's set: taken again, a wave each (comb_cap entries)
    u32 *order;       // the round's chains sorted by their number of members (fw_order_kernel), or NULL: as they were written
    unsigned long long *counters; // this round: [0] descriptors reserved, [1] items reserved, [2] sliding items, [3] chains to retry,
                                  // [FW_ORD_HIST ...) chains of each length, [FW_ORD_CUR ...) places taken in each length's run
    u8 *fb_flag;                  // [n_vars] the record goes to the workgroup kernel
};

// get_combs_on_the_right (step +1, var_block.hpp:436-525) / _left (step -1, :534-624) in private memory; false: a capacity
__device__ bool fw_walk(const BlockBatch &B, int b0, int b1, int i, int step, FwSide *out)
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
    for (int j = i + step; j >= b0 && j < b1 && !halt; j += step) {
        if (j - i > FW_REACH || i - j > FW_REACH) return false; // (the reference walks to the block's end: so does the workgroup kernel)
        if (!B.present[j]) continue;
        if (ov(i, j)) continue;
        const int gain = (int)B.ref_size[j] - (int)B.min_size[j];
        if (out->n == 0) {
            if (nr(i, j, 0)) {
                out->mem[0][0] = (signed char)(j - i);
                out->len[0] = 1;
                out->sum[0] = gain;
                out->n = 1;
            }
            continue;
        }
        bool added = false;
        const int n0 = out->n;
        for (int c = 0; c < n0; ++c) {
            if (!ov(i + out->mem[c][out->len[c] - 1], j)) {
                added = true;
                if (nr(i, j, out->sum[c])) {
                    if (out->len[c] >= FW_MAXM) return false;
                    out->mem[c][out->len[c]++] = (signed char)(j - i);
                    out->sum[c] += gain;
                }
            }
        }
        if (!added) {
            for (int c = 0; c < n0; ++c) {
                int len = out->len[c], ns = out->sum[c];
                while (len > 0 && ov(i + out->mem[c][len - 1], j)) {
                    const int m = i + out->mem[c][len - 1];
                    ns -= (int)B.ref_size[m] - (int)B.min_size[m];
                    --len;
                }
                if (nr(i, j, ns)) {
                    added = true;
                    if (out->n >= FW_MAXC || len + 1 > FW_MAXM) return false;
                    const int d = out->n++;
                    for (int q = 0; q < len; ++q) out->mem[d][q] = out->mem[c][q];
                    out->mem[d][len] = (signed char)(j - i);
                    out->len[d] = len + 1;
                    out->sum[d] = ns + gain;
                }
            }
            if (!added) halt = true;
        }
    }
    return true;
}

// MODE 0 call time, 1 index time counting pass, 2 index time insert pass (skips what pass 1 flagged)
template <int MODE>
__global__ void __launch_bounds__(TPB) fw_walk_kernel(BlockBatch B, FlatWork W, u32 *cov_out, u8 *overflow)
{
    const u64 tid = (u64)blockIdx.x * TPB + threadIdx.x;
    const u64 idx = W.base + tid;
    const bool in = tid < W.round_len && idx < *W.gen_count;
    FwSide L, R;
    L.n = R.n = 0;
    u32 g = 0, cid = 0, n_combs = 0;
    if (in) {
        g = W.gen_list[idx];
        const u32 a0 = B.var_allele_off[g], A = B.var_allele_off[g + 1] - a0;
        const u32 blk = B.var_block[g];
        const int b0 = (int)B.blk_var_off[blk], b1 = (int)B.blk_var_off[blk + 1];
        cid = B.contig_id[b0];
        const i32 ref_len = (i32)B.contig_len[cid];
        if (MODE == 0)
            for (u32 a = 0; a < A; ++a) cov_out[a0 + a] = 0;
        bool skip = MODE == 2 && (overflow[g] || W.fb_flag[g]);
        if (!skip && (A > 127 || B.k > MG_MAX_PACKED_K)) { // what the workgroup kernel flags at once
            if (MODE != 2) overflow[g] = 1;
            skip = true;
        }
        if (!skip && B.n_samples > FW_MAX_SAMPLES) {
            W.fb_flag[g] = 1;
            skip = true;
        }
        const bool eligible = B.present[g] && B.pos[g] >= B.k && B.pos[g] <= ref_len - B.k; // var_block.hpp:104
        if (!skip && eligible) {
            if (fw_walk(B, b0, b1, (int)g, -1, &L) && fw_walk(B, b0, b1, (int)g, +1, &R)) n_combs = (u32)((L.n ? L.n : 1) * (R.n ? R.n : 1));
            else W.fb_flag[g] = 1;
        }
    }
    // room for the workgroup's descriptors: one atomic (lanes scan inside the wave, waves through LDS)
    __shared__ u32 sh_tot[TPB / 64];
    __shared__ unsigned long long sh_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32 incl = n_combs;
    for (int d = 1; d < 64; d <<= 1) {
        const u32 up = __shfl_up(incl, d, 64);
        if (lane >= d) incl += up;
    }
    if (lane == 63) sh_tot[wave] = incl;
    __syncthreads();
    u32 before = 0, total = 0;
    for (int w = 0; w < TPB / 64; ++w) {
        if (w < wave) before += sh_tot[w];
        total += sh_tot[w];
    }
    if (!total) return;
    if (threadIdx.x == 0) sh_base = atomicAdd(&W.counters[0], (unsigned long long)total);
    __syncthreads();
    const unsigned long long base = sh_base + before;
    if (sh_base + total > W.comb_cap) { // (an average of FW_COMBS_PER_REC per record of the round was not enough)
        if (n_combs) W.fb_flag[g] = 1;
        CombDesc none{};                   // what of the reservation lies inside the buffer must not be read as descriptors
        for (u64 q = sh_base + threadIdx.x; q < W.comb_cap && q < sh_base + total; q += TPB) W.combs[q] = none;
        return;
    }
    if (!n_combs) return;
    CombDesc *out = W.combs + base + (incl - n_combs);
    const int nl = L.n ? L.n : 1, nrr = R.n ? R.n : 1;
    for (int c = 0; c < nl * nrr; ++c) { // combine_combs (var_block.hpp:630-677): every left chain (reversed) + the variant + every right chain
        const int cl = c / nrr, cr = c % nrr;
        const int len_l = L.n ? L.len[cl] : 0, len_r = R.n ? R.len[cr] : 0;
        CombDesc d;
        d.g = g;
        d.cid = cid;
        d.m = (u8)(len_l + 1 + len_r);
        d.jm = (u8)len_l;
        for (int j = 0; j < 22; ++j) d.rel[j] = 0;
        for (int j = 0; j < len_l; ++j) d.rel[j] = L.mem[cl][len_l - 1 - j];
        for (int j = 0; j < len_r; ++j) d.rel[len_l + 1 + j] = R.mem[cr][j];
        // rel[21] (never a member: a chain has at most 21) = 1: every member replaces one base by one base whatever the allele,
        // and -- along a chain of two or more, diploid -- no sample is unphased at any member (tier 1 left that in rec_class):
        // fw_snp_kernel takes such a chain whole unless its window leaves the sequence or holds a base outside ACGT
        bool snps = true;
        const unsigned short need = (unsigned short)(REC_SNP | REC_CODES | (d.m > 1 ? REC_PHASED : 0));
        for (int j = 0; j < d.m; ++j) {
            const u32 v = g + d.rel[j];
            snps = snps && (B.rec_class[v] & need) == need;
        }
        d.rel[21] = snps ? 1 : 0;
        out[c] = d;
    }
}

// The round's chains still to be taken -- not the reservations that did not fit, not the chains fw_snp_kernel took -- in order
// of their LENGTH (number of members): a counting sort by two small kernels.  The kernels that follow deal chains to lane
// groups in the order they come; every loop over a chain's members then runs to the longest chain of the wave, and a wave
// that mixes chains of one and of six members spends most of its lanes waiting (C5: tier 2 1.74 -> 1.58 ms with the order).
// On a whole-genome SNP panel the list is also what keeps those kernels from walking 1.2e7 descriptors to find the few
// fw_snp_kernel left.  A wave counts its chains of one length with one LDS atomic (a panel whose chains all have two members
// would otherwise send 256 atomics of a tile to one address).  Which chain of a length comes first is left to the hardware:
// nothing downstream depends on it.
__device__ __forceinline__ u64 fw_chains_todo(const FlatWork &W)
{
    if (!W.order) return min((unsigned long long)W.comb_cap, W.counters[0]);
    unsigned long long n = 0;
    for (int b = 1; b < FW_ORD_BINS; ++b) n += W.counters[FW_ORD_HIST + b];
    return n;
}
template <int PASS>
__global__ void __launch_bounds__(TPB) fw_order_kernel(FlatWork W)
{
    // A workgroup takes a CONTIGUOUS run of descriptors and touches the round's counters once per length: with a tile's worth
    // per workgroup, 4,096 workgroups sent their atomics to the same half-dozen words (44 us per pass for 1e6 descriptors, C5).
    __shared__ u32 sh_hist[FW_ORD_BINS], sh_tot[FW_ORD_BINS], sh_next[FW_ORD_BINS];
    const u64 n = min((unsigned long long)W.comb_cap, W.counters[0]);
    const int lane = threadIdx.x & 63;
    const u64 tiles = (n + TPB - 1) / TPB, per = (tiles + gridDim.x - 1) / gridDim.x;
    const u64 t_begin = min(tiles, (u64)blockIdx.x * per) * TPB, t_end = min(tiles, ((u64)blockIdx.x + 1) * per) * TPB;
    if (threadIdx.x < FW_ORD_BINS) sh_tot[threadIdx.x] = 0;
    auto length_of = [&](u64 i) { // 0: nothing to take (a reservation that did not fit, or a chain fw_snp_kernel took)
        u32 m = 0;
        if (i < n) {
            const u32 mark = *(const u32 *)&W.combs[i].rel[18] >> 24;
            m = mark == 2 ? 0u : (u32)W.combs[i].m;
            m = m < FW_ORD_BINS ? m : FW_ORD_BINS - 1;
        }
        return m;
    };
    auto rank_in = [&](u32 m, u32 *hist) { // the wave's chains of one length: one LDS atomic
        u32 rank = 0;
        for (u64 todo = __ballot(m != 0); todo;) {
            const int leader = __ffsll((unsigned long long)todo) - 1;
            const u32 v = (u32)__shfl((int)m, leader, 64);
            const u64 same = __ballot(m == v);
            u32 base = 0;
            if (lane == leader) base = atomicAdd(&hist[v], (u32)__popcll(same));
            base = (u32)__shfl((int)base, leader, 64);
            if (m == v) rank = base + (u32)__popcll(same & ((1ULL << lane) - 1));
            todo &= ~same;
        }
        return rank;
    };
    __syncthreads();
    for (u64 t0 = t_begin; t0 < t_end; t0 += TPB) rank_in(length_of(t0 + threadIdx.x), sh_tot); // the run's chains of each length
    __syncthreads();
    if (PASS == 0) { // count
        if (threadIdx.x < FW_ORD_BINS && sh_tot[threadIdx.x]) atomicAdd(&W.counters[FW_ORD_HIST + threadIdx.x], (unsigned long long)sh_tot[threadIdx.x]);
        return;
    }
    // place: a length's part of the list starts behind the shorter lengths', the workgroup takes its share of it at once
    if (threadIdx.x >= 1 && threadIdx.x < FW_ORD_BINS) {
        unsigned long long start = 0;
        for (int b = 1; b < (int)threadIdx.x; ++b) start += W.counters[FW_ORD_HIST + b];
        sh_next[threadIdx.x] = sh_tot[threadIdx.x] ? (u32)(start + atomicAdd(&W.counters[FW_ORD_CUR + threadIdx.x], (unsigned long long)sh_tot[threadIdx.x])) : 0u;
    }
    for (u64 t0 = t_begin; t0 < t_end; t0 += TPB) { // (block-uniform bounds: barriers inside)
        if (threadIdx.x < FW_ORD_BINS) sh_hist[threadIdx.x] = 0;
        __syncthreads();
        const u64 i = t0 + threadIdx.x;
        const u32 m = length_of(i);
        const u32 rank = rank_in(m, sh_hist);
        __syncthreads();
        if (m) W.order[sh_next[m] + rank] = (u32)i;
        __syncthreads();
        if (threadIdx.x < FW_ORD_BINS) sh_next[threadIdx.x] += sh_hist[threadIdx.x];
    }
}

// build_alleles_combs + combine_haplotypes (var_block.hpp:709-786): the distinct picks of all panel samples along a chain.
// A wave takes 64 / G chains at a time, G lanes each (G = the panel's sample count rounded up to a power of two, at most
// 64: a wave per chain leaves 62 lanes idle on a two-sample panel); lanes stride over the samples.  The picks of all the
// wave's chains share one LDS set, told apart by the chain's number inside the wave (FW_GRP_SHIFT).
constexpr int FW_WAVES = TPB / 64;
constexpr int FW_CODE_BITS = 55;                     // a pick as a code: bits per member = ceil(log2(alleles)), members left to right
constexpr unsigned long long FW_SLIDE_IN = 1ULL << 56; // set key: the chain is the variant alone and the allele has k bases or more
constexpr int FW_GRP_SHIFT = 57;                     // set key: the chain's number inside the wave (6 bits)
constexpr unsigned long long FW_SLIDE = 1ULL << 62;  // the same tag on an item
constexpr u32 FW_CHUNK = 512; // items a wave reserves at a time (one returning atomic per chain on ONE counter: 7 ms per 7e5 chains)
// Whether a chain's picks fit depends on that chain alone -- its share of the set is FW_SET * 3 / 4 / (chains per wave) -- never on
// which chains happen to share its wave: `index` runs the tiers twice (count, insert) and both passes must send the same
// records the same way.  A chain that outgrows its share is listed in `retry` and taken again alone (RETRY: G = 64, the whole
// set); one that outgrows that goes to the workgroup kernel.
template <bool RETRY>
__global__ void __launch_bounds__(TPB) fw_picks_kernel(BlockBatch B, FlatWork W, int G_in, u32 item_cap_eff)
{
    __shared__ unsigned long long sh_set[FW_WAVES][FW_SET]; // key + 1, 0 = free
    __shared__ unsigned short sh_list[FW_WAVES][FW_SET];    // slots taken, in the order they were taken
    __shared__ u32 sh_n[FW_WAVES];
    __shared__ u32 sh_cnt[FW_WAVES][32];                    // distinct picks per chain of the wave
    const int G = RETRY ? 64 : G_in;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane / G, sub = lane % G, n_grp = 64 / G;
    unsigned long long *set = sh_set[wave];
    unsigned short *list = sh_list[wave];
    for (int i = lane; i < FW_SET; i += 64) set[i] = 0;
    if (lane == 0) sh_n[wave] = 0;
    if (lane < 32) sh_cnt[wave][lane] = 0;
    wave_sync();
    const u32 share = (u32)(FW_SET * 3 / 4 / n_grp); // distinct picks a chain may have here
    const u64 n_combs = RETRY ? min((unsigned long long)W.comb_cap, W.counters[3]) : fw_chains_todo(W);
    const u64 n_waves = (u64)gridDim.x * FW_WAVES;
    unsigned long long chunk_at = 0; // next free item of the wave's chunk
    u32 chunk_left = 0;
    volatile u32 *my_cnt = &sh_cnt[wave][grp];
    auto insert = [&](unsigned long long key) {
        u32 at = (u32)((key * 0x9E3779B97F4A7C15ULL) >> 40) & (FW_SET - 1);
        for (int tries = 0; tries < FW_SET; ++tries) {
            const unsigned long long seen = atomicCAS(&set[at], 0ULL, key + 1);
            if (seen == 0ULL) {
                atomicAdd(&sh_cnt[wave][grp], 1u);
                const u32 q = atomicAdd(&sh_n[wave], 1u);
                if (q < FW_SET) list[q] = (unsigned short)at;
                return;
            }
            if (seen == key + 1) return;
            at = (at + 1) & (FW_SET - 1);
        }
    };
    for (u64 c0 = ((u64)blockIdx.x * FW_WAVES + wave) * n_grp; c0 < n_combs; c0 += n_waves * n_grp) {
        const u64 cpos = c0 + grp;                            // position in the list of chains this launch walks
        const u64 ci = cpos < n_combs ? (RETRY ? (u64)W.retry[cpos] : W.order ? (u64)W.order[cpos] : cpos) : 0; // the chain's descriptor
        CombDesc d{};
        if (cpos < n_combs) d = W.combs[ci];
        const int m = d.m;
        const u32 g = d.g;
        const bool valid = cpos < n_combs && m > 0 && d.rel[21] != 2 && !W.fb_flag[g]; // (m == 0: a reservation that did not fit; rel[21] == 2: fw_snp_kernel took the chain)
        bool fail = false;  // the record goes to the workgroup kernel
        if (valid)
            for (u32 s = sub; s < B.n_samples && !fail; s += G) {
                if (*my_cnt > share) break; // the chain outgrew its share of the set (all its lanes see that sooner or later)
                bool phased = true;
                unsigned long long c1 = 0, c2 = 0, bounds = 1; // bounds: bit set at every member's first code bit, and behind the last
                int sh = 0;
                for (int j = 0; j < m; ++j) {
                    const u32 v = g + d.rel[j];
                    const int bits = fw_bits(B.var_allele_off[v + 1] - B.var_allele_off[v]);
                    if (sh + bits > FW_CODE_BITS) {
                        fail = true;
                        break;
                    }
                    const u32 gt = gt_at(B, v, s);
                    phased = phased && ((gt >> 14) & 1);
                    c1 |= (unsigned long long)(gt & 127) << sh;
                    c2 |= (unsigned long long)((gt >> 7) & 127) << sh;
                    sh += bits;
                    bounds |= 1ULL << sh;
                }
                if (fail) break;
                const unsigned long long tag = (unsigned long long)grp << FW_GRP_SHIFT;
                if (m == 1) { // an allele of k bases or more on its own is a SLIDING signature (var_block.hpp:130-144): its own kind of item
                    const u32 a0 = B.var_allele_off[g];
                    const u32 l1 = B.allele_off[a0 + (u32)c1 + 1] - B.allele_off[a0 + (u32)c1], l2 = B.allele_off[a0 + (u32)c2 + 1] - B.allele_off[a0 + (u32)c2];
                    if ((int)l1 >= B.k) c1 |= FW_SLIDE_IN;
                    if ((int)l2 >= B.k) c2 |= FW_SLIDE_IN;
                }
                if (B.haploid) insert(c1 | tag);
                else if (phased || m == 1) { // (the mixes of a chain of one are its two alleles)
                    insert(c1 | tag);
                    insert(c2 | tag);
                } else if (m > FW_MAXU) fail = true;
                else { // every mix of the two haplotypes (combine_haplotypes): Gray-code walk, one member's field flipped per step
                    // (only the members at which the two haplotypes differ: 2^h codes for h of them, not 2^m steps of which most change nothing)
                    const unsigned long long diff = c1 ^ c2;
                    u32 het = 0;
                    {
                        unsigned long long t = bounds;
                        for (int j = 0; j < m; ++j) {
                            const int lo = __ffsll((unsigned long long)t) - 1;
                            t &= t - 1;
                            const int hi = __ffsll((unsigned long long)t) - 1;
                            if (diff & (((1ULL << (hi - lo)) - 1) << lo)) het |= 1u << j;
                        }
                    }
                    const int n_het = __popc(het);
                    unsigned long long code = c1;
                    insert(code | tag);
                    for (u32 i = 1; i < (1u << n_het); ++i) {
                        u32 hm = het; // the member this step flips: the (number of trailing zeros of i)-th of the differing ones
                        for (int q = __ffs((int)i) - 1; q > 0; --q) hm &= hm - 1;
                        const int j = __ffs((int)hm) - 1;
                        unsigned long long t = bounds;
                        for (int q = 0; q < j; ++q) t &= t - 1;
                        const int lo = __ffsll((unsigned long long)t) - 1;
                        t &= t - 1;
                        const int hi = __ffsll((unsigned long long)t) - 1;
                        code ^= diff & (((1ULL << (hi - lo)) - 1) << lo);
                        insert(code | tag);
                        if (*my_cnt > share) break;
                    }
                }
            }
        wave_sync();
        const u32 n = sh_n[wave];
        const bool all_fail = n > FW_SET; // (cannot happen: every chain stops at its share; kept as a guard)
        const bool grew = valid && sh_cnt[wave][grp] > share; // the chain outgrew its share: again alone, or -- already alone -- the workgroup kernel
        if (RETRY) fail = fail || grew;
        else if (grew && sub == 0) {
            const unsigned long long at = atomicAdd(&W.counters[3], 1ULL);
            if (at < W.comb_cap) W.retry[at] = (u32)ci;
            else fail = true;
        }
        // chains (by their number in the wave) nothing is written for
        u64 failed = 0;
        {
            const u64 fl = __ballot(fail || grew);
            for (int q = 0; q < n_grp; ++q) {
                const u64 qm = (G == 64 ? ~0ULL : ((1ULL << G) - 1)) << (q * G);
                if (all_fail || (fl & qm)) failed |= 1ULL << q;
            }
        }
        const u64 to_wg = __ballot(fail); // lanes whose chain's record goes to the workgroup kernel
        const u32 n_used = n < FW_SET ? n : FW_SET;
        bool buffer_full = false;
        if (!all_fail)
            for (u32 i0 = 0; i0 < n_used; i0 += 64) { // the set's entries, 64 at a time: ordinary items into the wave's chunk, sliding ones into their own list
                const u32 i = i0 + lane;
                const unsigned long long key = i < n_used ? set[list[i]] - 1 : 0ULL;
                const int kg = (int)(key >> FW_GRP_SHIFT) & 63;
                const bool live = i < n_used && !((failed >> kg) & 1);
                const bool slide = live && (key & FW_SLIDE_IN), norm = live && !slide;
                const unsigned long long code = key & ((1ULL << FW_CODE_BITS) - 1);
                const u64 nm = __ballot(norm);
                const u32 cnt = (u32)__popcll(nm);
                if (cnt > chunk_left) { // a fresh chunk (the rest of the old one is nulled so that it is skipped)
                    for (u32 q = lane; q < chunk_left; q += 64) W.items[chunk_at + q] = PickItem{0xFFFFFFFFu, 0u, 0ULL};
                    unsigned long long base = 0;
                    if (lane == 0) base = atomicAdd(&W.counters[1], (unsigned long long)FW_CHUNK);
                    base = __shfl(base, 0, 64);
                    if (base + FW_CHUNK > item_cap_eff) { // the round's buffer is full: everything still to be written goes to the workgroup kernel
                        for (u64 q = base + lane; q < W.item_cap && q < base + FW_CHUNK; q += 64) W.items[q] = PickItem{0xFFFFFFFFu, 0u, 0ULL};
                        chunk_left = 0;
                        buffer_full = true; // (records with items already written are redone whole: harmless)
                        break;
                    }
                    chunk_at = base;
                    chunk_left = FW_CHUNK;
                }
                const u32 kci = (u32)__shfl((int)(u32)ci, RETRY ? 0 : kg * G, 64); // the descriptor of the key's chain (alone in the wave, or the kg-th of the wave's run of chains)
                if (norm) W.items[chunk_at + __popcll(nm & ((1ULL << lane) - 1))] = PickItem{kci, 0u, code};
                chunk_at += cnt;
                chunk_left -= cnt;
                if (slide) {
                    const unsigned long long at = atomicAdd(&W.counters[2], 1ULL);
                    if (at < W.slide_cap) W.slides[at] = PickItem{kci, 0u, code | FW_SLIDE};
                    else W.fb_flag[W.combs[kci].g] = 1;
                }
            }
        if (valid && sub == 0 && (all_fail || buffer_full || ((to_wg >> (grp * G)) & (G == 64 ? ~0ULL : ((1ULL << G) - 1))))) W.fb_flag[g] = 1;
        for (u32 i = lane; i < n_used; i += 64) set[list[i]] = 0;
        if (lane < 32) sh_cnt[wave][lane] = 0;
        if (n > FW_SET) // (the list lost entries: clear the whole set)
            for (int i = lane; i < FW_SET; i += 64) set[i] = 0;
        wave_sync();
        if (lane == 0) sh_n[wave] = 0;
        wave_sync();
    }
    for (u32 i = lane; i < chunk_left; i += 64) W.items[chunk_at + i] = PickItem{0xFFFFFFFFu, 0u, 0ULL};
}

// Chains of SNPs on a panel of few samples -- every chain of a whole-genome SNP panel -- in ONE kernel, lanes = the panel's
// haplotypes along the chain (16 at most: 8 diploid samples), 64 / lanes chains per wave: a haplotype's k-mer is the reference
// window around the central record with the members' bases put in; equal k-mers of one chain are told apart by comparing keys
// across the chain's lanes, and the first of each goes on to canonical form, XXH3 and its lookup (or insert).  No set in LDS,
// no items written and read back, no second walk over the members: what fw_picks_kernel + fw_eval_kernel do for such a chain
// in 4.5 ms per 1.2e7 chains (C4) is one pass here.  A chain it cannot take whole -- a sample unphased along a chain of two or
// more members (2^m mixes), a picked allele longer than one base, a window that leaves the sequence or holds a base outside
// ACGT, more haplotypes than 16 -- is left exactly as it was for the two kernels; one it takes is marked (rel[21] = 2) and
// skipped by them.  The decision depends on the chain and the panel alone, so `index`'s two passes agree.
constexpr u32 FW_SNP_MAX_HAPS = 16;
template <int MODE>
__global__ void __launch_bounds__(TPB, 8) fw_snp_kernel(BlockBatch B, FlatWork W, int G, BFView bf, MapView map, u32 *cov_out, unsigned long long *cursor, u32 row0,
                                                     unsigned long long *n_evaluated)
{
    const int lane = threadIdx.x & 63;
    const int grp = lane / G, sub = lane % G, n_grp = 64 / G;
    const u64 gmask = (G == 64 ? ~0ULL : ((1ULL << G) - 1)) << (grp * G);
    const u32 n_haps = B.haploid ? B.n_samples : 2 * B.n_samples;
    const u64 n_combs = min((unsigned long long)W.comb_cap, W.counters[0]);
    const u64 n_waves = (u64)gridDim.x * FW_WAVES;
    const int k = B.k;
    u32 evaluated = 0, ref_rows = 0;
    // The wave reads 64 descriptors' marks at a time and deals the chains of SNPs among them to its lane groups: on a panel of
    // indels and MNPs one chain in twenty is such a chain, and a wave that gave a lane group to every descriptor spent
    // 0.19 ms per 1e6 chains finding that out (C5)
    for (u64 c0 = ((u64)blockIdx.x * FW_WAVES + (threadIdx.x >> 6)) * 64; c0 < n_combs; c0 += n_waves * 64) // (whole waves: ballots inside)
      for (u64 pending = __ballot(c0 + lane < n_combs && (*(const u32 *)&W.combs[c0 + lane < n_combs ? c0 + lane : 0].rel[18] >> 24) == 1u); pending;) {
        u64 mine_bit = pending; // the grp-th chain still to be dealt
        for (int q = 0; q < grp; ++q) mine_bit &= mine_bit - 1;
        const bool has = mine_bit != 0;
        const u64 ci = c0 + (u64)(has ? __ffsll((unsigned long long)mine_bit) - 1 : 0);
        for (int q = 0; q < n_grp && pending; ++q) pending &= pending - 1;
        CombDesc d{};
        if (has) d = W.combs[ci];
        const int m = d.m, jm = d.jm;
        const u32 g = d.g;
        const bool chain_ok = has && m > 0 && d.rel[21] == 1 && !W.fb_flag[g] && k >= 17 && k <= MG_MAX_PACKED_K;
        const bool mine = chain_ok && (u32)sub < n_haps; // this lane carries a haplotype of the chain
        const u32 smp = B.haploid ? (u32)sub : (u32)sub >> 1, second = B.haploid ? 0u : (u32)sub & 1u;
        bool good = true; // (of lanes that carry a haplotype) the haplotype fits the fixed geometry
        U128 Lf{0, 0};
        u32 mid_allele = 0;
        unsigned long long code = 0; // the haplotype's pick as fw_picks_kernel codes it: the members' alleles, ceil(log2(alleles)) bits each
        if (mine) {
            const i32 ref_len = (i32)B.contig_len[d.cid];
            const int w0 = B.pos[g] - k / 2;
            good = w0 >= 0 && w0 + k <= ref_len;
            if (good) {
                const u64 at = B.contig_base[d.cid] + (u64)w0;
                Lf.lo = ref_codes(B.ref2, at, k < 32 ? k : 32);
                if (k > 32) Lf.hi = ref_codes(B.ref2, at + 32, k - 32);
                good = !ref_bad(B.refbad, at, k < 32 ? k : 32) && (k <= 32 || !ref_bad(B.refbad, at + 32, k - 32));
                for (int j = 0; j < m && good; ++j) { // (a marked chain: every member one base for one base, at most four alleles, all
                                                      // ACGT, phased wherever that matters -- fw_walk_kernel saw to it, from tier 1's class words)
                    const u32 v = g + d.rel[j];
                    const u32 gt = gt_at(B, v, smp);
                    const u32 a = second ? (gt >> 7) & 127 : gt & 127;
                    const u32 codes = (u32)B.rec_class[v] >> 8;
                    if (a > 3) good = false; // (a genotype beyond the record's alleles: left to the picks kernel's reading of it)
                    code |= (unsigned long long)(a & 3) << (2 * j);
                    if (j == jm) mid_allele = a;
                    const int x = B.pos[v] - w0;
                    if (good && x >= 0 && x < k) { // the member's base at its own place in the window (a member outside it changes nothing)
                        const u64 c2 = (codes >> (2 * a)) & 3;
                        if (x < 32) Lf.lo = (Lf.lo & ~(3ULL << (2 * x))) | c2 << (2 * x);
                        else Lf.hi = (Lf.hi & ~(3ULL << (2 * (x - 32)))) | c2 << (2 * (x - 32));
                    }
                }
            }
        }
        // the chain is taken here iff every one of its haplotypes fits (the group's lanes agree through the ballot)
        const u64 bad = __ballot(mine && !good), have = __ballot(mine);
        const bool take = chain_ok && n_haps <= FW_SNP_MAX_HAPS && !(bad & gmask) && (have & gmask);
        // first lane of the group with each distinct PICK (what build_alleles_combs' set holds, var_block.hpp:734-786: two picks that
        // differ only outside the window are two signatures of one k-mer, evaluated twice as the reference does)
        bool first = take && mine;
        for (int q = 1; q < G; ++q) {
            const int src = grp * G + ((sub + G - q) % G); // every other lane of the group, one by one
            const unsigned long long ocode = (unsigned long long)__shfl((long long)code, src, 64);
            const bool omine = (have >> src) & 1;
            if (first && omine && (src % G) < sub && ocode == code) first = false;
        }
        if (take && sub == 0) W.combs[ci].rel[21] = 2; // fw_picks_kernel leaves the chain alone
        if (!first) continue;
        ++evaluated;
        const u32 a0 = B.var_allele_off[g];
        const u32 mid_canon = B.canon[a0 + mid_allele];
        const U128 mk = mask128(2 * k);
        const U128 mform = shr128(U128{pairrev64(Lf.hi), pairrev64(Lf.lo)}, 2 * (64 - k));
        const U128 rc{~mform.lo & mk.lo, ~mform.hi & mk.hi};
        const U128 key = lt128(Lf, rc) ? Lf : rc;
        const u64 h = k == 35 ? xxh3_packed_fixed<35>(key.lo, key.hi) : xxh3_packed(key, k);
        const u64 idx = mod_size(h, bf.mod);
        const bool is_ref = mid_canon == 0;
        if (MODE == 0) {
            i32 w;
            if (is_ref) w = k == (int)map.klen ? map_value(map, key, h, idx) : 0;
            else w = (i32)bucket_count(map, bf.counts, idx);
            if (w > 0) atomicMax(&cov_out[a0 + mid_canon], (u32)w);
        } else if (is_ref) {
            if (MODE == 1) ++ref_rows;
            else map_insert_key(map, bf, key, h, row0 + wave_take(cursor), row0);
        } else if (MODE == 2) {
            atomicOr((unsigned long long *)&bf.words[idx >> 6], 1ULL << (idx & 63)); // BF::add_key, bloom_filter.hpp:81-85
            gate_set(bf, idx);
        }
    }
    if (MODE == 1) {
        for (int dd = 32; dd; dd >>= 1) ref_rows += __shfl_xor(ref_rows, dd, 64);
        if (lane == 0 && ref_rows) atomicAdd(cursor, (unsigned long long)ref_rows);
    }
    if (n_evaluated) {
        for (int dd = 32; dd; dd >>= 1) evaluated += __shfl_xor(evaluated, dd, 64);
        if (lane == 0 && evaluated) atomicAdd(n_evaluated, (unsigned long long)evaluated);
    }
}

// one signature k-mer: assembly in 2-bit form, canonical, XXH3, then MODE 0 lookup + max into the allele's coverage,
// MODE 1 count the REF k-mers, MODE 2 insert
template <int MODE>
__global__ void __launch_bounds__(TPB) fw_eval_kernel(BlockBatch B, FlatWork W, BFView bf, MapView map, u32 *cov_out, u8 *overflow, unsigned long long *cursor, u32 row0,
                                                      unsigned long long *n_evaluated)
{
    const u64 n_items = min((unsigned long long)W.item_cap, W.counters[1]);
    const u64 n_threads = (u64)gridDim.x * TPB;
    const int k = B.k;
    u32 evaluated = 0, ref_rows = 0;
    for (u64 it = (u64)blockIdx.x * TPB + threadIdx.x; it < n_items; it += n_threads) {
        const PickItem item = W.items[it];
        if (item.comb == 0xFFFFFFFFu) continue; // a reservation that did not fit
        const CombDesc d = W.combs[item.comb];
        const u32 g = d.g;
        if (W.fb_flag[g]) continue;
        const int m = d.m, jm = d.jm;
        const i32 ref_len = (i32)B.contig_len[d.cid];
        // A chain of SNPs (every member one base for one base -- all of a whole-genome SNP panel's clusters) has the same geometry
        // whatever the pick: the window is the reference around the central record, k/2 bases to its left, with the members'
        // bases put in.  It is built beside the lengths below and used when every member turned out to be such a record.
        const u64 cbase = B.contig_base[d.cid];
        const int w0 = B.pos[g] - k / 2;
        const bool snp_try = d.rel[21] == 1 && B.snp_chains && k >= 17 && k <= MG_MAX_PACKED_K && w0 >= 0 && w0 + k <= ref_len;
        U128 Ls{0, 0};
        bool snp_ok = snp_try;
        if (snp_try) { // (requested before the members' loads are: they wait for nothing)
            const u64 at = cbase + (u64)w0;
            Ls.lo = ref_codes(B.ref2, at, k < 32 ? k : 32);
            if (k > 32) Ls.hi = ref_codes(B.ref2, at + 32, k - 32);
            snp_ok = !ref_bad(B.refbad, at, k < 32 ? k : 32) && (k <= 32 || !ref_bad(B.refbad, at + 32, k - 32));
        }
        bool all_snp = true;
        // lengths: virtual string V = A_0 R_0 A_1 ... A_{m-1}
        int len_v = 0, mid_pos = 0, mid_len = 0, sh = 0;
        u32 mid_allele = 0;
        const int first_pos = B.pos[g + d.rel[0]];
        int last_end = 0;
        for (int j = 0; j < m; ++j) {
            const u32 v = g + d.rel[j];
            const u32 s0 = B.var_allele_off[v];
            const int bits = fw_bits(B.var_allele_off[v + 1] - s0);
            const u32 a = (u32)(item.code >> sh) & ((1u << bits) - 1);
            sh += bits;
            const u32 ao = B.allele_off[s0 + a];
            const int al = (int)(B.allele_off[s0 + a + 1] - ao);
            const int pv = B.pos[v], rs = (int)B.ref_size[v];
            if (j == jm) {
                mid_pos = len_v;
                mid_len = al;
                mid_allele = a;
            }
            all_snp = all_snp && al == 1 && rs == 1;
            if (snp_try && al == 1) { // the member's base at its own place in the window (a member outside it changes nothing)
                const int x = pv - w0;
                if (x >= 0 && x < k) {
                    bool o;
                    const u64 code = acgt_code(B.pool[ao], &o);
                    snp_ok = snp_ok && o;
                    if (x < 32) Ls.lo = (Ls.lo & ~(3ULL << (2 * x))) | code << (2 * x);
                    else Ls.hi = (Ls.hi & ~(3ULL << (2 * (x - 32)))) | code << (2 * (x - 32));
                }
            }
            len_v += al;
            last_end = pv + rs;
            if (j + 1 < m) len_v += B.pos[g + d.rel[j + 1]] - last_end;
        }
        const u32 a0 = B.var_allele_off[g];
        const u32 mid_canon = B.canon[a0 + mid_allele];
        const int first_part = mid_pos + mid_len / 2;
        const int mp = k / 2 - first_part;                 // missing_prefix (negative: cut)
        const int ms = (k + 1) / 2 - (len_v - first_part); // missing_suffix
        if (first_pos - (mp > 0 ? mp : 0) < 0 || last_end + (ms > 0 ? ms : 0) > ref_len) {
            if (MODE != 2) overflow[g] = 1; // the reference clips or throws here: the host path's
            continue;
        }
        // W[x] = Vext[x - mp] for x in [0, k), Vext = V with the reference continuing on both sides.  Reference stretches come
        // out of the packed reference 32 bases at a time (two loads and a shift each), alleles byte by byte
        U128 Lf{0, 0};
        bool ok = k >= 17 && k <= MG_MAX_PACKED_K;
        const bool fast_snp = snp_try && all_snp;
        auto put = [&](int x, u32 byte) {
            bool o;
            const u64 code = acgt_code(byte, &o);
            ok = ok && o;
            if (x < 32) Lf.lo |= code << (2 * x);
            else Lf.hi |= code << (2 * (x - 32));
        };
        auto put_ref = [&](int x, int xe, long long from) { // W[x .. xe) = reference[from ..) (positions inside the contig)
            while (x < xe) {
                const int n = xe - x < 32 ? xe - x : 32;
                const u64 at = cbase + (u64)from;
                ok = ok && !ref_bad(B.refbad, at, n);
                const U128 sp = shl128(U128{ref_codes(B.ref2, at, n), 0}, 2 * x);
                Lf.lo |= sp.lo;
                Lf.hi |= sp.hi;
                x += n;
                from += n;
            }
        };
        if (fast_snp) {
            Lf = Ls;
            ok = snp_ok;
        } else {
        put_ref(0, mp < k ? (mp > 0 ? mp : 0) : k, (long long)first_pos - mp);
        int vs = 0;
        sh = 0;
        for (int j = 0; j < m; ++j) {
            const u32 v = g + d.rel[j];
            const u32 s0 = B.var_allele_off[v];
            const int bits = fw_bits(B.var_allele_off[v + 1] - s0);
            const u32 a = (u32)(item.code >> sh) & ((1u << bits) - 1);
            sh += bits;
            const u32 ao = B.allele_off[s0 + a];
            const u8 *ap = B.pool + ao;
            const int al = (int)(B.allele_off[s0 + a + 1] - ao);
            if (B.pool2) { // the allele's part of the window out of the packed pool: two loads and a shift per 32 bases, like the reference's
                           // (chains of SNPs, for whose single bytes this costs more than it saves, take the fixed-geometry form above)
                int x = max(0, vs + mp);
                const int xe = min(k, vs + al + mp);
                while (x < xe) {
                    const int n = xe - x < 32 ? xe - x : 32;
                    const u64 at = (u64)ao + (u64)(x - mp - vs);
                    ok = ok && !ref_bad(B.poolbad, at, n);
                    const U128 sp = shl128(U128{ref_codes(B.pool2, at, n), 0}, 2 * x);
                    Lf.lo |= sp.lo;
                    Lf.hi |= sp.hi;
                    x += n;
                }
            } else
                for (int x = max(0, vs + mp), xe = min(k, vs + al + mp); x < xe; ++x) put(x, ap[x - mp - vs]);
            vs += al;
            if (j + 1 < m) {
                const int gs = B.pos[v] + (int)B.ref_size[v];
                const int gl = B.pos[g + d.rel[j + 1]] - gs;
                const int x0 = max(0, vs + mp), xe = min(k, vs + gl + mp);
                if (x0 < xe) put_ref(x0, xe, (long long)gs + (x0 - mp - vs));
                vs += gl;
            }
        }
        {
            const int x0 = max(0, len_v + mp);
            if (x0 < k) put_ref(x0, k, (long long)last_end + (x0 - mp - len_v));
        }
        } // (general assembly)
        if (!ok) { // a base outside ACGT (or a k the packed form does not hold): the byte-wise path of the workgroup kernel
            W.fb_flag[g] = 1;
            continue;
        }
        ++evaluated;
        const U128 mk = mask128(2 * k);
        const U128 mform = shr128(U128{pairrev64(Lf.hi), pairrev64(Lf.lo)}, 2 * (64 - k));
        const U128 rc{~mform.lo & mk.lo, ~mform.hi & mk.hi};
        const U128 key = lt128(Lf, rc) ? Lf : rc;
        const u64 h = k == 35 ? xxh3_packed_fixed<35>(key.lo, key.hi) : xxh3_packed(key, k);
        const u64 idx = mod_size(h, bf.mod);
        const bool is_ref = mid_canon == 0;
        if (MODE == 0) {
            i32 w;
            if (is_ref) w = k == (int)map.klen ? map_value(map, key, h, idx) : 0;
            else w = (i32)bucket_count(map, bf.counts, idx);
            if (w > 0) atomicMax(&cov_out[a0 + mid_canon], (u32)w);
        } else if (is_ref) {
            if (MODE == 1) ++ref_rows;
            else map_insert_key(map, bf, key, h, row0 + wave_take(cursor), row0);
        } else if (MODE == 2) {
            atomicOr((unsigned long long *)&bf.words[idx >> 6], 1ULL << (idx & 63)); // BF::add_key, bloom_filter.hpp:81-85
            gate_set(bf, idx);
        }
    }
    if (MODE == 1) {
        for (int dd = 32; dd; dd >>= 1) ref_rows += __shfl_xor(ref_rows, dd, 64);
        if ((threadIdx.x & 63) == 0 && ref_rows) atomicAdd(cursor, (unsigned long long)ref_rows);
    }
    if (n_evaluated) {
        for (int dd = 32; dd; dd >>= 1) evaluated += __shfl_xor(evaluated, dd, 64);
        if ((threadIdx.x & 63) == 0 && evaluated) atomicAdd(n_evaluated, (unsigned long long)evaluated);
    }
}

// ---- tier 2: picks and evaluation of a chain in ONE kernel ----------------------------------------------------------------------
// fw_picks_kernel writes every distinct pick out as an item and fw_eval_kernel reads it back, fetches the chain's descriptor
// again and walks the members twice through six arrays of the panel -- 133 load instructions and 2,000 VALU instructions per
// wave of 64 k-mers at C5, which is what the pair's 1.4 ms per 6.9e6 k-mers is made of.  Here the wave that holds the picks
// evaluates them:
//   staging   the lanes of a chain's group put the chain's geometry into the wave's LDS area once: per member its position,
//             REF length, allele range and code width, and the offsets of all its alleles; they also look at every base a
//             window of the chain can hold -- the reference from k/2 before the first member to (k+1)/2 behind the last, and
//             every allele -- so that the assembly needs no test per piece
//   picks     as fw_picks_kernel: the distinct haplotype picks of the wave's chains in one LDS set
//   evaluate  the set's entries 64 at a time, a lane each: lengths and pieces from LDS, the bases from the packed reference
//             and the packed pool, canonical form, XXH3, lookup or insert
// A chain that does not fit its share of the staging area or of the set, or that has a base outside ACGT in reach, is listed
// for the pair above (fw_picks_kernel<true> takes the list a chain per wave, fw_eval_kernel its items) -- what happens to it
// there is what happened before this kernel existed, and the decision depends on the chain and the panel alone.
constexpr int FW_POOL = 2048; // bytes of staging area per wave, shared evenly by the chains the wave holds at a time
struct __attribute__((aligned(16))) FcHead {
    u64 cbase;
    i32 ref_len;
    u32 g, a0;
    i32 first_pos, last_end;
    u8 m, jm;
    u8 pad[2];
};
struct __attribute__((aligned(16))) FcMember {
    i32 pos;
    u32 rs, v;
    unsigned short off_at; // where the member's allele offsets start in the chain's table (A + 1 of them)
    u8 bits, A;
};
static_assert(sizeof(FcHead) == 32 && sizeof(FcMember) == 16, "staging layout");

template <int MODE>
__device__ __forceinline__ void fc_eval(const BlockBatch &B, const unsigned char *area, unsigned long long code, const BFView &bf, const MapView &map, u32 *cov_out,
                                        u8 *overflow, unsigned long long *cursor, u32 row0, u32 &evaluated, u32 &ref_rows)
{
    const FcHead hd = *(const FcHead *)area;
    const FcMember *mem = (const FcMember *)(area + sizeof(FcHead));
    const int m = hd.m, jm = hd.jm, k = B.k;
    const u32 *off = (const u32 *)(area + sizeof(FcHead) + sizeof(FcMember) * m);
    // lengths: virtual string V = A_0 R_0 A_1 ... A_{m-1}
    int len_v = 0, mid_pos = 0, mid_len = 0, sh = 0, prev_end = 0;
    u32 mid_allele = 0;
    for (int j = 0; j < m; ++j) {
        const FcMember mj = mem[j];
        const u32 a = (u32)(code >> sh) & ((1u << mj.bits) - 1);
        sh += mj.bits;
        const int al = (int)(off[mj.off_at + a + 1] - off[mj.off_at + a]);
        if (j) len_v += mj.pos - prev_end;
        if (j == jm) {
            mid_pos = len_v;
            mid_len = al;
            mid_allele = a;
        }
        len_v += al;
        prev_end = mj.pos + (int)mj.rs;
    }
    const int first_part = mid_pos + mid_len / 2;
    const int mp = k / 2 - first_part;                 // missing_prefix (negative: cut)
    const int ms = (k + 1) / 2 - (len_v - first_part); // missing_suffix
    if (hd.first_pos - (mp > 0 ? mp : 0) < 0 || hd.last_end + (ms > 0 ? ms : 0) > hd.ref_len) {
        if (MODE != 2) overflow[hd.g] = 1; // the reference clips or throws here: the host path's
        return;
    }
    U128 Lf{0, 0};
    auto put_codes = [&](const u64 *__restrict__ packed, int x, int xe, u64 at) { // W[x .. xe) = packed[at ..)
        while (x < xe) {
            const int n = xe - x < 32 ? xe - x : 32;
            const U128 sp = shl128(U128{ref_codes(packed, at, n), 0}, 2 * x);
            Lf.lo |= sp.lo;
            Lf.hi |= sp.hi;
            x += n;
            at += n;
        }
    };
    if (mp > 0) put_codes(B.ref2, 0, mp < k ? mp : k, hd.cbase + (u64)((long long)hd.first_pos - mp));
    int vs = 0;
    sh = 0;
    for (int j = 0; j < m; ++j) {
        const FcMember mj = mem[j];
        const u32 a = (u32)(code >> sh) & ((1u << mj.bits) - 1);
        sh += mj.bits;
        const u32 ao = off[mj.off_at + a];
        const int al = (int)(off[mj.off_at + a + 1] - ao);
        {
            const int x0 = max(0, vs + mp), xe = min(k, vs + al + mp);
            if (x0 < xe) {
                if (B.pool2) put_codes(B.pool2, x0, xe, (u64)ao + (u64)(x0 - mp - vs));
                else
                    for (int x = x0; x < xe; ++x) { // (every base was looked at when the chain was staged)
                        bool o;
                        const u64 c2 = acgt_code(B.pool[ao + (u32)(x - mp - vs)], &o);
                        if (x < 32) Lf.lo |= c2 << (2 * x);
                        else Lf.hi |= c2 << (2 * (x - 32));
                    }
            }
        }
        vs += al;
        if (j + 1 < m) {
            const int gs = mj.pos + (int)mj.rs;
            const int gl = mem[j + 1].pos - gs;
            const int x0 = max(0, vs + mp), xe = min(k, vs + gl + mp);
            if (x0 < xe) put_codes(B.ref2, x0, xe, hd.cbase + (u64)((long long)gs + (x0 - mp - vs)));
            vs += gl;
        }
        if (vs + mp >= k) break; // the window is full
    }
    {
        const int x0 = max(0, len_v + mp);
        if (x0 < k) put_codes(B.ref2, x0, k, hd.cbase + (u64)((long long)hd.last_end + (x0 - mp - len_v)));
    }
    ++evaluated;
    const u32 mid_canon = B.canon[hd.a0 + mid_allele];
    const U128 mk = mask128(2 * k);
    const U128 mform = shr128(U128{pairrev64(Lf.hi), pairrev64(Lf.lo)}, 2 * (64 - k));
    const U128 rc{~mform.lo & mk.lo, ~mform.hi & mk.hi};
    const U128 key = lt128(Lf, rc) ? Lf : rc;
    const u64 h = k == 35 ? xxh3_packed_fixed<35>(key.lo, key.hi) : xxh3_packed(key, k);
    const u64 idx = mod_size(h, bf.mod);
    const bool is_ref = mid_canon == 0;
    if (MODE == 0) {
        i32 w;
        if (is_ref) w = k == (int)map.klen ? map_value(map, key, h, idx) : 0;
        else w = (i32)bucket_count(map, bf.counts, idx);
        if (w > 0) atomicMax(&cov_out[hd.a0 + mid_canon], (u32)w);
    } else if (is_ref) {
        if (MODE == 1) ++ref_rows;
        else map_insert_key(map, bf, key, h, row0 + wave_take(cursor), row0);
    } else if (MODE == 2) {
        atomicOr((unsigned long long *)&bf.words[idx >> 6], 1ULL << (idx & 63)); // BF::add_key, bloom_filter.hpp:81-85
        gate_set(bf, idx);
    }
}

constexpr int FC_SET = 512; // slots of a wave's set of distinct picks in fw_chain_kernel
// A set entry is 32 bits: the pick's code (at most FC_CODE_BITS wide: six members of up to sixteen alleles; a wider chain takes the
// list), the sliding mark, the chain's number in the wave -- half the LDS of 64-bit entries, one more workgroup per CU
constexpr int FC_CODE_BITS = 24;
constexpr u32 FC_SLIDE_IN = 1u << FC_CODE_BITS;
constexpr int FC_GRP_SHIFT = FC_CODE_BITS + 1;
template <int MODE>
__global__ void __launch_bounds__(TPB, 6) fw_chain_kernel(BlockBatch B, FlatWork W, int G, BFView bf, MapView map, u32 *cov_out, u8 *overflow, unsigned long long *cursor,
                                                       u32 row0, unsigned long long *n_evaluated)
{
    __shared__ u32 sh_set[FW_WAVES][FC_SET];                // key + 1, 0 = free
    __shared__ unsigned short sh_list[FW_WAVES][FC_SET];    // slots taken, in the order they were taken
    __shared__ u32 sh_n[FW_WAVES];
    __shared__ u32 sh_cnt[FW_WAVES][32];                    // distinct picks per chain of the wave
    __shared__ u32 sh_state[FW_WAVES][32];                  // per chain of the wave: 0 staged, else it goes to the list
    __shared__ __attribute__((aligned(16))) unsigned char sh_pool[FW_WAVES][FW_POOL];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int grp = lane / G, sub = lane % G, n_grp = 64 / G;
    const int budget = FW_POOL / n_grp;
    const int k = B.k;
    u32 *set = sh_set[wave];
    unsigned short *list = sh_list[wave];
    unsigned char *area = sh_pool[wave] + grp * budget;
    FcHead *hd = (FcHead *)area;
    FcMember *mem = (FcMember *)(area + sizeof(FcHead));
    for (int i = lane; i < FC_SET; i += 64) set[i] = 0;
    if (lane == 0) sh_n[wave] = 0;
    if (lane < 32) sh_cnt[wave][lane] = 0;
    wave_sync();
    const u32 share = (u32)(FC_SET * 3 / 4 / n_grp); // distinct picks a chain may have here
    const u64 n_combs = fw_chains_todo(W);
    const u64 n_waves = (u64)gridDim.x * FW_WAVES;
    volatile u32 *my_cnt = &sh_cnt[wave][grp];
    u32 evaluated = 0, ref_rows = 0;
    auto insert = [&](u32 key) {
        u32 at = ((key * 0x9E3779B9u) >> 20) & (FC_SET - 1);
        for (int tries = 0; tries < FC_SET; ++tries) {
            const u32 seen = atomicCAS(&set[at], 0u, key + 1);
            if (seen == 0u) {
                atomicAdd(&sh_cnt[wave][grp], 1u);
                const u32 q = atomicAdd(&sh_n[wave], 1u);
                if (q < FC_SET) list[q] = (unsigned short)at;
                return;
            }
            if (seen == key + 1) return;
            at = (at + 1) & (FC_SET - 1);
        }
    };
    for (u64 c0 = ((u64)blockIdx.x * FW_WAVES + wave) * n_grp; c0 < n_combs; c0 += n_waves * n_grp) {
        const u64 cpos = c0 + grp;
        const u64 ci = cpos < n_combs ? (W.order ? (u64)W.order[cpos] : cpos) : 0;
        // (the descriptor's head and its mark; the members' offsets are read where they are used, a byte each: an array indexed
        // by a loop counter would live in scratch memory)
        uint4 dh{0, 0, 0, 0};
        u32 dmark = 0;
        if (cpos < n_combs) {
            dh = *(const uint4 *)&W.combs[ci];
            dmark = *(const u32 *)&W.combs[ci].rel[18] >> 24;
        }
        const signed char *rel = W.combs[ci].rel;
        const int m = (int)(dh.z & 255u);
        const u32 g = dh.x, cid = dh.y;
        const bool valid = cpos < n_combs && m > 0 && dmark != 2 && !W.fb_flag[g]; // (m == 0: a reservation that did not fit; mark 2: fw_snp_kernel took the chain)
        // ---- staging: members, then (one lane) where each member's allele offsets go, then the offsets and the look at every base
        const bool room = (int)(sizeof(FcHead) + sizeof(FcMember) * m) <= budget;
        if (sub == 0) sh_state[wave][grp] = valid && room ? 0u : 1u;
        if (valid && room) {
            for (int j = sub; j < m; j += G) {
                const u32 v = g + rel[j];
                const u32 A = B.var_allele_off[v + 1] - B.var_allele_off[v];
                mem[j] = FcMember{B.pos[v], B.ref_size[v], v, (unsigned short)0, (u8)fw_bits(A), (u8)A};
            }
            if (sub == 0) {
                FcHead h{};
                h.cbase = B.contig_base[cid];
                h.ref_len = (i32)B.contig_len[cid];
                h.g = g;
                h.a0 = B.var_allele_off[g];
                h.m = (u8)m;
                h.jm = (u8)((dh.z >> 8) & 255u);
                *hd = h;
            }
        }
        wave_sync();
        if (valid && room && sub == 0) {
            u32 acc = 0, bit_at = 0;
            bool wide = false; // (a member of 128 alleles or more: its count does not fit the entry, its genotypes not the pick's code)
            for (int j = 0; j < m; ++j) {
                mem[j].off_at = (unsigned short)acc;
                acc += (u32)mem[j].A + 1;
                wide = wide || mem[j].bits > 7;
                bit_at += mem[j].bits;
            }
            hd->first_pos = mem[0].pos;
            hd->last_end = mem[m - 1].pos + (i32)mem[m - 1].rs;
            if (wide || bit_at > (u32)FC_CODE_BITS || (int)(sizeof(FcHead) + sizeof(FcMember) * m + 4 * acc) > budget) sh_state[wave][grp] = 1;
        }
        wave_sync();
        if (valid && sh_state[wave][grp] == 0) {
            u32 *off = (u32 *)(area + sizeof(FcHead) + sizeof(FcMember) * m);
            bool bad = false;
            for (int j = 0; j < m; ++j) {
                const FcMember mj = mem[j];
                const u32 s0 = B.var_allele_off[mj.v];
                for (u32 a = (u32)sub; a < mj.A; a += (u32)G) {
                    const u32 ao = B.allele_off[s0 + a], an = B.allele_off[s0 + a + 1];
                    off[mj.off_at + a] = ao;
                    if (a + 1 == mj.A) off[mj.off_at + a + 1] = an;
                    if (B.pool2)
                        for (u32 x = ao; x < an; x += 32) bad = bad || ref_bad(B.poolbad, (u64)x, an - x < 32 ? (int)(an - x) : 32);
                    else
                        for (u32 x = ao; x < an; ++x) {
                            bool o;
                            acgt_code(B.pool[x], &o);
                            bad = bad || !o;
                        }
                }
            }
            const i32 lo = max(0, hd->first_pos - k / 2), hi = min(hd->ref_len, hd->last_end + (k + 1) / 2);
            for (i32 x = lo + 32 * sub; x < hi; x += 32 * G) bad = bad || ref_bad(B.refbad, hd->cbase + (u64)x, hi - x < 32 ? hi - x : 32);
            if (bad) sh_state[wave][grp] = 1;
        }
        wave_sync();
        const bool staged = valid && sh_state[wave][grp] == 0;
        // ---- picks (build_alleles_combs + combine_haplotypes, var_block.hpp:709-786)
        bool fail = false; // the record goes to the workgroup kernel
        if (staged) {
            const u32 *off = (const u32 *)(area + sizeof(FcHead) + sizeof(FcMember) * m);
            for (u32 s = sub; s < B.n_samples && !fail; s += G) {
                if (*my_cnt > share) break; // the chain outgrew its share of the set (all its lanes see that sooner or later)
                bool phased = true;
                unsigned long long c1 = 0, c2 = 0, bounds = 1; // bounds: bit set at every member's first code bit, and behind the last
                int sh = 0;
                for (int j = 0; j < m; ++j) {
                    const int bits = mem[j].bits;
                    if (sh + bits > FW_CODE_BITS) {
                        fail = true;
                        break;
                    }
                    const u32 gt = gt_at(B, mem[j].v, s);
                    phased = phased && ((gt >> 14) & 1);
                    c1 |= (unsigned long long)(gt & 127) << sh;
                    c2 |= (unsigned long long)((gt >> 7) & 127) << sh;
                    sh += bits;
                    bounds |= 1ULL << sh;
                }
                if (fail) break;
                const u32 tag = (u32)grp << FC_GRP_SHIFT;
                if (m == 1) { // an allele of k bases or more on its own is a SLIDING signature (var_block.hpp:130-144): its own kind of item
                    const u32 l1 = off[(u32)c1 + 1] - off[(u32)c1], l2 = off[(u32)c2 + 1] - off[(u32)c2];
                    if ((int)l1 >= k) c1 |= FC_SLIDE_IN;
                    if ((int)l2 >= k) c2 |= FC_SLIDE_IN;
                }
                if (B.haploid) insert((u32)c1 | tag);
                else if (phased || m == 1) { // (the mixes of a chain of one are its two alleles)
                    insert((u32)c1 | tag);
                    insert((u32)c2 | tag);
                } else if (m > FW_MAXU) fail = true;
                else { // every mix of the two haplotypes (combine_haplotypes): Gray-code walk, one member's field flipped per step
                    // (only the members at which the two haplotypes differ: 2^h codes for h of them, not 2^m steps of which most change nothing)
                    const unsigned long long diff = c1 ^ c2;
                    u32 het = 0;
                    {
                        unsigned long long t = bounds;
                        for (int j = 0; j < m; ++j) {
                            const int lo = __ffsll((unsigned long long)t) - 1;
                            t &= t - 1;
                            const int hi = __ffsll((unsigned long long)t) - 1;
                            if (diff & (((1ULL << (hi - lo)) - 1) << lo)) het |= 1u << j;
                        }
                    }
                    const int n_het = __popc(het);
                    unsigned long long code = c1;
                    insert((u32)code | tag);
                    for (u32 i = 1; i < (1u << n_het); ++i) {
                        u32 hm = het; // the member this step flips: the (number of trailing zeros of i)-th of the differing ones
                        for (int q = __ffs((int)i) - 1; q > 0; --q) hm &= hm - 1;
                        const int j = __ffs((int)hm) - 1;
                        unsigned long long t = bounds;
                        for (int q = 0; q < j; ++q) t &= t - 1;
                        const int lo = __ffsll((unsigned long long)t) - 1;
                        t &= t - 1;
                        const int hi = __ffsll((unsigned long long)t) - 1;
                        code ^= diff & (((1ULL << (hi - lo)) - 1) << lo);
                        insert((u32)code | tag);
                        if (*my_cnt > share) break;
                    }
                }
            }
        }
        wave_sync();
        const u32 n = sh_n[wave];
        const bool all_fail = n > FC_SET; // (cannot happen: every chain stops at its share; kept as a guard)
        const bool grew = staged && sh_cnt[wave][grp] > share; // the chain outgrew its share of the set
        if (valid && sub == 0 && (!staged || grew) && !fail) { // -> the list of chains fw_picks_kernel<true> takes, a wave each
            const unsigned long long at = atomicAdd(&W.counters[3], 1ULL);
            if (at < W.comb_cap) W.retry[at] = (u32)ci;
            else fail = true;
        }
        u64 skipped = 0; // chains (by their number in the wave) whose picks are not evaluated here
        {
            const u64 fl = __ballot(fail || grew);
            for (int q = 0; q < n_grp; ++q) {
                const u64 qm = (G == 64 ? ~0ULL : ((1ULL << G) - 1)) << (q * G);
                if (all_fail || (fl & qm)) skipped |= 1ULL << q;
            }
        }
        const u64 to_wg = __ballot(fail); // lanes whose chain's record goes to the workgroup kernel
        if (valid && sub == 0 && (all_fail || ((to_wg >> (grp * G)) & (G == 64 ? ~0ULL : ((1ULL << G) - 1))))) W.fb_flag[g] = 1;
        const u32 n_used = n < FC_SET ? n : FC_SET;
        if (!all_fail)
            for (u32 i0 = 0; i0 < n_used; i0 += 64) { // the set's entries, 64 at a time, a lane each
                const u32 i = i0 + lane;
                const u32 key = i < n_used ? set[list[i]] - 1 : 0u;
                const int kg = (int)(key >> FC_GRP_SHIFT) & 31;
                const bool live = i < n_used && !((skipped >> kg) & 1);
                const unsigned long long code = key & (FC_SLIDE_IN - 1);
                const unsigned char *karea = sh_pool[wave] + kg * budget;
                const u32 kci = (u32)__shfl((int)(u32)ci, kg * G, 64); // the descriptor of the key's chain
                if (live && (key & FC_SLIDE_IN)) {
                    const unsigned long long at = atomicAdd(&W.counters[2], 1ULL);
                    if (at < W.slide_cap) W.slides[at] = PickItem{kci, 0u, code | FW_SLIDE};
                    else W.fb_flag[((const FcHead *)karea)->g] = 1;
                } else if (live)
                    fc_eval<MODE>(B, karea, code, bf, map, cov_out, overflow, cursor, row0, evaluated, ref_rows);
            }
        for (u32 i = lane; i < n_used; i += 64) set[list[i]] = 0;
        if (lane < 32) sh_cnt[wave][lane] = 0;
        if (n > FC_SET) // (the list lost entries: clear the whole set)
            for (int i = lane; i < FC_SET; i += 64) set[i] = 0;
        wave_sync();
        if (lane == 0) sh_n[wave] = 0;
        wave_sync();
    }
    if (MODE == 1) {
        for (int dd = 32; dd; dd >>= 1) ref_rows += __shfl_xor(ref_rows, dd, 64);
        if (lane == 0 && ref_rows) atomicAdd(cursor, (unsigned long long)ref_rows);
    }
    if (n_evaluated) {
        for (int dd = 32; dd; dd >>= 1) evaluated += __shfl_xor(evaluated, dd, 64);
        if (lane == 0 && evaluated) atomicAdd(n_evaluated, (unsigned long long)evaluated);
    }
}

// Sliding signatures (var_block.hpp:130-144): the chain is the variant alone and the allele has k bases or more -- every k-mer of
// the allele, coverage = truncating running mean over those with a weight, in order (main.cpp:162-176).  SIXTEEN LANES per item,
// four items per wave: the lanes take the k-mers (assembly, hash, lookup or insert) sixteen at a time, the group's first lane
// folds their weights in order.  (An insertion of 35 to 60 bases at k = 35 has 1 to 26 k-mers, 13 on average: a whole wave
// per item left four lanes in five idle.)
constexpr int FW_SLIDE_G = 16;
template <int MODE>
__global__ void __launch_bounds__(TPB) fw_slide_kernel(BlockBatch B, FlatWork W, BFView bf, MapView map, u32 *cov_out, unsigned long long *cursor, u32 row0,
                                                       unsigned long long *n_evaluated)
{
    constexpr int G = FW_SLIDE_G, N_GRP = 64 / G;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, grp = lane / G, sub = lane % G;
    const u64 n_items = min((unsigned long long)W.slide_cap, W.counters[2]);
    const u64 n_waves = (u64)gridDim.x * FW_WAVES;
    const u64 gmask = ((1ULL << G) - 1) << (grp * G);
    const int k = B.k;
    for (u64 i0 = ((u64)blockIdx.x * FW_WAVES + wave) * N_GRP; i0 < n_items; i0 += n_waves * N_GRP) { // (whole waves: ballots and shuffles inside)
        const u64 it = i0 + grp;
        const bool have = it < n_items;
        PickItem item{0u, 0u, 0ULL};
        if (have) item = W.slides[it];
        const u32 g = have ? W.combs[item.comb].g : 0u;
        bool live = have && !W.fb_flag[g];
        const u32 a0 = live ? B.var_allele_off[g] : 0u;
        const u32 a = (u32)(item.code & 127);
        const u32 canon = live ? (u32)B.canon[a0 + a] : 0u;
        const u32 ao = live ? B.allele_off[a0 + a] : 0u;
        const u8 *ap = B.pool + ao;
        const int al = live ? (int)(B.allele_off[a0 + a + 1] - ao) : 0;
        const bool is_ref = canon == 0;
        const int n_kmers = live ? al - k + 1 : 0;
        // first: is every base ACGT?  (nothing is counted or inserted for a record the workgroup kernel will redo)
        bool good = k >= 17 && k <= MG_MAX_PACKED_K;
        if (B.pool2) {
            for (int x = 32 * sub; x < al; x += 32 * G) good = good && !ref_bad(B.poolbad, (u64)ao + (u64)x, al - x < 32 ? al - x : 32);
        } else
            for (int x = sub; x < al; x += G) {
                bool o;
                acgt_code(ap[x], &o);
                good = good && o;
            }
        const bool group_bad = (__ballot(live && !good) & gmask) != 0;
        if (live && group_bad) {
            if (sub == 0) W.fb_flag[g] = 1;
            live = false;
        }
        u32 curr = 0;
        i32 nn = 0;
        int longest = live ? n_kmers : 0; // (the wave walks to its longest item)
        for (int d = 32; d; d >>= 1) longest = max(longest, __shfl_xor(longest, d, 64));
        for (int p0 = 0; p0 < longest; p0 += G) {
            const int p = p0 + sub;
            i32 w = 0;
            if (live && p < n_kmers) {
                U128 Lf{0, 0};
                if (B.pool2) { // the window out of the packed pool
                    Lf.lo = ref_codes(B.pool2, (u64)ao + (u64)p, k < 32 ? k : 32);
                    if (k > 32) Lf.hi = ref_codes(B.pool2, (u64)ao + (u64)p + 32, k - 32);
                } else
                    for (int x = 0; x < k; ++x) {
                        bool o;
                        const u64 code = acgt_code(ap[p + x], &o);
                        if (x < 32) Lf.lo |= code << (2 * x);
                        else Lf.hi |= code << (2 * (x - 32));
                    }
                const U128 mk = mask128(2 * k);
                const U128 mform = shr128(U128{pairrev64(Lf.hi), pairrev64(Lf.lo)}, 2 * (64 - k));
                const U128 rc{~mform.lo & mk.lo, ~mform.hi & mk.hi};
                const U128 key = lt128(Lf, rc) ? Lf : rc;
                const u64 h = k == 35 ? xxh3_packed_fixed<35>(key.lo, key.hi) : xxh3_packed(key, k);
                const u64 idx = mod_size(h, bf.mod);
                if (MODE == 0) {
                    if (is_ref) w = k == (int)map.klen ? map_value(map, key, h, idx) : 0;
                    else w = (i32)bucket_count(map, bf.counts, idx);
                } else if (MODE == 2) {
                    if (is_ref) map_insert_key(map, bf, key, h, row0 + wave_take(cursor), row0);
                    else {
                        atomicOr((unsigned long long *)&bf.words[idx >> 6], 1ULL << (idx & 63));
                        gate_set(bf, idx);
                    }
                }
            }
            if (MODE == 0)
                for (int j = 0; j < G; ++j) { // in order: the mean truncates at every step (every group's first lane folds its own)
                    const i32 wj = __shfl(w, grp * G + j, 64);
                    if (sub == 0 && wj > 0) {
                        curr = (curr * (u32)nn + (u32)wj) / (u32)(nn + 1);
                        ++nn;
                    }
                }
        }
        if (live && sub == 0) {
            if (MODE == 0 && curr) atomicMax(&cov_out[a0 + canon], curr);
            if (MODE == 1 && is_ref) atomicAdd(cursor, (unsigned long long)n_kmers);
            if (n_evaluated) atomicAdd(n_evaluated, (unsigned long long)n_kmers);
        }
    }
}

// A/B and tests (use_flat_tier = 0): every general record is handed on to the workgroup kernel
__global__ void __launch_bounds__(TPB) flag_all_kernel(const u32 *__restrict__ gen_list, const unsigned long long *__restrict__ gen_count, u8 *fb_flag, u32 *cov_out,
                                                       const u32 *__restrict__ var_allele_off)
{
    const u64 n = *gen_count;
    const u64 n_threads = (u64)gridDim.x * TPB;
    for (u64 i = (u64)blockIdx.x * TPB + threadIdx.x; i < n; i += n_threads) {
        const u32 g = gen_list[i];
        fb_flag[g] = 1;
        if (cov_out)
            for (u32 a = var_allele_off[g]; a < var_allele_off[g + 1]; ++a) cov_out[a] = 0;
    }
}
// the general records tier 2 handed on -> a list for the workgroup kernel
__global__ void __launch_bounds__(TPB) fb_compact_kernel(const u32 *__restrict__ gen_list, const unsigned long long *__restrict__ gen_count, const u8 *__restrict__ fb_flag,
                                                         u32 *fb_list, unsigned long long *fb_count)
{
    const u64 n = *gen_count;
    const u64 n_threads = (u64)gridDim.x * TPB;
    for (u64 base = (u64)blockIdx.x * TPB; base < n; base += n_threads) { // (whole waves stay together: list_append ballots)
        const u64 i = base + threadIdx.x;
        const u32 g = i < n ? gen_list[i] : 0;
        list_append(i < n && fb_flag[g], g, fb_list, fb_count);
    }
}
// coverages as set_variant_coverage leaves them (through a float, var_block.hpp:84); zero where the record goes to the host
__global__ void __launch_bounds__(TPB) fw_finish_kernel(BlockBatch B, const u32 *__restrict__ gen_list, const unsigned long long *__restrict__ gen_count,
                                                        const u8 *__restrict__ overflow, u32 *cov_out)
{
    const u64 n = *gen_count;
    const u64 n_threads = (u64)gridDim.x * TPB;
    for (u64 i = (u64)blockIdx.x * TPB + threadIdx.x; i < n; i += n_threads) {
        const u32 g = gen_list[i];
        const u32 a0 = B.var_allele_off[g], A = B.var_allele_off[g + 1] - a0;
        const bool host = overflow[g];
        for (u32 a = 0; a < A; ++a) cov_out[a0 + a] = host ? 0 : (u32)(float)cov_out[a0 + a];
    }
}
