// Panel genotypes decoded from VCF text on the device: Variant::extract_genotypes (variant.hpp:158-211) over what htslib's
// bcf_get_genotypes hands it for text GT fields, restated from the text itself.
//
// A panel record is its nine fixed columns and then one column per sample -- 27,934 of them on the SARS-CoV-2 panel
// (BASELINE config C1), 56 KB of "0\t0\t1\t0..." per record, 847 MB per file -- and nearly all of what the host spent on such a
// file was turning that text into genotype pairs, sample by sample.  Here the host finds the line's ends and its ninth tab, the
// sample columns cross PCIe as they are, and a workgroup per record turns them into the record loop's sparse layout
// (mg_panel_dev.sp_*: the samples whose word is not the default) plus what the host still wants to know per record: which
// raw allele numbers occur at all (the lone-variant path's presence mask) and the largest one (the reader's range check).
//
// What extract_genotypes sees, per kept sample i (after `-s`): bcf_get_genotypes returns `ploidy` ints per sample, ploidy =
// the largest number of alleles any kept sample of the RECORD has; shorter samples are padded with vector_end.  The
// function reads curr_gt[0] and curr_gt[1] of the flat array at i * ploidy: with ploidy 1 the second is the NEXT sample's
// first value (and past the last sample, nothing: restated as vector_end).  A second value of
// vector_end makes the sample (a, a) phased; otherwise (first, second) with the phase bit of the second.  Missing alleles
// (".", -1) and negative numbers read as 0.  The word is a1 | a2 << 7 | phased << 14, or a1 | 1 << 14 in haploid mode
// (var_block.hpp:751 reads the first allele only).
#pragma once
#include "kmer_dev.h"

namespace {
using namespace mg;

constexpr int GT_TPB = 256;
constexpr int GT_END = -2; // bcf_int32_vector_end as the token arrays hold it (a literal "-2" in the text reads the same way, as on the host)

// one kept sample's tokens: first value, second value (both clamped into i16), how many there were (saturating), phase of the second
__device__ __forceinline__ u64 gt_pack(int t0, int t1, int cnt, bool ph1)
{
    return (u64)(u32)(t0 & 0xFFFF) | (u64)(u32)(t1 & 0xFFFF) << 16 | (u64)(u32)(cnt > 255 ? 255 : cnt) << 32 | (u64)(ph1 ? 1 : 0) << 40;
}
__device__ __forceinline__ int gt_t0(u64 e) { return (int)(short)(e & 0xFFFF); }
__device__ __forceinline__ int gt_t1(u64 e) { return (int)(short)((e >> 16) & 0xFFFF); }
__device__ __forceinline__ int gt_cnt(u64 e) { return (int)((e >> 32) & 0xFF); }
__device__ __forceinline__ bool gt_ph1(u64 e) { return (e >> 40) & 1; }
__device__ __forceinline__ int gt_clamp(long v) { return v > 32767 ? 32767 : v < GT_END ? -1 : (int)v; } // (below -2: negative, reads as 0 like -1)

// the GT sub-field (the gi-th ':'-separated one) of the column that starts at p; e = end of the record
__device__ inline u64 gt_column(const char *p, const char *e, int gi)
{
    int sub = 0;
    while (sub < gi && p < e && *p != '\t') {
        if (*p == ':') ++sub;
        ++p;
    }
    if (sub < gi) return gt_pack(-1, -1, 1, false); // sub-field absent: GT "."
    int t0 = -1, t1 = -1, cnt = 0;
    bool ph = false, ph1 = false;
    for (;;) { // tokens separated by '/' or '|'; each read as atoi would (sign, leading digits)
        const char *a = p;
        while (p < e && *p != '/' && *p != '|' && *p != ':' && *p != '\t') ++p;
        long val = -1;
        if (p > a && !(p == a + 1 && *a == '.')) {
            const char *q = a;
            bool neg = false;
            if (*q == '-' || *q == '+') neg = *q++ == '-';
            long acc = 0;
            while (q < p && *q >= '0' && *q <= '9') {
                acc = acc * 10 + (*q++ - '0');
                if (acc > 1000000) acc = 1000000; // (far beyond any allele number: the record is handed to the host)
            }
            val = neg ? -acc : acc;
        }
        if (cnt == 0) t0 = gt_clamp(val);
        else if (cnt == 1) {
            t1 = gt_clamp(val);
            ph1 = ph;
        }
        ++cnt;
        if (p < e && (*p == '/' || *p == '|')) {
            ph = *p == '|';
            ++p;
            continue;
        }
        break;
    }
    return gt_pack(t0, t1, cnt, ph1);
}

struct GtStats { // per record
    unsigned long long raw_mask; // raw allele numbers that occur (mod 64)
    u32 max_allele;
    u32 n_phased0;   // words equal to 0|0 phased
    u32 n_unphased0; // words equal to 0/0
};

// One workgroup per record at a time (a persistent grid walks the batch).  tok = this workgroup's n_keep entries of scratch.
__global__ void __launch_bounds__(GT_TPB) gt_decode_kernel(const char *__restrict__ text, const unsigned long long *__restrict__ span_off,
                                                           const u32 *__restrict__ span_len, const i32 *__restrict__ gt_index, u32 n_records, u32 n_columns,
                                                           const u32 *__restrict__ keep_rank /* [n_columns]: index among the kept samples, or ~0 */, u32 n_keep,
                                                           int haploid, unsigned long long *tok_all, uint16_t *words, GtStats *stats)
{
    __shared__ u32 sh_scan[GT_TPB];
    __shared__ u32 sh_ge2, sh_max, sh_p0, sh_u0;
    __shared__ unsigned long long sh_mask;
    unsigned long long *tok = tok_all + (u64)blockIdx.x * n_keep;
    const int tid = threadIdx.x;
    for (u32 r = blockIdx.x; r < n_records; r += gridDim.x) {
        const char *base = text + span_off[r];
        const u32 len = span_len[r];
        const char *end = base + len;
        const int gi = gt_index[r];
        for (u32 i = tid; i < n_keep; i += GT_TPB) tok[i] = gt_pack(-1, -1, 1, false); // a record with fewer columns than the header: GT "."
        if (tid == 0) {
            sh_ge2 = 0;
            sh_max = 0;
            sh_p0 = sh_u0 = 0;
            sh_mask = 0;
        }
        __syncthreads();
        auto column = [&](u32 c, u32 start) { // column c starts at byte `start`
            if (c >= n_columns) return;
            const u32 rank = keep_rank[c];
            if (rank != 0xFFFFFFFFu) tok[rank] = gt_column(base + start, end, gi);
        };
        if (tid == 0 && len) column(0, 0);
        u32 cols_before = 0; // tabs in front of this tile
        for (u32 t0 = 0; t0 < len; t0 += GT_TPB * 16) {
            const u32 off = t0 + (u32)tid * 16;
            u32 tabs = 0; // bit b: byte off + b is a tab
            if (off < len) {
                const u32 n = len - off < 16 ? len - off : 16;
                for (u32 b = 0; b < n; ++b) tabs |= (u32)(base[off + b] == '\t') << b;
            }
            // exclusive scan of the tab counts over the workgroup
            const u32 cnt = (u32)__popc(tabs);
            u32 incl = cnt;
            for (int d = 1; d < 64; d <<= 1) {
                const u32 o = (u32)__shfl_up((int)incl, d, 64);
                if ((tid & 63) >= d) incl += o;
            }
            if ((tid & 63) == 63) sh_scan[tid >> 6] = incl;
            __syncthreads();
            u32 wave_base = 0, total = 0;
            for (int w = 0; w < GT_TPB / 64; ++w) {
                if (w < (tid >> 6)) wave_base += sh_scan[w];
                total += sh_scan[w];
            }
            u32 c = cols_before + wave_base + incl - cnt; // tabs in front of this thread's bytes
            u32 m = tabs;
            while (m) {
                const int b = __ffs((int)m) - 1;
                m &= m - 1;
                ++c;
                column(c, off + (u32)b + 1);
            }
            cols_before += total;
            __syncthreads(); // (sh_scan is rewritten by the next tile)
        }
        __syncthreads();
        // ploidy 1 or more than 1
        u32 ge2 = 0;
        for (u32 i = tid; i < n_keep; i += GT_TPB) ge2 |= (u32)(gt_cnt(tok[i]) >= 2);
        if (ge2) sh_ge2 = 1;
        __syncthreads();
        const bool p1 = sh_ge2 == 0;
        u32 mx = 0, np0 = 0, nu0 = 0;
        unsigned long long mask = 0;
        for (u32 i = tid; i < n_keep; i += GT_TPB) {
            const u64 e = tok[i];
            const int first = gt_t0(e);
            int second;
            bool second_ph;
            if (!p1) {
                second = gt_cnt(e) >= 2 ? gt_t1(e) : GT_END;
                second_ph = gt_ph1(e);
            } else if (i + 1 < n_keep) { // curr_gt[1] of a ploidy-1 record: the next sample's value (variant.hpp:184)
                second = gt_t0(tok[i + 1]);
                second_ph = false;
            } else {
                second = GT_END;
                second_ph = false;
            }
            int a1, a2;
            bool ph;
            if (second == GT_END) {
                a1 = a2 = first;
                ph = true;
            } else {
                a1 = first;
                a2 = second;
                ph = second_ph;
            }
            if (a1 < 0) a1 = 0;
            if (a2 < 0) a2 = 0;
            const u32 big = (u32)(a1 > a2 ? a1 : a2);
            mx = mx > big ? mx : big;
            mask |= 1ULL << (a1 & 63);
            if (!haploid) mask |= 1ULL << (a2 & 63);
            const u32 w = haploid ? ((u32)(a1 & 127) | 1u << 14) : ((u32)(a1 & 127) | (u32)(a2 & 127) << 7 | (u32)ph << 14);
            np0 += w == (1u << 14);
            nu0 += w == 0;
            words[(u64)r * n_keep + i] = (uint16_t)w;
        }
        for (int d = 32; d; d >>= 1) {
            const u32 o = (u32)__shfl_xor((int)mx, d, 64);
            mx = mx > o ? mx : o;
            np0 += (u32)__shfl_xor((int)np0, d, 64);
            nu0 += (u32)__shfl_xor((int)nu0, d, 64);
            mask |= (unsigned long long)__shfl_xor((long long)mask, d, 64);
        }
        if ((tid & 63) == 0) {
            atomicMax(&sh_max, mx);
            atomicAdd(&sh_p0, np0);
            atomicAdd(&sh_u0, nu0);
            atomicOr(&sh_mask, mask);
        }
        __syncthreads();
        if (tid == 0) stats[r] = GtStats{sh_mask, sh_max, sh_p0, sh_u0};
        __syncthreads(); // (tok and the shared cells are reused by the next record)
    }
}

// the words other than `dflt`, per record in sample order, at sp_off[r]
__global__ void __launch_bounds__(GT_TPB) gt_compact_kernel(const uint16_t *__restrict__ words, u32 n_records, u32 n_keep, u32 dflt, const u32 *__restrict__ sp_off,
                                                            u32 *__restrict__ sp_sample, uint16_t *__restrict__ sp_gt)
{
    __shared__ u32 sh_scan[GT_TPB / 64];
    __shared__ u32 sh_base;
    const int tid = threadIdx.x;
    for (u32 r = blockIdx.x; r < n_records; r += gridDim.x) {
        if (tid == 0) sh_base = sp_off[r];
        __syncthreads();
        for (u32 i0 = 0; i0 < n_keep; i0 += GT_TPB) {
            const u32 i = i0 + (u32)tid;
            const u32 w = i < n_keep ? (u32)words[(u64)r * n_keep + i] : dflt;
            const bool take = w != dflt;
            const u64 m = __ballot(take);
            if ((tid & 63) == 0) sh_scan[tid >> 6] = (u32)__popcll(m);
            __syncthreads();
            u32 before = sh_base, total = 0;
            for (int q = 0; q < GT_TPB / 64; ++q) {
                if (q < (tid >> 6)) before += sh_scan[q];
                total += sh_scan[q];
            }
            if (take) {
                const u32 at = before + (u32)__popcll(m & ((1ULL << (tid & 63)) - 1));
                sp_sample[at] = i;
                sp_gt[at] = (uint16_t)w;
            }
            __syncthreads();
            if (tid == 0) sh_base += total;
            __syncthreads();
        }
    }
}

} // namespace
