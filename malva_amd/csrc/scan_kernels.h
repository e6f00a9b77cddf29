// scan_kernels.h -- the index-time reference-context scan and the call-time KMC scan (filter / probe / hits)
// Part of the malva_hip translation unit: included by malva_hip.hip inside its anonymous namespace, after
// geno_dev.h (which brings xxh3_dev.h and kmer_dev.h).  See DESIGN.md section 4 for the kernels' rooflines.
#pragma once

// ---- H11: reference-context scan (main.cpp:383-401) --------------------------
// One thread per window start p; the workgroup stages its TPB + ref_k - 1 bytes
// of the contig through LDS once.  Windows are full length (contigs shorter
// than ref_k are handled by the host wrapper with the row kernels).
struct LdsIn {
    const u8 *p;
    __device__ __forceinline__ u32 operator()(int i) const { return p[i]; }
};
// is window w of a contig (its first base at `at` in the packed reference) one the packed kernel leaves to the byte-wise one?
__device__ __forceinline__ bool ref_window_slow_any(const u32 *__restrict__ refbad, u64 at, u64 w, int k, int ref_k)
{
    if (((ref_k - k) & 1) && w >= 1 && w < (u64)k) return true;
    for (int done = 0; done < ref_k; done += 32) {
        const int n = ref_k - done < 32 ? ref_k - done : 32;
        const u64 s = at + done, wd = s >> 5;
        const int o = (int)(s & 31);
        const u64 b = ((u64)refbad[wd] | ((u64)refbad[wd + 1] << 32)) >> o;
        if (b & (n >= 32 ? 0xFFFFFFFFULL : ((1ULL << n) - 1))) return true;
    }
    return false;
}
// The reference slides its centre k-mer by appending reference[p - (ref_k-k)/2]
// (main.cpp:395-397).  When ref_k - k is odd that append runs one base ahead of a
// true slide: window w >= 1 reads the centre at offset (ref_k-k) - (ref_k-k)/2,
// and windows 1..k-1 still carry the tail of the first k-mer, i.e. a string with a
// one-base gap.  CentreIn reproduces exactly that string (for even ref_k - k it is
// the plain centred k-mer).
struct CentreIn {
    const u8 *p; // window start
    int off_first, off_slide, keep; // keep = bytes still taken at the first window's offset
    __device__ __forceinline__ u32 operator()(int i) const { return p[(i < keep ? off_first : off_slide) + i]; }
};
// refbad != NULL: only the windows ref_scan_packed_kernel leaves out (`slow_base` = offset of window w0 in the packed arrays)
__global__ void __launch_bounds__(TPB) ref_scan_kernel(const u8 *contig, u64 w0, u64 n_windows, int k, int ref_k, BFView bf,
                                                       BFView ctx, const u32 *__restrict__ refbad, u64 slow_base)
{
    __shared__ u8 sh[TPB + MG_MAX_KMER];
    __shared__ int sh_any;
    const u64 p0 = (u64)blockIdx.x * TPB;
    const u64 avail = n_windows - p0 < TPB ? n_windows - p0 : TPB;
    const int nbytes = (int)avail + ref_k - 1;
    bool mine = threadIdx.x < avail;
    if (refbad) { // (a workgroup with nothing to do does not even stage its bytes)
        if (threadIdx.x == 0) sh_any = 0;
        __syncthreads();
        mine = mine && ref_window_slow_any(refbad, slow_base + p0 + threadIdx.x, w0 + p0 + threadIdx.x, k, ref_k);
        if (mine) sh_any = 1;
        __syncthreads();
        if (!sh_any) return;
    }
    for (int i = threadIdx.x; i < nbytes; i += TPB) sh[i] = contig[p0 + i];
    __syncthreads();
    if (!mine) return;
    const u64 w = w0 + p0 + threadIdx.x; // window index inside the contig
    const int off = (ref_k - k) / 2;
    const int keep = w < (u64)k ? k - (int)w : 0;
    CanonBytes<CentreIn> ck(CentreIn{sh + threadIdx.x, off, (ref_k - k) - off, keep}, k);
    const u64 idx = mod_size(xxh3_bytes(ck, k), bf.mod);
    if (!gate_open(bf, idx) || !bf_bit(bf, idx)) return;
    CanonBytes<LdsIn> cc(LdsIn{sh + threadIdx.x}, ref_k);
    const u64 cidx = mod_size(xxh3_bytes(cc, ref_k), ctx.mod);
    atomicOr((unsigned long long *)&ctx.words[cidx >> 6], 1ULL << (cidx & 63));
}

// ---- the reference in 2-bit form ------------------------------------------------------------------------------------------
// Beside the upper-cased ASCII text the context keeps the reference packed: ref2 holds base i as a 2-bit code (A0 C1 G2 T3) at
// bits 2 (i % 32) of word i / 32 -- the L-form's order, so a span comes out as the L-form of its bases with two loads and a
// shift -- and refbad one bit per base, set where the byte is not one of ACGT (N, IUPAC, a lower-case letter that escaped
// the caller's toupper).  Both arrays are padded with two zero words.  One thread packs 32 bases from eight aligned dwords.
__global__ void __launch_bounds__(TPB) ref_pack_kernel(const u8 *__restrict__ ascii, u64 n, u64 *__restrict__ ref2, u32 *__restrict__ refbad, u64 n_words)
{
    const u64 w = (u64)blockIdx.x * TPB + threadIdx.x;
    if (w >= n_words) return;
    u64 codes = 0;
    u32 bad = 0;
    const u32 *q = (const u32 *)(ascii + 32 * w); // (the buffer is padded to a multiple of 64 bytes)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const u64 at = 32 * w + 4 * (u64)j;
        u32 d = 0u;
        if (at + 4 <= n) d = q[j];
        else // the last, partial dword byte by byte: the buffer may end with its last base (the record loop's allele pool)
            for (u64 i = at; i < n; ++i) d |= (u32)ascii[i] << (8 * (i - at));
        u32 t = (d >> 1) & 0x03030303u; // per byte: A0 C1 G3 T2
        t ^= (t >> 1) & 0x01010101u;    //           A0 C1 G2 T3
        const u32 c8 = (t * 0x01041040u) >> 24;
        const u32 x = expand4(c8) ^ d; // non-zero bytes: not ACGT
        u32 b4 = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (((x >> (8 * i)) & 0xFF) && at + i < n) b4 |= 1u << i;
        codes |= (u64)c8 << (8 * j);
        bad |= b4 << (4 * j);
    }
    ref2[w] = codes;
    refbad[w] = bad;
}
// n <= 32 bases from position `start` as an L-form (base i of the span at bits 2i)
__device__ __forceinline__ u64 ref_codes(const u64 *__restrict__ ref2, u64 start, int n)
{
    const u64 w = start >> 5;
    const int o = (int)(start & 31) * 2;
    u64 v = ref2[w] >> o;
    if (o) v |= ref2[w + 1] << (64 - o);
    return n >= 32 ? v : v & ((1ULL << (2 * n)) - 1);
}
// bit i set: base start + i (i < n <= 32) is outside ACGT
__device__ __forceinline__ u64 ref_badbits(const u32 *__restrict__ refbad, u64 start, int n)
{
    const u64 w = start >> 5;
    const int o = (int)(start & 31);
    const u64 b = ((u64)refbad[w] | ((u64)refbad[w + 1] << 32)) >> o;
    return b & (n >= 32 ? 0xFFFFFFFFULL : ((1ULL << n) - 1));
}
// is any of the n <= 32 bases from `start` outside ACGT?
__device__ __forceinline__ bool ref_bad(const u32 *__restrict__ refbad, u64 start, int n)
{
    const u64 w = start >> 5;
    const int o = (int)(start & 31);
    const u64 b = ((u64)refbad[w] | ((u64)refbad[w + 1] << 32)) >> o;
    return (b & (n >= 32 ? 0xFFFFFFFFULL : ((1ULL << n) - 1))) != 0;
}

// H11 on the packed reference: one thread takes REF_SCAN_W consecutive windows of a contig that lies at `base` in the packed
// arrays (three code words and the bad bits loaded once, the windows slid out of registers): centre k-mer -> canonical -> XXH3
// -> gate -> `bf` bit; on a hit the ref_k-mer -> canonical -> XXH3 -> context_bf bit.  Exactly scan_filter's arithmetic, on
// L-forms cut from the reference instead of M-forms read from a table.  Windows it cannot take -- a base outside ACGT
// inside the window, or the first k windows of a contig when ref_k - k is odd (the reference's sliding quirk, above) -- are
// left to ref_scan_kernel, which is launched over the same range with `only_slow` set and returns at once elsewhere.
constexpr int REF_SCAN_W = 8;
template <int KC, int RC>
__global__ void __launch_bounds__(TPB) ref_scan_packed_kernel(const u64 *__restrict__ ref2, const u32 *__restrict__ refbad, u64 base, u64 first_window, u64 n_windows,
                                                              int k_rt, int r_rt, BFView bf, BFView ctx)
{
    __shared__ u32 sh_lut[256];
    ascii_lut_fill(sh_lut);
    __syncthreads();
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2, off_slide = (r - k) - off;
    const U128 mk = mask128(2 * k), mr = mask128(2 * r);
    const u64 n_threads = (u64)gridDim.x * TPB;
    for (u64 t = (u64)blockIdx.x * TPB + threadIdx.x; t * REF_SCAN_W < n_windows; t += n_threads) {
        const u64 w0 = t * REF_SCAN_W;
        const u64 at0 = base + w0;
        // the 2-bit codes of [at0, at0 + W + r - 1): at most 71 bases behind an offset of up to 31 -> four words
        const u64 wi = at0 >> 5;
        const int o = (int)(at0 & 31) * 2;
        u64 q0 = ref2[wi], q1 = ref2[wi + 1], q2 = ref2[wi + 2], q3 = ref2[wi + 3];
        if (o) {
            q0 = (q0 >> o) | (q1 << (64 - o));
            q1 = (q1 >> o) | (q2 << (64 - o));
            q2 = (q2 >> o) | (q3 << (64 - o));
        }
        // bad bits of the same span (up to 71 + 31 bits: four 32-bit words)
        const u64 b01 = (u64)refbad[wi] | ((u64)refbad[wi + 1] << 32), b23 = (u64)refbad[wi + 2] | ((u64)refbad[wi + 3] << 32);
        const int ob = (int)(at0 & 31);
        const u64 blo = ob ? (b01 >> ob) | (b23 << (64 - ob)) : b01, bhi = b23 >> ob; // bit i: base at0 + i is not ACGT
        for (int j = 0; j < REF_SCAN_W; ++j) {
            if (w0 + j >= n_windows) break;
            const u64 w = first_window + w0 + j; // the window's number inside its contig (`base` is where window first_window starts)
            // any bad base in [j, j + r)?
            const U128 bw = shr128(U128{blo, bhi}, j);
            const u64 bm_lo = r >= 64 ? ~0ULL : ((1ULL << r) - 1);
            bool slow = (bw.lo & bm_lo) != 0;
            if (((r - k) & 1) && w >= 1 && w < (u64)k) slow = true;
            if (slow) continue;
            U128 L = shr128(U128{q0, q1}, 2 * j); // bases [j, j + 64 - j) ...
            if (j) L.hi |= q2 << (64 - 2 * j);   // ... and the rest of the second word
            L.lo &= mr.lo;
            L.hi &= mr.hi;
            const int coff = w == 0 ? off : off_slide;
            U128 f = shr128(L, 2 * coff);
            f.lo &= mk.lo;
            f.hi &= mk.hi;
            const U128 fm = shr128(U128{pairrev64(f.hi), pairrev64(f.lo)}, 2 * (64 - k)); // M-form of the centre k-mer
            const U128 frc{~fm.lo & mk.lo, ~fm.hi & mk.hi};
            const U128 key = lt128(f, frc) ? f : frc;
            const u64 idx = mod_size(xxh3_packed_k<KC>(key, k, sh_lut), bf.mod);
            if (!gate_open(bf, idx) || !bf_bit(bf, idx)) continue;
            const U128 Lm = shr128(U128{pairrev64(L.hi), pairrev64(L.lo)}, 2 * (64 - r));
            const U128 Lrc{~Lm.lo & mr.lo, ~Lm.hi & mr.hi};
            const U128 ckey = lt128(L, Lrc) ? L : Lrc;
            const u64 cidx = mod_size(xxh3_packed_k<RC>(ckey, r, sh_lut), ctx.mod);
            atomicOr((unsigned long long *)&ctx.words[cidx >> 6], 1ULL << (cidx & 63));
        }
    }
}

// ---- H10: KMC scan (main.cpp:482-500) -----------------------------------------
// The scan is three kernels, each one dense in what it does:
//
//   scan_filter_kernel  every table row: canonicalise the centre k-mer, XXH3, slot,
//                       ONE probe of the L2-resident gate.  Rows whose gate is open
//                       (~3-4 %: true matches plus the gate's false positives) are
//                       appended to the "open" list.  This kernel streams the table
//                       and is the dominant one.
//   scan_probe_kernel   open rows only: ref_bf.increment (tag walk in the exact map,
//                       counter add) and the test of the real `bf` bit.  Rows whose
//                       bf bit is set go to the "hit" list.
//   scan_hits_kernel    hit rows only: context_bf.test_key on the ref_k-mer, then
//                       bf.increment's rank + counter add.
//
// Doing the rare work inline instead (first versions) made nearly every wave walk
// the rare path with 2-3 live lanes and eat its HBM latency: 2.0 ms vs 0.8 ms.
// Order of operations vs the reference (main.cpp:495-499): `bf.increment` is a
// no-op unless the bf bit is set, so testing bf before context_bf, and doing the map
// increment in a different kernel, gives identical counters (all adds commute).
//
// List appends are staged per workgroup in LDS and flushed with ONE returning global
// atomic per ~500+ entries: a returning atomic per appending wave on a single counter
// word serialises at ~11 ns each (90 % of the first version's time, and still a third
// of the filter kernel with per-wave staging at a 5 % append rate).
//
// A list entry IS the table row (hi, lo, count): the consumers never go back to the
// table, which would cost two or three random 128-byte lines per entry.
struct RowList {
    u64 *hi, *lo;
    u32 *cnt;
    u64 *aux = nullptr; // the hit list only: what the probe kernel knew of the row's filter entry (counter index | record * 2 + entry << 32)
};
template <int CAP> struct BlockStage {
    u64 *hi, *lo; // [CAP]
    u32 *cnt;     // [CAP]
    u32 *n;       // entries staged
    unsigned long long *base;
    u64 *aux = nullptr; // [CAP], staged beside the rows where the list takes it
    // every lane of the wave must call this (it ballots)
    __device__ __forceinline__ void push(bool take, U128 m, u32 count, u64 extra = 0)
    {
        const u64 mask = __ballot(take);
        if (!mask) return;
        const int lane = threadIdx.x & 63, leader = __ffsll((unsigned long long)mask) - 1;
        u32 off = 0;
        if (lane == leader) off = atomicAdd(n, (u32)__popcll(mask));
        off = __shfl(off, leader, 64);
        if (take) {
            const u32 q = off + __popcll(mask & ((1ULL << lane) - 1));
            lo[q] = m.lo;
            hi[q] = m.hi;
            cnt[q] = count;
            if (aux) aux[q] = extra;
        }
    }
    // every thread of the workgroup must call this; flushes when more than `keep` entries are staged
    __device__ __forceinline__ void flush_if_above(u32 keep, const RowList &g, unsigned long long *g_count)
    {
        __syncthreads();
        const u32 c = *n;
        if (c > keep) {
            if (threadIdx.x == 0) *base = atomicAdd(g_count, (unsigned long long)c);
            __syncthreads();
            const unsigned long long b = *base;
            for (u32 j = threadIdx.x; j < c; j += TPB) {
                if (g.hi) { // (a list of row numbers only has no hi / lo arrays)
                    g.hi[b + j] = hi[j];
                    g.lo[b + j] = lo[j];
                }
                g.cnt[b + j] = cnt[j];
                if (aux && g.aux) g.aux[b + j] = aux[j];
            }
            __syncthreads();
            if (threadIdx.x == 0) *n = 0;
        }
        __syncthreads();
    }
};

// counters[0] = open rows, [1] = hit rows of the current chunk, [2] = hit rows of the whole call
//
// ROWS table rows per thread and tile, in phases so that the memory operations of
// one phase are all in flight together:
//   A  load ROWS x (hi, lo, cnt)            -- coalesced, non-temporal: the only HBM stream
//   B  canonicalise, XXH3, slot             -- pure VALU
//   C  load ROWS gate words                 -- random 8-byte loads from a 4 MiB bitmap (L2): the gate, or the
//                                              coarse gate in front of it when the index is large
//   D  test, stage open rows
// `ablate` is a timing-only diagnostic (results are wrong when it is non-zero):
// 1 = no gate load, 2 = gate load but nothing passes, 4 = no XXH3, 8 = no canonicalisation.
// VAR bit 1 (ROWS == 2 only): each thread takes two ADJACENT rows with 16-byte loads instead of two rows TPB
// apart with 8-byte loads (measured 4 % faster).  Per-wave staging without barriers was tried as VAR bit 0 and
// was 10 % slower (four times the flush atomics).
template <int KC, int RC, int ROWS, int VAR>
__global__ void __launch_bounds__(TPB) scan_filter_kernel(const u64 *__restrict__ hi, const u64 *__restrict__ lo,
                                                          const u32 *__restrict__ cnt, u64 n, int k_rt, int r_rt, BFView bf,
                                                          RowList open, unsigned long long *counters, int ablate)
{
    constexpr bool VEC = (VAR & 2) && ROWS == 2;
    constexpr int CAP = TPB * ROWS + 256;
    __shared__ u64 sh_hi[CAP], sh_lo[CAP];
    __shared__ u32 sh_cnt[CAP];
    __shared__ u32 sh_n;
    __shared__ unsigned long long sh_base;
    BlockStage<CAP> st{sh_hi, sh_lo, sh_cnt, &sh_n, &sh_base};
    __shared__ u32 sh_lut[256];
    ascii_lut_fill(sh_lut);
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    struct Tile {
        U128 m[ROWS];
        u32 count[ROWS]; // not the count: the row's index in this launch's table.  Only ~2 % of the rows survive the gate, so
                         // the counts are fetched for those alone, by the probe kernel (16 streamed bytes per row instead of 20)
        u64 idx[ROWS], gate[ROWS];
        bool valid[ROWS];
    };
    auto load_rows = [&](u64 base, Tile &t) { // A (block-uniform base; nothing is loaded past the table)
        if (VEC && base + (u64)TPB * 2 <= n) { // whole tile inside the table (table bases are 16-byte aligned)
            typedef unsigned long long __attribute__((ext_vector_type(2))) v2u64;
            const u64 i = base + 2 * (u64)threadIdx.x;
            const v2u64 l2 = __builtin_nontemporal_load((const v2u64 *)(lo + i));
            const v2u64 h2 = __builtin_nontemporal_load((const v2u64 *)(hi + i));
            t.m[0] = U128{l2.x, h2.x};
            t.m[ROWS - 1] = U128{l2.y, h2.y};
            t.count[0] = (u32)i;
            t.count[ROWS - 1] = (u32)i + 1;
            t.valid[0] = t.valid[ROWS - 1] = true;
        } else {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const u64 i = VEC ? base + 2 * (u64)threadIdx.x + j : base + (u64)j * TPB + threadIdx.x;
                t.valid[j] = base < n && i < n && (!VEC || i < base + (u64)TPB * 2);
                t.m[j].lo = t.valid[j] ? __builtin_nontemporal_load(lo + i) : 0;
                t.m[j].hi = t.valid[j] ? __builtin_nontemporal_load(hi + i) : 0;
                t.count[j] = (u32)i;
            }
        }
    };
    auto hash_and_probe = [&](Tile &t) { // B, C
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            U128 c = t.m[j];
            if (!(ablate & 8)) c = canon_sub(t.m[j], mform_to_lform(t.m[j], r), r, off, k);
            const u64 h = (ablate & 4) ? (c.lo ^ c.hi) * 0x9E3779B97F4A7C15ULL : xxh3_packed_k<KC>(c, k, sh_lut);
            t.idx[j] = mod_size(h, bf.mod);
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j)
            t.gate[j] = (ablate & 1) ? 0ULL : !bf.use_gate ? ~0ULL : bf.pregate ? bf.pregate[pre_word(bf, t.idx[j])] : bf.gate[gate_word(bf, t.idx[j])];
    };
    auto finish = [&](Tile &t) { // D
        bool open_j[ROWS];
        if (bf.pregate && bf.use_gate) { // coarse gate first; the fine gate's line only for the rows that pass it,
#pragma unroll                           // and all of those loads in flight together
            for (int j = 0; j < ROWS; ++j) {
                const u64 pm = pre_mask(bf, t.idx[j]);
                open_j[j] = t.valid[j] && (t.gate[j] & pm) == pm;
            }
#pragma unroll
            for (int j = 0; j < ROWS; ++j) t.gate[j] = open_j[j] ? bf.gate[gate_word(bf, t.idx[j])] : 0ULL;
        } else {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) open_j[j] = t.valid[j];
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const u64 gm = gate_mask(bf, t.idx[j]);
            const bool take = open_j[j] && !(ablate & 2) && (t.gate[j] & gm) == gm;
            if (ablate) asm volatile("" ::"v"((u32)t.idx[j]), "v"((u32)t.m[j].hi));
            if (ablate & 16) asm volatile("" ::"v"((u32)take)); // 16 = no staging, no barriers (is the per-tile rendezvous what costs?)
            else st.push(take, t.m[j], t.count[j]);
        }
        if (!(ablate & 16)) st.flush_if_above(CAP - TPB * ROWS, open, &counters[0]); // room for one more full iteration
    };
    // (A software pipeline over the tiles -- tile i+1 hashed while tile i waits for its gate words -- was tried and
    // changed nothing: with the gate loads present the kernel takes ~0.75 ms per 1e8 rows whatever the hashing
    // costs; the 1e8 random 8-byte L2 reads are the bound, see DESIGN.md.)
    const u64 step = (u64)gridDim.x * TPB * ROWS;
    for (u64 base = (u64)blockIdx.x * TPB * ROWS; base < n; base += step) {
        Tile t;
        load_rows(base, t);
        hash_and_probe(t);
        finish(t);
    }
    st.flush_if_above(0, open, &counters[0]);
}

// ---- compact table rows ------------------------------------------------------------------------------------------
// The SoA table streams 16 B per row through the filter kernel (plus a random count fetch per open row).  A 43-mer is
// 86 bits and a KMC count at most 255 (-cs255, MALVA:107): a row fits 12 bytes, which is also what a KMC database
// spends (10).  Packed rows: three little-endian dwords per row, the 96-bit value  count << 2 ref_k | k-mer (M-form);
// valid while 2 ref_k + 8 <= 96 and every count < 2^(96 - 2 ref_k)  (mg_kmc_pack_rows checks).  The open rows are listed
// with their counts, so the probe kernel fetches nothing extra; everything after the load is scan_filter_kernel's.
// Measured and dropped on this kernel (round 2, C3, all at 0.70 +- 0.02 ms per 1e8 rows -- the same as the 16-byte SoA
// stream, so the kernel is not bound by its stream): four rows per thread with 16-byte loads (0.74); a variant with the
// filter shape fixed at compile time (60 instead of 106 SGPRs, 8 instead of 6 workgroups per CU: 0.70); the row stream
// prefetched two tiles ahead into rotating register buffers with the gate probes issued first (0.72); no staging and
// no barriers at all (timing only: 0.76).  Without the gate probe the loop takes 0.51 ms: the probe's 0.2 ms is what a
// random L2 word per row costs on top of the hashing, whatever surrounds it.
template <int KC, int RC>
__global__ void __launch_bounds__(TPB) scan_filter12_kernel(const u32 *__restrict__ rows, u64 n, int k_rt, int r_rt, BFView bf, RowList open,
                                                            unsigned long long *counters, int ablate) // ablate: timing / counter calibration only, as in scan_filter_kernel (1, 2)
{
    constexpr int ROWS = 2, CAP = TPB * ROWS + 256; // two adjacent rows per thread: three 8-byte loads
    __shared__ u64 sh_hi[CAP], sh_lo[CAP];
    __shared__ u32 sh_cnt[CAP];
    __shared__ u32 sh_n;
    __shared__ unsigned long long sh_base;
    BlockStage<CAP> st{sh_hi, sh_lo, sh_cnt, &sh_n, &sh_base};
    __shared__ u32 sh_lut[256];
    ascii_lut_fill(sh_lut);
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    const u32 kbits_hi = (u32)(2 * r - 64); // bits of the k-mer in the third dword (2 ref_k > 64 is checked by the host)
    const u32 kmask_hi = (1u << kbits_hi) - 1;
    const u64 n_groups = ((n + 3) / 4 * 4) / ROWS; // the buffer is padded to whole quads (mg_kmc_pack_rows)
    const u64 step = (u64)gridDim.x * TPB;
    typedef unsigned int __attribute__((ext_vector_type(2))) v2u32;
    for (u64 qbase = (u64)blockIdx.x * TPB; qbase < n_groups; qbase += step) {
        const u64 q = qbase + threadIdx.x;
        const bool in = q < n_groups;
        u32 w[3 * ROWS];
#pragma unroll
        for (int i = 0; i < 3 * ROWS; ++i) w[i] = 0;
        if (in) { // A: 24 contiguous bytes, the only HBM stream
            const v2u32 *src = (const v2u32 *)rows + 3 * q;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const v2u32 v = __builtin_nontemporal_load(src + i);
                w[2 * i] = v.x;
                w[2 * i + 1] = v.y;
            }
        }
        U128 m[ROWS];
        u32 count[ROWS];
        u64 idx[ROWS], gate[ROWS];
        bool valid[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            m[j] = U128{w[3 * j] | (u64)w[3 * j + 1] << 32, (u64)(w[3 * j + 2] & kmask_hi)};
            count[j] = w[3 * j + 2] >> kbits_hi;
            valid[j] = in && ROWS * q + j < n;
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { // B
            const U128 cn = canon_sub(m[j], mform_to_lform(m[j], r), r, off, k);
            idx[j] = mod_size(xxh3_packed_k<KC>(cn, k, sh_lut), bf.mod);
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) // C
            gate[j] = (ablate & 1) ? 0ULL : !bf.use_gate ? ~0ULL : bf.pregate ? bf.pregate[pre_word(bf, idx[j])] : bf.gate[gate_word(bf, idx[j])];
        bool open_j[ROWS];
        if (bf.pregate && bf.use_gate) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const u64 pm = pre_mask(bf, idx[j]);
                open_j[j] = valid[j] && (gate[j] & pm) == pm;
            }
#pragma unroll
            for (int j = 0; j < ROWS; ++j) gate[j] = open_j[j] ? bf.gate[gate_word(bf, idx[j])] : 0ULL;
        } else {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) open_j[j] = valid[j];
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { // D
            const u64 gm = gate_mask(bf, idx[j]);
            if (ablate) asm volatile("" ::"v"((u32)idx[j]), "v"((u32)m[j].hi));
            st.push(open_j[j] && !(ablate & 2) && (gate[j] & gm) == gm, m[j], count[j]);
        }
        st.flush_if_above(CAP - TPB * ROWS, open, &counters[0]);
    }
    st.flush_if_above(0, open, &counters[0]);
}

// SoA rows -> packed rows (the producer side of the layout above); *bad is set when a count does not fit
__global__ void __launch_bounds__(TPB) pack_rows12_kernel(const u64 *__restrict__ hi, const u64 *__restrict__ lo, const u32 *__restrict__ cnt, u64 n,
                                                          int ref_k, u32 *__restrict__ out, int *bad)
{
    const u64 i = (u64)blockIdx.x * TPB + threadIdx.x;
    const u64 n_pad = (n + 3) / 4 * 4;
    if (i >= n_pad) return;
    const u32 kbits_hi = (u32)(2 * ref_k - 64);
    u64 l = 0, h = 0;
    u32 c = 0;
    if (i < n) {
        l = lo[i];
        h = hi[i];
        c = cnt[i];
        if ((c >> (32 - kbits_hi)) != 0 || (h >> kbits_hi) != 0) *bad = 1;
    }
    out[3 * i] = (u32)l;
    out[3 * i + 1] = (u32)(l >> 32);
    out[3 * i + 2] = (u32)h | (c << kbits_hi);
}

// ---- partitioned second level -------------------------------------------------------------------------------
// For an index whose gate is far beyond L2 (tens of MiB) the direct form above pays one random HBM line per row
// that passes the coarse gate -- at whole-genome scale that is nearly every second row, and the kernel drops to
// ~0.3 of the roofline.  Here the coarse gate's survivors are instead BINNED by the slice of the fine gate they
// will probe (slices of half the L2-resident size), and a second kernel walks the bins in order, so the slice in
// use stays in L2: the random HBM lines become two extra sequential passes over the survivors only.
constexpr int BIN_MAXP = 32; // bins (fine gate up to 32 slices = 64 MiB)
constexpr int BIN_LDS_ROWS = 2048; // LDS staging rows per workgroup over all bins (40 KB): a ring of 64..256 rows per bin
// Each workgroup of the binning kernel owns one segment of every bin's region, so appending needs no global
// atomic at all (one shared counter per bin was tried first: 2e6 returning atomics on 16 addresses made the
// kernel six times slower than the direct form).
struct BinSet {
    RowList rows;               // [nbins][nseg] segments of `segcap` rows
    u32 *counts;                // [nbins][nseg] rows in each segment
    RowList spill;              // rows that did not fit their segment
    unsigned long long *spill_count;
    u64 segcap;
    u32 nbins, nseg, word_shift; // bin = fine-gate word index >> word_shift, < nbins <= BIN_MAXP; nseg = binning grid
    u32 ring;                    // staging rows per bin: BIN_LDS_ROWS / nbins rounded down to a power of two, at most 256
};

template <int KC, int RC, int ROWS>
__global__ void __launch_bounds__(TPB) scan_bin_kernel(const u64 *__restrict__ hi, const u64 *__restrict__ lo,
                                                       const u32 *__restrict__ cnt, u64 n, int k_rt, int r_rt, BFView bf, BinSet bins,
                                                       int ablate) // timing-only diagnostic: 16 no stores, 32 no staging, 64 no barriers
{
    extern __shared__ u64 sh_rows[]; // nbins * ring rows: hi, lo, cnt (the launch sizes it: 20 B per row)
    __shared__ u32 sh_n[BIN_MAXP], sh_head[BIN_MAXP], sh_pos[BIN_MAXP]; // rows staged so far / flushed so far / rows in the segment
    const int P = (int)bins.nbins;
    const u32 ring = bins.ring, unit = ring / 2;
    u64 *const sh_hi = sh_rows, *const sh_lo = sh_rows + P * ring;
    u32 *const sh_cnt = (u32 *)(sh_rows + 2 * P * ring); // rows leave in units of half a ring: whole cache lines at line-aligned addresses
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __shared__ u32 sh_lut[256];
    ascii_lut_fill(sh_lut);
    if (threadIdx.x < BIN_MAXP) sh_n[threadIdx.x] = sh_head[threadIdx.x] = sh_pos[threadIdx.x] = 0;
    __syncthreads();
    // every thread of the workgroup calls this, between barriers; `all`: also the last partial unit of every bin
    auto flush_bins = [&](bool all) {
        for (int b = wave; b < P; b += TPB / 64) {
            const u32 head = sh_head[b];
            const u32 staged = min(sh_n[b], head + ring); // lanes that found the ring full bumped the counter past it
            const u32 c = all ? staged - head : (staged - head) / unit * unit;
            if (lane == 0) sh_n[b] = staged;
            if (c == 0) continue;
            const u32 pos = sh_pos[b];
            RowList dst = bins.rows;
            unsigned long long at = ((unsigned long long)b * bins.nseg + blockIdx.x) * bins.segcap + pos;
            const bool fits = pos + c <= bins.segcap;
            if (!fits) { // the segment is full: this flush goes to the spill list
                if (lane == 0) at = atomicAdd(bins.spill_count, (unsigned long long)c);
                at = __shfl(at, 0, 64);
                dst = bins.spill;
            }
            for (u32 o = lane; o < c && !(ablate & 16); o += 64) {
                const u32 src = b * ring + ((head + o) & (ring - 1));
                dst.hi[at + o] = sh_hi[src];
                dst.lo[at + o] = sh_lo[src];
                dst.cnt[at + o] = sh_cnt[src];
            }
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                sh_head[b] = head + c;
                if (fits) sh_pos[b] = pos + c;
            }
        }
    };
    const u64 step = (u64)gridDim.x * TPB * ROWS;
    for (u64 base = (u64)blockIdx.x * TPB * ROWS; base < n; base += step) {
        U128 m[ROWS];
        u32 count[ROWS];
        u64 idx[ROWS], gate[ROWS];
        bool pending[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { // A
            const u64 i = base + (u64)j * TPB + threadIdx.x;
            pending[j] = i < n;
            m[j].lo = pending[j] ? __builtin_nontemporal_load(lo + i) : 0;
            m[j].hi = pending[j] ? __builtin_nontemporal_load(hi + i) : 0;
            count[j] = pending[j] ? __builtin_nontemporal_load(cnt + i) : 0;
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { // B
            const U128 c = canon_sub(m[j], mform_to_lform(m[j], r), r, off, k);
            idx[j] = mod_size(xxh3_packed_k<KC>(c, k, sh_lut), bf.mod);
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) gate[j] = bf.pregate[pre_word(bf, idx[j])]; // C: coarse gate, L2-resident
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const u64 pm = pre_mask(bf, idx[j]);
            pending[j] = pending[j] && (gate[j] & pm) == pm;
            if (ablate & 32) {
                asm volatile("" ::"v"((u32)pending[j]));
                pending[j] = false;
            }
        }
        if (ablate & 64) continue;
        // D: survivors into their bin's ring; a full ring defers the lane until the flush that follows
        bool again;
        do {
            bool mine = false;
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                if (pending[j]) {
                    const u32 b = (u32)(gate_word(bf, idx[j]) >> bins.word_shift);
                    const u32 slot = atomicAdd(&sh_n[b], 1u);
                    if (slot - sh_head[b] < ring) {
                        const u32 at = b * ring + (slot & (ring - 1));
                        sh_hi[at] = m[j].hi;
                        sh_lo[at] = m[j].lo;
                        sh_cnt[at] = count[j];
                        pending[j] = false;
                    } else
                        mine = true;
                }
            again = __syncthreads_or(mine);
            flush_bins(false);
            __syncthreads();
        } while (again);
    }
    flush_bins(true);
    __syncthreads();
    if ((int)threadIdx.x < P) bins.counts[threadIdx.x * bins.nseg + blockIdx.x] = sh_pos[threadIdx.x];
}

template <int KC, int RC>
__global__ void __launch_bounds__(TPB) scan_bin_gate_kernel(int k_rt, int r_rt, BFView bf, BinSet bins, RowList open,
                                                            unsigned long long *counters)
{
    constexpr int CAP = 2 * TPB + 256;
    __shared__ u64 sh_hi[CAP], sh_lo[CAP];
    __shared__ u32 sh_cnt[CAP];
    __shared__ u32 sh_n;
    __shared__ unsigned long long sh_base;
    BlockStage<CAP> st{sh_hi, sh_lo, sh_cnt, &sh_n, &sh_base};
    __shared__ u32 sh_lut[256];
    ascii_lut_fill(sh_lut);
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    const int P = (int)bins.nbins;
    auto take_rows = [&](const RowList &src, u64 first, u64 np) { // block-uniform arguments
        for (u64 base = 0; base < np; base += 2 * TPB) {
            bool take[2];
            U128 m[2];
            u32 count[2];
            u64 word[2], idx[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u64 j = base + q * TPB + threadIdx.x;
                take[q] = j < np;
                m[q] = take[q] ? U128{__builtin_nontemporal_load(src.lo + first + j), __builtin_nontemporal_load(src.hi + first + j)} : U128{0, 0};
                count[q] = take[q] ? __builtin_nontemporal_load(src.cnt + first + j) : 0;
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const U128 c = canon_sub(m[q], mform_to_lform(m[q], r), r, off, k);
                idx[q] = mod_size(xxh3_packed_k<KC>(c, k, sh_lut), bf.mod);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) word[q] = bf.gate[gate_word(bf, idx[q])];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const u64 gm = gate_mask(bf, idx[q]);
                st.push(take[q] && (word[q] & gm) == gm, m[q], count[q]);
            }
            st.flush_if_above(CAP - 2 * TPB, open, &counters[0]);
        }
    };
    // Workgroups b and b + 8 share an XCD (round-robin dispatch), and each XCD has its own L2: XCD x takes bins x,
    // x + 8, ... so that its L2 holds one slice of the fine gate for as long as its workgroups need it.  (All
    // workgroups walking all bins in the same order was tried first: they drift apart by several bins within
    // microseconds, and 3 of 4 gate loads missed L2.)  Placement only decides speed, never the result.
    const u32 xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nlocal = gridDim.x >> 3; // the grid is a multiple of 8
    for (int p = (int)xcd; p < P; p += 8)
        for (u32 w = local; w < bins.nseg; w += nlocal)
            take_rows(bins.rows, ((u64)p * bins.nseg + w) * bins.segcap, bins.counts[p * bins.nseg + w]);
    {
        const u64 ns = *bins.spill_count, chunk = (ns + gridDim.x - 1) / gridDim.x;
        const u64 lo_ = min(ns, chunk * blockIdx.x), hi_ = min(ns, lo_ + chunk);
        take_rows(bins.spill, lo_, hi_ - lo_);
    }
    st.flush_if_above(0, open, &counters[0]);
}


// ---- tickets: the partitioned second level for whole-genome indexes ------------------------------------------------
// With 1e8 index entries the fine gate is hundreds of MiB and no L2-sized gate in front of it can reject anything
// (it saturates), so the direct form pays one random HBM line for EVERY table row and runs at a quarter of the
// roofline.  What a row needs from the gate is a function of its slot idx alone, so the first pass does not move
// rows at all: it streams the table (16 B per row), hashes, and files an 8-byte TICKET {idx, row number} under the
// 2 MiB slice of the fine gate that idx falls into.  The second pass gives XCD x the slices x, x + 8, ... : its
// workgroups read the tickets of one slice (sequential, 8 B each) while that slice sits in the XCD's L2, test the
// gate, and fetch from the table only the rows that pass (~1-5 %), which go to the usual open list.
// Traffic per row: 16 (stream) + 8 (ticket out) + 8 (ticket in) = 32 B sequential and one L2 probe, against
// 20 B + one random HBM line (4-5x an L2 read).
constexpr int TK_MAXP = 256;       // slices: fine gates up to 512 MiB
struct TicketSet {
    u64 *tickets;               // [nbins][nseg] segments of `segcap` tickets
    u32 *counts;                // [nbins][nseg]
    u64 *spill;                 // tickets that did not fit their segment
    unsigned long long *spill_count;
    u64 segcap;
    u32 nbins, nseg, word_shift; // bin = fine-gate word index >> word_shift
    u32 row_bits;                // ticket = idx << row_bits | row
    u32 ablate;                  // timing-only diagnostic of pass one (results are wrong when non-zero): 1 = tickets not stored, 2 = tile not sorted either
};

// Pass one: a tile of TK_TILE rows is SORTED by gate slice in LDS (one returning LDS atomic per row gives its rank inside
// its slice, a prefix sum over the slices gives the slice's place in the tile) and leaves as runs of consecutive tickets
// into segments the workgroup owns (no global atomics), every lane of every wave storing.  1.0 ms per 1.3e8 rows: the
// hashing's VALU time, as in the filter kernel.  (First form: an LDS ring per slice, flushed slice by slice by a quarter
// of a wave: a scan over up to 256 slices per 1,024 rows and 16-lane stores, 2.0 ms.)
constexpr int TK_TILE_ROWS = 8;                  // rows per thread and tile
constexpr int TK_TILE = TPB * TK_TILE_ROWS;      // 2,048 rows: 16 tickets per slice on average at 128 slices
// The table is either the SoA arrays (hi, lo) or, when `rows12` is not null, the compact 12-byte rows of scan_filter12_kernel.
template <int KC, int RC>
__global__ void __launch_bounds__(TPB) scan_ticket_sort_kernel(const u64 *__restrict__ hi, const u64 *__restrict__ lo, const u32 *__restrict__ rows12, u64 n,
                                                               int k_rt, int r_rt, BFView bf, TicketSet ts)
{
    // Three barriers per tile: the slice counts alternate between two arrays (the one of tile t + 1 is cleared while tile t
    // is sorted), every wave keeps its own copy of the prefix sums, and a ticket's slice is recomputed from the ticket
    // when it leaves (first form: five barriers and a 4 KB slice-of-position array).
    __shared__ u64 sh_sorted[TK_TILE];
    __shared__ u32 sh_hist[2][TK_MAXP], sh_off[TPB / 64][TK_MAXP + 1], sh_pos[TK_MAXP]; // a tile's counts / places in the tile (per wave) / tickets already in the segment
    __shared__ u32 sh_lut[256];
    const int P = (int)ts.nbins;
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    const int lane = threadIdx.x & 63;
    u32 *const my_off = sh_off[threadIdx.x >> 6];
    ascii_lut_fill(sh_lut);
    for (int b = threadIdx.x; b < P; b += TPB) sh_hist[0][b] = sh_hist[1][b] = sh_pos[b] = 0;
    __syncthreads();
    u32 parity = 0;
    typedef unsigned long long __attribute__((ext_vector_type(2))) v2u64;
    const bool vec_ok = ((((uintptr_t)hi | (uintptr_t)lo) & 15) == 0);
    const u64 step = (u64)gridDim.x * TK_TILE;
    for (u64 base = (u64)blockIdx.x * TK_TILE; base < n; base += step) {
        u64 tk[TK_TILE_ROWS];
        u32 binrank[TK_TILE_ROWS]; // bin << 16 | rank inside the bin (a tile holds 2,048 rows: both fit 16 bits), ~0u = no row
#pragma unroll
        for (int g = 0; g < TK_TILE_ROWS / 2; ++g) { // two adjacent rows per load pair, hashed as they arrive
            const u64 i = base + (u64)g * 2 * TPB + 2 * (u64)threadIdx.x;
            U128 m[2];
            bool live[2];
            if (rows12) { // 24 contiguous bytes: rows i and i + 1 (i is even; the buffer is padded to whole quads of rows)
                typedef unsigned int __attribute__((ext_vector_type(2))) v2u32;
                u32 w[6] = {0, 0, 0, 0, 0, 0};
                if (i < n) {
                    const v2u32 *src = (const v2u32 *)rows12 + 3 * (i / 2);
#pragma unroll
                    for (int q = 0; q < 3; ++q) {
                        const v2u32 v = __builtin_nontemporal_load(src + q);
                        w[2 * q] = v.x;
                        w[2 * q + 1] = v.y;
                    }
                }
                const u32 kmask_hi = (1u << ((2 * r - 64) & 31)) - 1; // bits of the k-mer in the third dword, the count above them (33 <= ref_k <= 44 here)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    m[j] = U128{w[3 * j] | (u64)w[3 * j + 1] << 32, (u64)(w[3 * j + 2] & kmask_hi)};
                    live[j] = i + j < n;
                }
            } else if (vec_ok && i + 1 < n) {
                const v2u64 l2 = __builtin_nontemporal_load((const v2u64 *)(lo + i));
                const v2u64 h2 = __builtin_nontemporal_load((const v2u64 *)(hi + i));
                m[0] = U128{l2.x, h2.x};
                m[1] = U128{l2.y, h2.y};
                live[0] = live[1] = true;
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    live[j] = i + j < n;
                    m[j].lo = live[j] ? __builtin_nontemporal_load(lo + i + j) : 0;
                    m[j].hi = live[j] ? __builtin_nontemporal_load(hi + i + j) : 0;
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const U128 c = canon_sub(m[j], mform_to_lform(m[j], r), r, off, k);
                const u64 idx = mod_size(xxh3_packed_k<KC>(c, k, sh_lut), bf.mod);
                tk[2 * g + j] = (idx << ts.row_bits) | (i + j);
                const u32 bin = (u32)(gate_word(bf, idx) >> ts.word_shift);
                binrank[2 * g + j] = live[j] ? (bin << 16) | atomicAdd(&sh_hist[parity][bin], 1u) : ~0u;
            }
        }
        __syncthreads(); // 1: the tile's slice counts are complete (and the previous tile is done with everything)
        { // every wave: exclusive prefix sum of the slice counts into its own copy (P <= 256: four slices per lane)
            u32 v[4], run = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = lane * 4 + q;
                v[q] = b < P ? sh_hist[parity][b] : 0;
                run += v[q];
            }
            u32 incl = run;
            for (int o = 1; o < 64; o <<= 1) {
                const u32 t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            u32 ex = incl - run;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int b = lane * 4 + q;
                if (b < P) my_off[b] = ex;
                ex += v[q];
            }
            if (lane == 63) my_off[P] = incl; // the tile's ticket count (P may be 256: no lane's b reaches it)
        }
        for (int b = threadIdx.x; b < P; b += TPB) sh_hist[parity ^ 1][b] = 0; // the next tile's counts (last read before barrier 1)
#pragma unroll
        for (int j = 0; j < TK_TILE_ROWS; ++j)
            if (binrank[j] != ~0u && !(ts.ablate & 2)) sh_sorted[my_off[binrank[j] >> 16] + (binrank[j] & 0xFFFF)] = tk[j];
            else if (ts.ablate & 2) asm volatile("" ::"v"((u32)tk[j]), "v"((u32)(tk[j] >> 32)));
        __syncthreads(); // 2: the tile is sorted
        const u32 total = (ts.ablate & 3) ? 0 : my_off[P];
        for (u32 e = threadIdx.x; e < total; e += TPB) { // runs of consecutive tickets, one per slice, into the workgroup's segments
            const u64 t = sh_sorted[e];
            const u32 b = (u32)(gate_word(bf, t >> ts.row_bits) >> ts.word_shift);
            const u32 at = sh_pos[b] + (e - my_off[b]);
            if (at < ts.segcap) ts.tickets[((unsigned long long)b * ts.nseg + blockIdx.x) * ts.segcap + at] = t;
            else ts.spill[atomicAdd(ts.spill_count, 1ULL)] = t; // the segment is full (skewed input): rare
        }
        __syncthreads(); // 3: everybody has read the segments' fills
        for (int b = threadIdx.x; b < P && !(ts.ablate & 3); b += TPB) sh_pos[b] = min(sh_pos[b] + sh_hist[parity][b], (u32)ts.segcap);
        parity ^= 1; // (the fills are next read behind barrier 2 of the next tile; these counts are cleared behind its barrier 1)
    }
    __syncthreads();
    for (int b = threadIdx.x; b < P; b += TPB) ts.counts[b * ts.nseg + blockIdx.x] = sh_pos[b];
}

// Pass two.  XCD x takes the slices x, x + 8, ... (workgroups b and b + 8 share an XCD under round-robin dispatch, and each
// XCD has its own L2); with fewer than 8 slices, slice x mod P, its segments split between the XCDs that share it.  Inside
// a slice a workgroup takes a run of consecutive segments and walks their tickets as ONE dense sequence (a prefix sum of
// the segments' fills in LDS maps a position to its segment): TKG_U tickets per thread and step whatever the segments'
// fills are, the next step's tickets requested before this step's gate words are waited for.  (First form: segment by
// segment, 512 tickets at a time -- a segment holds 512 on average, so every second one took a second, nearly empty step,
// and every step began with a chain count -> tickets -> gate word of dependent loads: 1.3 ms per 1.3e8 tickets, three
// times what the L2 gather costs.)  ONE workgroup per CU, all resident together: a grid larger than what is resident starts
// its late workgroups at the first slice again while the early ones are on the last, two or three slices then fight over
// the 4 MiB L2 and 42 % of the gate loads miss it (TCC_MISS, first form).  The barrier at the end of a walk is what keeps
// a workgroup's waves on the same slice (a flat walk over all slices without it: 15 % slower); a rendezvous of the XCD's
// workgroups between slices (bounded wait) was built and measured too: no gain at this shape, a loss with more
// workgroups.  Survivors are listed by row number alone: the probe kernel, which has no stream to disturb, fetches the
// rows.  Placement only decides speed, never the result.
constexpr int TKG_TPB = 512, TKG_U = 8, TKG_SPW = 64; // threads per workgroup; tickets per thread and step; segments of one slice per walk
constexpr int TKG_WSTAGE = 2048;                       // staged survivors per wave (8 waves: 64 KB)
// GK: the gate's bits per entry fixed at compile time (0 = read from the view)
template <int GK>
__global__ void __launch_bounds__(TKG_TPB) scan_ticket_gate_kernel(BFView bf, TicketSet ts, u32 *__restrict__ open_rows, unsigned long long *counters)
{
    constexpr int STEP = TKG_U * TKG_TPB;
    static_assert(TKG_WSTAGE >= 2 * 64 * TKG_U, "a wave's stage must take a whole step beyond its flush mark");
    __shared__ u32 sh_row[TKG_TPB / 64][TKG_WSTAGE];
    __shared__ u32 sh_start[TKG_SPW + 1];
    const u64 row_mask = (1ULL << ts.row_bits) - 1;
    const u32 gate_k = GK > 0 ? (u32)GK : bf.gate_k;
    const int lane = threadIdx.x & 63;
    u32 *const my_row = sh_row[threadIdx.x >> 6];
    u32 wn = 0; // survivors this wave has staged (the same in every lane).  Staging and flushing are the wave's own business: no
                // barrier inside a walk, so a wave waiting for its gate words never holds up the other fifteen
    auto flush_if_above = [&](u32 keep) { // every lane of the wave
        if (wn <= keep) return;
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(&counters[0], (unsigned long long)wn);
        b = __shfl(b, 0, 64);
        for (u32 j = lane; j < wn; j += 64) open_rows[b + j] = my_row[j];
        wn = 0;
    };
    // The loads of the loop below carry no predicate (a lane without a ticket reads the run's first ticket and the gate's
    // first word): with `if (live) load` the compiler can no longer count outstanding loads and waits for ALL of them before
    // every use, which serialised ticket and gate latency (first form: 0.98 ms per 1.3e8 tickets, 0.41 of it with both loads
    // compiled out and 0.35-0.5 for the gate words alone -- no overlap at all).
    struct Tk {
        u64 t[TKG_U];
        bool live[TKG_U];
    };
    auto gate_loads = [&](const Tk &k, u64 (&word)[TKG_U]) {
#pragma unroll
        for (int u = 0; u < TKG_U; ++u) word[u] = bf.gate[k.live[u] ? gate_word(bf, k.t[u] >> ts.row_bits) : 0]; // the slice in this XCD's L2
    };
    auto test_and_stage = [&](const Tk &k, const u64 (&word)[TKG_U]) { // every lane of the wave
#pragma unroll
        for (int u = 0; u < TKG_U; ++u) {
            const u64 gm = gate_mask_sk(k.t[u] >> ts.row_bits, bf.gate_shift, gate_k);
            const bool take = k.live[u] && (word[u] & gm) == gm;
            const u64 mask = __ballot(take);
            if (take) my_row[wn + __popcll(mask & ((1ULL << lane) - 1))] = (u32)(k.t[u] & row_mask);
            wn += (u32)__popcll(mask);
        }
        flush_if_above(TKG_WSTAGE - 64 * TKG_U);
    };
    // `nsegs` <= TKG_SPW consecutive segments starting at ticket `first`, fills at cnts[0..nsegs): block-uniform arguments
    auto walk = [&](const u64 *first, const u32 *cnts, u32 nsegs) {
        if (threadIdx.x < 64) { // inclusive prefix sum of the fills, one wave
            u32 incl = threadIdx.x < nsegs ? cnts[threadIdx.x] : 0;
            for (int o = 1; o < 64; o <<= 1) {
                const u32 v = __shfl_up(incl, o, 64);
                if ((int)threadIdx.x >= o) incl += v;
            }
            sh_start[threadIdx.x + 1] = incl;
            if (threadIdx.x == 0) sh_start[0] = 0;
        }
        __syncthreads();
        const u32 total = sh_start[nsegs];
        u32 seg[TKG_U], lo_[TKG_U], hi_[TKG_U]; // per position of the step: its segment and that segment's range in the sequence
#pragma unroll
        for (int u = 0; u < TKG_U; ++u) seg[u] = 0, lo_[u] = 0, hi_[u] = sh_start[1];
        auto fetch = [&](u32 base, Tk &k) { // bases must come in increasing order (the cursors only move forward)
#pragma unroll
            for (int u = 0; u < TKG_U; ++u) {
                const u32 q = base + u * TKG_TPB + threadIdx.x;
                k.live[u] = q < total;
                if (k.live[u])
                    while (q >= hi_[u]) { // q < total = sh_start[nsegs]: ends inside the table
                        ++seg[u];
                        lo_[u] = hi_[u];
                        hi_[u] = sh_start[seg[u] + 1];
                    }
                k.t[u] = __builtin_nontemporal_load(first + (k.live[u] ? (u64)seg[u] * ts.segcap + (q - lo_[u]) : 0));
            }
        };
        // tickets are requested two steps ahead: per step the wave issues this step's gate loads, then the tickets of step
        // + 2, and waits for the gate words only (loads return in order, so the tickets of step + 1 are home by then too)
        Tk a, b, c;
        fetch(0, a);
        fetch(STEP, b);
        for (u32 base = 0; base < total; base += STEP) {
            u64 word[TKG_U];
            gate_loads(a, word);
            asm volatile("" ::: "memory");
            fetch(base + 2 * STEP, c);
            asm volatile("" ::: "memory");
            test_and_stage(a, word);
            a = b;
            b = c;
        }
        __syncthreads(); // sh_start is rewritten by the next walk
    };
    const u32 xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nlocal = gridDim.x >> 3; // the grid is a multiple of 8
    const int P = (int)ts.nbins;
    const u32 nshare = P >= 8 ? 1 : (8 - xcd % (u32)P + (u32)P - 1) / (u32)P; // XCDs p, p + P, ... hold slice p when P < 8
    const u32 me = P >= 8 ? local : local * nshare + xcd / (u32)P, nme = nlocal * nshare;
    const u32 spw = (ts.nseg + nme - 1) / nme, s0 = me * spw;
    const u32 s1 = s0 < ts.nseg ? (s0 + spw < ts.nseg ? s0 + spw : ts.nseg) : s0;
    for (int p = P >= 8 ? (int)xcd : (int)(xcd % (u32)P); p < P; p += 8) {
        for (u32 s = s0; s < s1; s += TKG_SPW)
            walk(ts.tickets + ((u64)p * ts.nseg + s) * ts.segcap, ts.counts + (u64)p * ts.nseg + s, s1 - s < (u32)TKG_SPW ? s1 - s : (u32)TKG_SPW);
    }
    { // the spill list: one dense run, an even share per workgroup
        const u64 ns = *ts.spill_count, chunk = (ns + gridDim.x - 1) / gridDim.x;
        const u64 b0 = ns < chunk * blockIdx.x ? ns : chunk * blockIdx.x, b1 = ns < b0 + chunk ? ns : b0 + chunk;
        for (u64 base = b0; base < b1; base += STEP) {
            Tk k;
            u64 word[TKG_U];
#pragma unroll
            for (int u = 0; u < TKG_U; ++u) {
                const u64 q = base + u * TKG_TPB + threadIdx.x;
                k.live[u] = q < b1;
                k.t[u] = k.live[u] ? __builtin_nontemporal_load(ts.spill + q) : 0;
            }
            gate_loads(k, word);
            test_and_stage(k, word);
        }
    }
    flush_if_above(0);
}

// ---- sub-slices: the gate pass answered out of LDS --------------------------------------------------------------------
// The ticket form's second pass reads one random 8-byte word of an L2-resident slice per ticket, and an XCD's L2 serves
// about 14 such reads per clock whatever surrounds them: 0.5 ms per 2^27 tickets for the gather alone, 0.93 ms measured.
// An LDS read costs a twentieth of that.  So this form files the tickets under SUB-SLICES of the gate small enough for one
// workgroup's LDS (2^14 words = 128 KiB; up to 1,024 of them = a 128 MiB gate) and pass two becomes a stream: a workgroup
// copies its sub-slice into LDS once, walks that bin's tickets (8 B each, sequential) and answers every one from LDS.
// Filing under a thousand bins needs long tiles to leave in runs: pass one sorts 16,384 rows at a time (all of a CU's
// LDS, one 1,024-thread workgroup per CU), 16 tickets = one 128-byte line per bin and tile on average.  Survivors are
// listed per (bin, part) region -- one workgroup owns it, so an LDS counter places them and no global atomic is involved
// -- and the probe kernel walks the regions.
// Measured on one GPU's share of C4 (3.75e8 rows, 8e7-SNP index; profiles/r04_c4share_forms.txt), per 2^27 rows: pass one
// 0.97 ms (0.75 of it hashing: without its stores it takes 0.79), pass two 0.36, against 0.88 + 0.95 for the ticket form.
// Built on top and dropped, each measured: 2,048 sub-slices (a 256 MiB gate: a quarter of the false positives, but runs
// of 8 tickets = 64 B: stores and pass two +0.5 ms, the probe kernel -0.23); the next tile's rows requested before the
// sorted tile's stores, so that the hashing would run beside the stores instead of behind them (24-48 more live registers:
// the kernel spills at the 128 a 1,024-thread workgroup may use and its hashing takes 1.0 ms instead of 0.79; the same
// with the tickets waiting in LDS instead of registers: 1.03); two tickets per 16-byte store (no gain: the stores are
// bound by HBM's write rate, 1.07 GB in 0.22 ms, not by their issue).
constexpr int SB_TPB = 1024;                 // threads per workgroup, both passes (one workgroup per CU: LDS)
constexpr int SB_ROWS = 16;                  // rows per thread and tile
constexpr int SB_TILE = SB_TPB * SB_ROWS;    // 16,384 rows: the sorted tile is 128 KiB
constexpr int SB_MAXB = SB_TPB;              // bins: one thread per bin in the prefix sum
constexpr int SB_WORDS_LOG2 = 14;            // largest sub-slice: 2^14 gate words = 128 KiB
constexpr int SBG_U = 10;                    // pass two: ticket loads in flight per lane (640 tickets per wave and step)
struct SubSet {
    u64 *tickets;               // [nbins][nseg] segments of `segcap` tickets
    u32 *counts;                // [nbins][nseg]
    u64 *spill;                 // tickets that did not fit their segment
    unsigned long long *spill_count;
    u64 segcap;
    u32 nbins, nseg;            // nseg = pass one's grid
    u32 bin_shift;              // bin = idx >> bin_shift (gate_shift + 6 + words_log2)
    u32 words_log2;             // gate words per sub-slice (<= SB_WORDS_LOG2)
    u32 row_bits;               // ticket = idx << row_bits | row
    u32 parts;                  // workgroups of pass two that share one bin (its segments are split between them)
    u64 *out_tk;                // [nbins * parts] regions of `ucap` surviving tickets, then the region of the spill list's survivors
    u32 *out_counts;            // [nbins * parts + 1]
    u64 ucap;
    u64 n_gate_words;
    u32 ablate;                 // timing-only diagnostic of pass one (results are wrong when non-zero): 1 = tickets not stored
};

// Pass one.  Four barriers per tile of 16,384 rows: counts complete / wave sums / places known, next counts cleared / tile
// sorted.  The sorted tile does not leave at once: its stores are dealt out over the NEXT tile's hashing, two store
// iterations behind every pair of rows hashed.  One workgroup per CU means nothing else covers its phases, and a wave
// waits for its memory operations in issue order: the rows requested after a block of stores arrive, for the wave, only
// when the last store has been acknowledged -- 0.22 ms per 2^27 rows spent behind 1.07 GB of ticket stores (0.97 ms
// against 0.75 without them).  Spread between the pairs, the stores are younger than the loads the next pair waits for and
// drain beside the arithmetic.  A tile one of whose segments would overflow (skewed input: rare) leaves the old way, at
// once and through the spill list.  R12: the table is compact 12-byte rows (rows12), else the SoA arrays (hi, lo).
template <int KC, int RC, bool R12>
__global__ void __launch_bounds__(SB_TPB) scan_sub_sort_kernel(const u64 *__restrict__ hi, const u64 *__restrict__ lo, const u32 *__restrict__ rows12, u64 n,
                                                                int k_rt, int r_rt, BFView bf, SubSet ss)
{
    __shared__ u64 sh_sorted[SB_TILE];
    __shared__ u32 sh_hist[2][SB_MAXB], sh_off[SB_MAXB], sh_pos[SB_MAXB], sh_wsum[SB_TPB / 64];
    __shared__ u32 sh_lut[256];
    __shared__ u32 sh_slow;
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    ascii_lut_fill(sh_lut);
    sh_hist[0][threadIdx.x] = sh_hist[1][threadIdx.x] = sh_pos[threadIdx.x] = 0; // (SB_TPB == SB_MAXB)
    if (threadIdx.x == 0) sh_slow = 0;
    __syncthreads();
    u32 parity = 0;
    const u32 segcap = (u32)ss.segcap;
    u32 prev_total = 0, prev_mine = 0; // the tile still waiting in sh_sorted: its tickets, and those of bin threadIdx.x among them
    auto store_one = [&](u32 e, u32 total_) { // ticket e of the sorted tile into its bin's segment (the tile fits its segments: checked before it was left waiting)
        if (e < total_) {
            const u64 t = sh_sorted[e];
            const u32 b = (u32)((t >> ss.row_bits) >> ss.bin_shift);
            ss.tickets[((unsigned long long)b * ss.nseg + blockIdx.x) * ss.segcap + sh_pos[b] + (e - sh_off[b])] = t;
        }
    };
    for (u64 base = (u64)blockIdx.x * SB_TILE; base < n; base += (u64)gridDim.x * SB_TILE) {
        u64 tk[SB_ROWS];
        u32 binrank[SB_ROWS]; // bin << 16 | rank inside the bin (a tile holds 16,384 rows), ~0u = no row
        // Two adjacent rows (a "pair") per load group, hashed as they arrive -- with the loads of the pairs AHEAD already
        // requested: left to itself the compiler sinks every pair's loads to their use (fewest live registers), and a wave
        // then sits out one whole memory latency per pair, eight per tile, with three other waves on its SIMD to cover it.
        constexpr int RAW = R12 ? 6 : 8, AHEAD = 2;
        u32 raw[SB_ROWS / 2][RAW];
        auto request = [&](int g) { // (no predicate on a load: a pair beyond the table reads the last pair again and is dropped by its row number)
            u32 (&w)[RAW] = raw[g];
            u64 i = base + (u64)g * 2 * SB_TPB + 2 * (u64)threadIdx.x; // rows i and i + 1 (i is even)
            const u64 last = (n - 1) & ~1ULL;
            i = i < last ? i : last;
            if constexpr (R12) { // 24 contiguous bytes (the buffer is padded to whole quads of rows)
                typedef unsigned int __attribute__((ext_vector_type(2))) v2u32;
                const v2u32 *src = (const v2u32 *)rows12 + 3 * (i / 2);
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const v2u32 v = __builtin_nontemporal_load(src + q);
                    w[2 * q] = v.x;
                    w[2 * q + 1] = v.y;
                }
            } else {
                const u64 i1 = i + 1 < n ? i + 1 : i;
                const u64 l0 = __builtin_nontemporal_load(lo + i), h0 = __builtin_nontemporal_load(hi + i);
                const u64 l1 = __builtin_nontemporal_load(lo + i1), h1 = __builtin_nontemporal_load(hi + i1);
                w[0] = (u32)l0, w[1] = (u32)(l0 >> 32), w[2] = (u32)h0, w[3] = (u32)(h0 >> 32);
                w[4] = (u32)l1, w[5] = (u32)(l1 >> 32), w[6] = (u32)h1, w[7] = (u32)(h1 >> 32);
            }
        };
#pragma unroll
        for (int g = 0; g < AHEAD; ++g) request(g);
        const u32 kmask_hi = (1u << ((2 * r - 64) & 31)) - 1; // compact rows: bits of the k-mer in the third dword, the count above them
#pragma unroll
        for (int g = 0; g < SB_ROWS / 2; ++g) {
            if (g + AHEAD < SB_ROWS / 2) request(g + AHEAD);
            asm volatile("" ::: "memory"); // (the requests stay in front of this pair's arithmetic)
            const u64 i = base + (u64)g * 2 * SB_TPB + 2 * (u64)threadIdx.x;
            const u32 (&w)[RAW] = raw[g];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                U128 m;
                if constexpr (R12) m = U128{w[3 * j] | (u64)w[3 * j + 1] << 32, (u64)(w[3 * j + 2] & kmask_hi)};
                else m = U128{w[4 * j] | (u64)w[4 * j + 1] << 32, w[4 * j + 2] | (u64)w[4 * j + 3] << 32};
                const U128 c = canon_sub(m, mform_to_lform(m, r), r, off, k);
                const u64 idx = mod_size(xxh3_packed_k<KC>(c, k, sh_lut), bf.mod);
                tk[2 * g + j] = (idx << ss.row_bits) | (i + j);
                const u32 bin = (u32)(idx >> ss.bin_shift);
                binrank[2 * g + j] = i + j < n ? (bin << 16) | atomicAdd(&sh_hist[parity][bin], 1u) : ~0u;
            }
            if (!(ss.ablate & 1)) { // two store iterations of the tile before (SB_ROWS of them in all: sixteen per thread)
                store_one((u32)(2 * g) * SB_TPB + threadIdx.x, prev_total);
                store_one((u32)(2 * g + 1) * SB_TPB + threadIdx.x, prev_total);
            }
        }
        __syncthreads(); // 1: the tile's counts are complete, and the previous tile has left the sorted array
        if (!(ss.ablate & 1)) sh_pos[threadIdx.x] = min(sh_pos[threadIdx.x] + prev_mine, segcap); // (its segments' fills: nobody reads them before barrier 3)
        const u32 mine = sh_hist[parity][threadIdx.x]; // this tile's tickets of bin threadIdx.x (bins beyond nbins stay empty)
        u32 incl = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const u32 t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        if (lane == 63) sh_wsum[wave] = incl;
        __syncthreads(); // 2
        u32 before = 0, total = 0;
#pragma unroll
        for (int w = 0; w < SB_TPB / 64; ++w) {
            const u32 s = sh_wsum[w];
            before += w < wave ? s : 0u;
            total += s;
        }
        sh_off[threadIdx.x] = before + incl - mine;
        sh_hist[parity ^ 1][threadIdx.x] = 0; // the next tile's counts
        if (sh_pos[threadIdx.x] + mine > segcap) sh_slow = 1; // a segment would overflow: this tile goes at once, through the spill list
        __syncthreads(); // 3: every bin's place in the tile is known
#pragma unroll
        for (int j = 0; j < SB_ROWS; ++j)
            if (binrank[j] != ~0u) sh_sorted[sh_off[binrank[j] >> 16] + (binrank[j] & 0xFFFFu)] = tk[j];
        __syncthreads(); // 4: the tile is sorted
        prev_total = total;
        prev_mine = mine;
        if (sh_slow) { // (block-uniform) runs of consecutive tickets, one per bin, into the workgroup's segments; what does not fit, to the spill list
            if (!(ss.ablate & 1))
                for (u32 e = threadIdx.x; e < total; e += SB_TPB) {
                    const u64 t = sh_sorted[e];
                    const u32 b = (u32)((t >> ss.row_bits) >> ss.bin_shift);
                    const u32 at = sh_pos[b] + (e - sh_off[b]);
                    if (at < segcap) ss.tickets[((unsigned long long)b * ss.nseg + blockIdx.x) * ss.segcap + at] = t;
                    else ss.spill[atomicAdd(ss.spill_count, 1ULL)] = t;
                }
            __syncthreads(); // everybody has read the segments' fills
            if (!(ss.ablate & 1)) sh_pos[threadIdx.x] = min(sh_pos[threadIdx.x] + mine, segcap);
            if (threadIdx.x == 0) sh_slow = 0;
            prev_total = prev_mine = 0;
            // (the next write of sh_slow is behind the next tile's barriers 1 and 2)
        }
        parity ^= 1;
    }
    if (!(ss.ablate & 1)) // the last tile
        for (u32 e = threadIdx.x; e < prev_total; e += SB_TPB) store_one(e, prev_total);
    __syncthreads();
    if (!(ss.ablate & 1)) sh_pos[threadIdx.x] = min(sh_pos[threadIdx.x] + prev_mine, segcap);
    __syncthreads();
    if (threadIdx.x < ss.nbins) ss.counts[threadIdx.x * ss.nseg + blockIdx.x] = sh_pos[threadIdx.x];
}

// Pass two.  Unit u = (bin, part): the workgroup copies the bin's sub-slice of the gate into LDS, then its waves take the
// part's segments one by one -- a segment's tickets requested together, ten loads per lane -- and answer every ticket
// from LDS.  Survivors go to the unit's own region as row numbers, placed by a counter in LDS.  The spill list (tickets of
// any bin) is answered from the gate in global memory at the end.  GK: the gate's bits per entry at compile time (0 = view).
template <int GK>
__global__ void __launch_bounds__(SB_TPB) scan_sub_gate_kernel(BFView bf, SubSet ss)
{
    __shared__ u64 sh_gate[1 << SB_WORDS_LOG2];
    __shared__ u32 sh_out;
    const u32 gate_k = GK > 0 ? (u32)GK : bf.gate_k;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u32 wmask = (1u << ss.words_log2) - 1;
    const u32 units = ss.nbins * ss.parts;
    const u32 spp = (ss.nseg + ss.parts - 1) / ss.parts; // segments per part
    for (u32 unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const u32 bin = unit / ss.parts, part = unit % ss.parts;
        const u32 s0 = part * spp, s1 = min(ss.nseg, s0 + spp);
        const u64 w0 = (u64)bin << ss.words_log2;
        const u32 nw = (u32)min((u64)1 << ss.words_log2, ss.n_gate_words > w0 ? ss.n_gate_words - w0 : (u64)0);
        for (u32 i = threadIdx.x; i < nw; i += SB_TPB) sh_gate[i] = bf.gate[w0 + i];
        if (threadIdx.x == 0) sh_out = 0;
        __syncthreads();
        u64 *const out = ss.out_tk + (u64)unit * ss.ucap;
        // The wave's segments are s0 + wave, + 16, ...: their fills are fetched together up front (lane i holds the i-th), and a
        // segment's tickets are requested while the segment before it is answered -- a wave that asked for a fill, waited, asked
        // for the tickets, waited, and only then worked spent most of its time on the two round trips (0.36 ms per 2^27 tickets
        // against 0.24 for the bytes).
        constexpr int NW = SB_TPB / 64;
        const u32 n_mine = s1 > s0 + wave ? (s1 - s0 - wave + NW - 1) / NW : 0; // (<= 64 while nseg <= 1,024: the layout's bound)
        const u32 my_fill = (u32)lane < n_mine ? ss.counts[bin * ss.nseg + s0 + wave + lane * NW] : 0u;
        struct Tk {
            u64 t[SBG_U];
            u32 cnt;
        };
        auto request = [&](u32 i, u32 q0, Tk &k) { // tickets q0 ... of the wave's i-th segment (no predicate on the loads themselves)
            const u32 sg = s0 + wave + i * NW;
            k.cnt = (u32)__shfl((int)my_fill, (int)(i & 63), 64);
            const u64 *seg = ss.tickets + ((u64)bin * ss.nseg + sg) * ss.segcap;
#pragma unroll
            for (int u = 0; u < SBG_U; ++u) {
                const u32 q = q0 + u * 64 + lane;
                k.t[u] = __builtin_nontemporal_load(seg + (q < k.cnt ? q : 0u));
            }
        };
        auto answer = [&](const Tk &k, u32 q0) {
#pragma unroll
            for (int u = 0; u < SBG_U; ++u) {
                const u64 idx = k.t[u] >> ss.row_bits;
                const u64 word = sh_gate[(u32)(idx >> (bf.gate_shift + 6)) & wmask];
                const u64 gm = gate_mask_sk(idx, bf.gate_shift, gate_k);
                const bool take = q0 + u * 64 + lane < k.cnt && (word & gm) == gm;
                const u64 mask = __ballot(take);
                if (mask) {
                    const int leader = __ffsll((unsigned long long)mask) - 1;
                    u32 b = 0;
                    if (lane == leader) b = atomicAdd(&sh_out, (u32)__popcll(mask));
                    b = __shfl(b, leader, 64);
                    if (take) out[b + __popcll(mask & ((1ULL << lane) - 1))] = k.t[u];
                }
            }
        };
        auto rest = [&](u32 i, const Tk &k) { // a segment fuller than one step holds (1.25 times the mean): rare
            for (u32 q0 = 64 * SBG_U; q0 < k.cnt; q0 += 64 * SBG_U) {
                Tk more;
                request(i, q0, more);
                answer(more, q0);
            }
        };
        Tk a, b; // (two buffers taken in turn: copying one into the other would wait for its loads)
        if (n_mine) request(0, 0, a);
        for (u32 i = 0; i < n_mine; i += 2) {
            if (i + 1 < n_mine) request(i + 1, 0, b);
            asm volatile("" ::: "memory"); // (the next segment's requests stay in front of this one's answers)
            answer(a, 0);
            rest(i, a);
            if (i + 1 >= n_mine) break;
            if (i + 2 < n_mine) request(i + 2, 0, a);
            asm volatile("" ::: "memory");
            answer(b, 0);
            rest(i + 1, b);
        }
        __syncthreads();
        if (threadIdx.x == 0) ss.out_counts[unit] = sh_out;
        __syncthreads(); // (the next unit rewrites the sub-slice and the counter)
    }
    { // the spill list: one dense run, an even share per workgroup; any bin, so the gate in global memory answers
        const u64 ns = *ss.spill_count, chunk = (ns + gridDim.x - 1) / gridDim.x;
        const u64 b0 = ns < chunk * blockIdx.x ? ns : chunk * blockIdx.x, b1 = ns < b0 + chunk ? ns : b0 + chunk;
        u64 *const out = ss.out_tk + (u64)units * ss.ucap;
        for (u64 base = b0; base < b1; base += SB_TPB) {
            const u64 q = base + threadIdx.x;
            const bool live = q < b1;
            const u64 t = live ? ss.spill[q] : 0;
            const u64 idx = t >> ss.row_bits;
            const u64 word = live ? bf.gate[gate_word(bf, idx)] : 0;
            const u64 gm = gate_mask_sk(idx, bf.gate_shift, gate_k);
            const bool take = live && (word & gm) == gm;
            const u64 mask = __ballot(take);
            if (mask) {
                const int leader = __ffsll((unsigned long long)mask) - 1;
                u32 b = 0;
                if (lane == leader) b = atomicAdd(&ss.out_counts[units], (u32)__popcll(mask));
                b = __shfl(b, leader, 64);
                if (take) out[b + __popcll(mask & ((1ULL << lane) - 1))] = t;
            }
        }
    }
}
// the open rows of a chunk, for mg_scan_stats: the sum of the units' counts
__global__ void __launch_bounds__(SB_TPB) sub_total_kernel(const u32 *__restrict__ counts, u32 n, unsigned long long *counters)
{
    __shared__ unsigned long long sh[SB_TPB / 64];
    unsigned long long v = 0;
    for (u32 i = threadIdx.x; i < n; i += SB_TPB) v += counts[i];
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long t = 0;
        for (int w = 0; w < SB_TPB / 64; ++w) t += sh[w];
        counters[0] = t;
    }
}

// (Probe and hit pass fused in one kernel -- no second list, no re-hash -- was measured: 0.276 ms against 0.100 +
// 0.050 ms; a third of the lanes running a second XXH3 while the others idle costs more than the list.)
template <int KC, int RC>
__global__ void __launch_bounds__(TPB) scan_probe_kernel(int k_rt, int r_rt, BFView bf, MapView map, RowList open, RowList hits,
                                                         unsigned long long *counters, const u32 *__restrict__ cnt_table,
                                                         const u64 *__restrict__ row_hi, const u64 *__restrict__ row_lo, const u32 *__restrict__ rows12)
{
    constexpr int CAP = TPB + 256;
    __shared__ u64 sh_hi[CAP], sh_lo[CAP];
    __shared__ u32 sh_cnt[CAP];
    __shared__ u32 sh_n;
    __shared__ unsigned long long sh_base;
    __shared__ u64 sh_aux[CAP];
    BlockStage<CAP> st{sh_hi, sh_lo, sh_cnt, &sh_n, &sh_base, hits.aux ? sh_aux : (u64 *)nullptr};
    __shared__ u32 sh_lut[256];
    ascii_lut_fill(sh_lut);
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    // one tile of TPB open rows: `live` = this thread has one, `count` = what the list holds for it (the row's count, or
    // its number in the table), j = its place in the flat open list (the forms whose list carries the rows).  Every thread
    // of the workgroup calls it (barriers inside).
    auto tile = [&](bool live, u32 count, u64 j) {
        bool hit = false;
        U128 m{0, 0};
        u64 extra = 0;
        if (live) {
            if (rows12) { // ticket form over compact rows: the list holds row numbers, a row is 12 contiguous bytes (one line, now and then two)
                typedef u32 __attribute__((ext_vector_type(3), aligned(4))) row12_t; // ONE load instruction (global_load_dwordx3): one translation per row
                const row12_t wv = __builtin_nontemporal_load((const row12_t *)(rows12 + 3 * (u64)count));
                const u32 w0 = wv.x, w1 = wv.y, w2 = wv.z;
                const u32 kbits_hi = (u32)(2 * r - 64) & 31; // (33 <= ref_k <= 44 here)
                m = U128{w0 | (u64)w1 << 32, (u64)(w2 & ((1u << kbits_hi) - 1))};
                count = w2 >> kbits_hi;
            } else if (row_hi) { // ticket form: the list holds row numbers only; the row's three words are requested together
                m = U128{__builtin_nontemporal_load(row_lo + count), __builtin_nontemporal_load(row_hi + count)};
                count = __builtin_nontemporal_load(cnt_table + count);
            } else {
                m = U128{open.lo[j], open.hi[j]};
                if (cnt_table) count = __builtin_nontemporal_load(cnt_table + count); // the filter kernel listed the row's index (requested beside the record below)
            }
        }
        {
            const U128 c = canon_sub(m, mform_to_lform(m, r), r, off, k);
            const u64 h = xxh3_packed_k<KC>(c, k, sh_lut);
            const u64 idx = mod_size(h, bf.mod);
            long long id, rank; // one record answers both: exact-map key?  bit idx of bf set?
            u64 slot = 0, ent = 0;
            bucket_probe_coop(map, c, h, idx, live, &id, &rank, &slot, &ent); // (whole waves: the records are fetched four lanes to a record)
            if (id >= 0) {
                atomicAdd(&map.vals[id], count); // ref_bf.increment (main.cpp:495)
                if (map.epoch) rec_add_val(&map.slots[slot], map.epoch, count); // ... and the record's copy, on the line just read
            }
            hit = rank >= 0;
            extra = (u64)(u32)rank | ent << 32; // (the hit kernel need not find the entry again)
        }
        st.push(hit, m, count, extra);
        st.flush_if_above(CAP - TPB, hits, &counters[1]);
    };
    const u64 n_open = counters[0];
    const u64 step = (u64)gridDim.x * TPB;
    for (u64 base = (u64)blockIdx.x * TPB; base < n_open; base += step) {
        const u64 j = base + threadIdx.x;
        tile(j < n_open, j < n_open ? open.cnt[j] : 0u, j);
    }
    st.flush_if_above(0, hits, &counters[1]);
}

// The sub-slice form's probe AND hit pass.  Its open list holds TICKETS (filter slot | row number), region by region, so
// the record -- which is addressed by the slot alone -- is fetched FIRST and whole (64 bytes: key, the filter's two
// entries, the record's copies of its counters), and the table row only where the record says it can matter: it holds a
// key (which may be this row's k-mer), or the slot is a set bit of `bf`.  A row that got here by a false positive of the
// gate finds an empty record five times in six and ends after ONE random line instead of two; the slot came with the
// ticket and a key is compared as it is, so the centre k-mer is never hashed again.  A row whose slot IS a set bit goes on
// at once -- ref_k-mer hashed, context filter asked, counter and its copy in the record added to -- on the record's line
// while it is still near, with the copy's old word already in hand for the compare-and-swap: no hit list, no second fetch
// of row or record.  Where the records' copies are the counters of record (MapView::lazy: one GPU, a large index) the adds to
// vals[] / counts[] -- a random line each, of vectors nobody reads before the record loop has read the copies -- are left out.  (Earlier forms fused the two passes and lost, DESIGN_NOTES: every lane then hashed the centre k-mer
// too and the vector pipe was the bound; here the kernel waits on memory and the ref_k-mer hash of one lane in five runs
// in its shadow.)  Nothing synchronises: workgroup w takes every `split`-th group of four tiles of region w / split, each
// of its waves one tile of the four; a lane that has to walk on, or retry a compare-and-swap, holds up its own wave only.
struct SubOpen {
    const u32 *counts = nullptr; // tickets in each region
    const u64 *tickets = nullptr;
    u64 ucap = 0;                // region u starts at tickets + u * ucap
    u32 n_units = 0, split = 1, row_bits = 27;
};
constexpr int SUBP_WAVES = TPB / 64;
template <int KC, int RC>
__global__ void __launch_bounds__(TPB) scan_sub_probe_kernel(int k_rt, int r_rt, BFView bf, BFView ctx, MapView map, SubOpen sub, unsigned long long *counters,
                                                             const u32 *__restrict__ cnt_table, const u64 *__restrict__ row_hi, const u64 *__restrict__ row_lo,
                                                             const u32 *__restrict__ rows12)
{
    __shared__ u32 sh_hits;
    if (threadIdx.x == 0) sh_hits = 0;
    __syncthreads();
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const u64 row_mask = (1ULL << sub.row_bits) - 1, smask = (1ULL << map.cap_log2) - 1;
    U128 m{0, 0}, key{0, 0};
    u32 count = 0, n_hits = 0;
    auto fetch_row = [&](u32 row) { // the table row (ref_k-mer, count) and its canonical centre k-mer
        if (rows12) { // 12 contiguous bytes: ONE load instruction (global_load_dwordx3)
            typedef u32 __attribute__((ext_vector_type(3), aligned(4))) row12_t;
            const row12_t wv = __builtin_nontemporal_load((const row12_t *)(rows12 + 3 * (u64)row));
            const u32 kbits_hi = (u32)(2 * r - 64) & 31; // (33 <= ref_k <= 44 here)
            m = U128{wv.x | (u64)wv.y << 32, (u64)(wv.z & ((1u << kbits_hi) - 1))};
            count = wv.z >> kbits_hi;
        } else {
            m = U128{__builtin_nontemporal_load(row_lo + row), __builtin_nontemporal_load(row_hi + row)};
            count = __builtin_nontemporal_load(cnt_table + row);
        }
        key = canon_sub(m, mform_to_lform(m, r), r, off, k);
    };
    for (u32 w = blockIdx.x; w < sub.n_units * sub.split; w += gridDim.x) {
        const u32 unit = w / sub.split, sl = w % sub.split;
        const u32 cnt = sub.counts[unit];
        const u64 *tks = sub.tickets + (u64)unit * sub.ucap;
        for (u32 base = (sl * SUBP_WAVES + wave) * 64; base < cnt; base += sub.split * TPB) { // (wave-uniform bounds)
            const u32 j = base + lane;
            const bool live = j < cnt;
            const u64 t = live ? __builtin_nontemporal_load(tks + j) : 0;
            const u64 idx = t >> sub.row_bits, want = idx + 1;
            u64 s = map_home(map, idx);
            uint4 a, b, c, d;
            records_load_coop4(map, s, live, &a, &b, &c, &d); // (whole waves: the records are fetched four lanes to a record)
            bool map_open = live && a.x != 0, bf_open = live, have_row = false;
            for (;;) {
                if (bf_open) { // bit idx of bf: an entry of the record, or of one further on while both entries are taken by other bits
                    const u64 b0 = c.x | (u64)c.y << 32, b1 = c.z | (u64)c.w << 32;
                    const int e = b0 == want ? 0 : b1 == want ? 1 : -1;
                    if (e >= 0) {
                        bf_open = false;
                        ++n_hits;
                        if (!have_row) fetch_row((u32)(t & row_mask)), have_row = true;
                        const U128 cc = canon_sub(m, mform_to_lform(m, r), r, 0, r);
                        const u64 cidx = mod_size(xxh3_packed_k<RC>(cc, r), ctx.mod);
                        if (!bf_bit_via_set(ctx, cidx)) {                     // context_bf.test_key (main.cpp:496)
                            if (!map.lazy) atomicAdd(&bf.counts[e ? b.w : b.z], count); // bf.increment (main.cpp:498)
                            if (map.epoch) rec_add_bf_from(&map.slots[s], e, map.epoch, count, d.z | (unsigned long long)d.w << 32); // ... and its copy in the record
                        }
                    } else if (b0 == 0 || b1 == 0)
                        bf_open = false;
                }
                if (map_open) { // a key lives here: the row's k-mer decides whether it is this one
                    if (!have_row) fetch_row((u32)(t & row_mask)), have_row = true;
                    if (a.x == 0) map_open = false;
                    else if (a.z == (u32)key.lo && a.w == (u32)(key.lo >> 32) && b.x == (u32)key.hi && b.y == (u32)(key.hi >> 32)) {
                        map_open = false;
                        if (!map.lazy) atomicAdd(&map.vals[a.y], count); // ref_bf.increment (main.cpp:495)
                        if (map.epoch) rec_add_val_from(&map.slots[s], map.epoch, count, d.x | (unsigned long long)d.y << 32); // ... and the record's copy, on the line just read
                    }
                }
                if (!map_open && !bf_open) break;
                s = (s + 1) & smask; // (one lane in five goes on to the next record: nearly always the same page)
                const uint4 *p = reinterpret_cast<const uint4 *>(&map.slots[s]);
                a = p[0];
                b = p[1];
                c = p[2];
                d = p[3];
            }
        }
    }
    // mg_scan_stats: the rows whose slot was a set bit of bf (counters[1]: this launch group, [2]: the whole call)
    for (int o = 32; o > 0; o >>= 1) n_hits += __shfl_down(n_hits, o, 64);
    if (lane == 0 && n_hits) atomicAdd(&sh_hits, n_hits);
    __syncthreads();
    if (threadIdx.x == 0 && sh_hits) {
        atomicAdd(&counters[1], (unsigned long long)sh_hits);
        atomicAdd(&counters[2], (unsigned long long)sh_hits);
    }
}

template <int KC, int RC>
__global__ void __launch_bounds__(TPB) scan_hits_kernel(int k_rt, int r_rt, BFView bf, BFView ctx, MapView map, RowList hits,
                                                        unsigned long long *counters)
{
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    const u64 nh = counters[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[2] += nh;
    for (u64 base = (u64)blockIdx.x * TPB; base < nh; base += (u64)gridDim.x * TPB) { // (whole waves stay together: the records are fetched cooperatively)
        const u64 j = base + threadIdx.x;
        const bool live = j < nh;
        const U128 m{live ? hits.lo[j] : 0ULL, live ? hits.hi[j] : 0ULL};
        const U128 l = mform_to_lform(m, r);
        const U128 cc = canon_sub(m, l, r, 0, r);
        const u64 cidx = mod_size(xxh3_packed_k<RC>(cc, r), ctx.mod);
        // (the centre k-mer's slot and its record are computed and requested before the context bit is looked at:
        // both random reads are in flight together)
        if (hits.aux) { // the probe kernel left the entry with the row: no second hash, no second walk to the record
            const bool in_ctx = live && bf_bit_via_set(ctx, cidx);        // context_bf.test_key (main.cpp:496)
            if (live && !in_ctx) {
                const u64 x = hits.aux[j];
                atomicAdd(&bf.counts[(u32)x], hits.cnt[j]);               // bf.increment (main.cpp:498)
                if (map.epoch) rec_add_bf(&map.slots[(x >> 32) >> 1], (int)((x >> 32) & 1), map.epoch, hits.cnt[j]);
            }
            continue;
        }
        const u64 idx = mod_size(xxh3_packed_k<KC>(canon_sub(m, l, r, off, k), k), bf.mod);
        const bool in_ctx = live && bf_bit_via_set(ctx, cidx);            // context_bf.test_key (main.cpp:496)
        u64 ent = 0;
        const long long rank = bucket_rank_coop(map, idx, live, &ent);    // set for every row of this list
        if (!in_ctx && rank >= 0) {
            atomicAdd(&bf.counts[rank], hits.cnt[j]); // bf.increment (main.cpp:498)
            if (map.epoch) rec_add_bf(&map.slots[ent >> 1], (int)(ent & 1), map.epoch, hits.cnt[j]);
        }
    }
}

// ---- KMC database records -> table rows -------------------------------------------------------------------
// A KMC (>= 2, "0x200") database lists its k-mers as fixed-size records in <db>.kmc_suf: the SUFFIX of the k-mer
// (k - p symbols, 2 bits each, first symbol in the top bits of the first byte) followed by the counter
// (little-endian); the PREFIX (p symbols) is not stored per record: <db>.kmc_pre holds, for every bin and every
// prefix value, the index of the first record that carries it (CKMCFile::ReadNextKmer walks that table while it
// lists; main.cpp:482-490 is its caller).  This kernel turns a run of raw records into the scan's SoA rows on the
// device, so the host moves 10 bytes per 43-mer over PCIe instead of 20 and parses nothing.
//   lut[j], j < n_lut   first record of prefix (j mod 4^p) in bin j / 4^p;  lut[n_lut] = a value above every record
// A workgroup decodes KMC_TILE consecutive records: their bytes are staged through LDS with coalesced dword loads,
// one lane finds the table entry of the tile's first record (binary search over the whole table, once per tile),
// the next KMC_WIN entries go to LDS and every record finds its own entry there (a tile normally spans a handful);
// records beyond the window (runs of empty prefixes) search the global table.
constexpr int KMC_TILE = 1024, KMC_WIN = 512, KMC_MAX_REC = 20;
__device__ __forceinline__ u64 kmc_upper_bound(const u64 *a, u64 lo, u64 hi, u64 x) // first index in [lo, hi) with a[i] > x, else hi
{
    while (lo < hi) {
        const u64 mid = lo + (hi - lo) / 2;
        if (a[mid] <= x) lo = mid + 1;
        else hi = mid;
    }
    return lo;
}
__global__ void __launch_bounds__(TPB) kmc_decode_kernel(const u8 *__restrict__ rec, u64 n, u64 first_record, u32 suffix_bytes,
                                                         u32 counter_bytes, u32 prefix_len, const u64 *__restrict__ lut, u64 n_lut,
                                                         u32 min_count, u64 max_count, u64 *__restrict__ hi, u64 *__restrict__ lo,
                                                         u32 *__restrict__ cnt)
{
    __shared__ u32 sh_bytes[KMC_TILE * KMC_MAX_REC / 4 + 1];
    __shared__ u64 sh_lut[KMC_WIN + 1];
    __shared__ u64 sh_j0;
    const u32 rs = suffix_bytes + counter_bytes;
    const u64 t0 = (u64)blockIdx.x * KMC_TILE;
    if (t0 >= n) return;
    const u32 nt = (u32)min((u64)KMC_TILE, n - t0);
    // the tile's bytes: `rec` is 4-byte aligned and t0 * rs is a multiple of 4 (KMC_TILE is); reads stay inside
    // the buffer's padding (the launcher rounds the allocation up)
    const u32 nd = (nt * rs + 3) / 4;
    const u32 *src = (const u32 *)(rec + t0 * rs);
    for (u32 i = threadIdx.x; i < nd; i += TPB) sh_bytes[i] = __builtin_nontemporal_load(src + i);
    if (threadIdx.x == 0) sh_j0 = kmc_upper_bound(lut, 0, n_lut + 1, first_record + t0) - 1; // lut[0] == 0 <= every record
    __syncthreads();
    const u64 j0 = sh_j0;
    for (u32 i = threadIdx.x; i <= (u32)KMC_WIN; i += TPB) sh_lut[i] = j0 + i <= n_lut ? lut[j0 + i] : ~0ULL;
    __syncthreads();
    const u8 *b8 = (const u8 *)sh_bytes;
    const u64 pmask = prefix_len >= 32 ? ~0ULL : (1ULL << (2 * prefix_len)) - 1;
    for (u32 q = threadIdx.x; q < nt; q += TPB) {
        const u64 g = first_record + t0 + q;
        u64 j;
        if (g < sh_lut[KMC_WIN]) { // inside the window: last entry <= g
            u32 a = 0, b = KMC_WIN;
            while (a < b) {
                const u32 mid = (a + b) / 2;
                if (sh_lut[mid] <= g) a = mid + 1;
                else b = mid;
            }
            j = j0 + a - 1;
        } else
            j = kmc_upper_bound(lut, j0 + KMC_WIN, n_lut + 1, g) - 1;
        const u64 prefix = j & pmask;
        const u8 *r = b8 + q * rs;
        U128 v{0, 0};
        for (u32 s = 0; s < suffix_bytes; ++s) { // big-endian suffix
            v.hi = (v.hi << 8) | (v.lo >> 56);
            v.lo = (v.lo << 8) | r[s];
        }
        const U128 pre = shl128(U128{prefix, 0}, (int)(8 * suffix_bytes));
        v.lo |= pre.lo;
        v.hi |= pre.hi;
        u64 c = 0;
        for (u32 s = 0; s < counter_bytes; ++s) c |= (u64)r[suffix_bytes + s] << (8 * s);
        // CKMCFile::ReadNextKmer skips records outside [min_count, max_count]; a zero count adds nothing anywhere
        if (c < min_count || c > max_count) c = 0;
        hi[t0 + q] = v.hi;
        lo[t0 + q] = v.lo;
        cnt[t0 + q] = (u32)c;
    }
}

// debug: hash % size of packed k-mers (M-form, klen bases)
__global__ void __launch_bounds__(TPB) packed_index_kernel(const u64 *hi, const u64 *lo, u64 n, int klen, ModDesc mod,
                                                           u64 *out)
{
    const u64 i = (u64)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const U128 m{lo[i], hi[i]};
    const U128 l = mform_to_lform(m, klen);
    const U128 c = canon_sub(m, l, klen, 0, klen);
    out[i] = mod_size(xxh3_lform(c, klen), mod);
}

