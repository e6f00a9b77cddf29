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
__global__ void __launch_bounds__(TPB) ref_scan_kernel(const u8 *contig, u64 w0, u64 n_windows, int k, int ref_k, BFView bf,
                                                       BFView ctx)
{
    __shared__ u8 sh[TPB + MG_MAX_KMER];
    const u64 p0 = (u64)blockIdx.x * TPB;
    const u64 avail = n_windows - p0 < TPB ? n_windows - p0 : TPB;
    const int nbytes = (int)avail + ref_k - 1;
    for (int i = threadIdx.x; i < nbytes; i += TPB) sh[i] = contig[p0 + i];
    __syncthreads();
    if (threadIdx.x >= avail) return;
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
};
template <int CAP> struct BlockStage {
    u64 *hi, *lo; // [CAP]
    u32 *cnt;     // [CAP]
    u32 *n;       // entries staged
    unsigned long long *base;
    // every lane of the wave must call this (it ballots)
    __device__ __forceinline__ void push(bool take, U128 m, u32 count)
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
                g.hi[b + j] = hi[j];
                g.lo[b + j] = lo[j];
                g.cnt[b + j] = cnt[j];
            }
            __syncthreads();
            if (threadIdx.x == 0) *n = 0;
        }
        __syncthreads();
    }
};

// The same per wave (no workgroup barrier anywhere): the four waves of a workgroup then never
// wait for each other, which matters in the filter kernel where the barrier pair per iteration
// made every wave run at the pace of the slowest.
template <int WCAP> struct WaveStage {
    u64 *hi, *lo; // this wave's [WCAP] slices
    u32 *cnt;
    int staged;   // wave-uniform
    __device__ __forceinline__ void flush(const RowList &g, unsigned long long *g_count)
    {
        const int lane = threadIdx.x & 63;
        unsigned long long b = 0;
        if (lane == 0) b = atomicAdd(g_count, (unsigned long long)staged);
        b = __shfl(b, 0, 64);
        for (int j = lane; j < staged; j += 64) {
            g.hi[b + j] = hi[j];
            g.lo[b + j] = lo[j];
            g.cnt[b + j] = cnt[j];
        }
        staged = 0;
        __builtin_amdgcn_wave_barrier();
    }
    // every lane of the wave must call this (it ballots)
    __device__ __forceinline__ void push(bool take, U128 m, u32 count, const RowList &g, unsigned long long *g_count)
    {
        const u64 mask = __ballot(take);
        if (!mask) return;
        if (take) {
            const int q = staged + __popcll(mask & ((1ULL << (threadIdx.x & 63)) - 1));
            lo[q] = m.lo;
            hi[q] = m.hi;
            cnt[q] = count;
        }
        staged += __popcll(mask);
        __builtin_amdgcn_wave_barrier();
        if (staged > WCAP - 64) flush(g, g_count);
    }
};

// counters[0] = open rows, [1] = hit rows of the current chunk, [2] = hit rows of the whole call
//
// ROWS table rows per thread and iteration, in phases so that the memory operations of
// one phase are all in flight together:
//   A  load ROWS x (hi, lo, cnt)            -- coalesced, non-temporal: the only HBM stream
//   B  canonicalise, XXH3, slot             -- pure VALU
//   C  load ROWS gate words                 -- random 8-byte loads from a 4 MiB bitmap (L2): the gate, or the
//                                              coarse gate in front of it when the index is large
//   D  test, stage open rows
// `ablate` is a timing-only diagnostic (results are wrong when it is non-zero):
// 1 = no gate load, 2 = gate load but nothing passes, 4 = no XXH3, 8 = no canonicalisation.
// VAR bit 0: per-wave staging (no barriers) instead of per-workgroup; bit 1 (ROWS == 2 only): each
// thread takes two ADJACENT rows with 16-byte loads instead of two rows TPB apart with 8-byte loads.
template <int KC, int RC, int ROWS, int VAR>
__global__ void __launch_bounds__(TPB) scan_filter_kernel(const u64 *__restrict__ hi, const u64 *__restrict__ lo,
                                                          const u32 *__restrict__ cnt, u64 n, int k_rt, int r_rt, BFView bf,
                                                          RowList open, unsigned long long *counters, int ablate)
{
    constexpr bool WAVE = VAR & 1, VEC = (VAR & 2) && ROWS == 2;
    constexpr int CAP = WAVE ? (TPB / 64) * 192 : TPB * ROWS + 256;
    __shared__ u64 sh_hi[CAP], sh_lo[CAP];
    __shared__ u32 sh_cnt[CAP];
    __shared__ u32 sh_n;
    __shared__ unsigned long long sh_base;
    BlockStage<CAP> st{sh_hi, sh_lo, sh_cnt, &sh_n, &sh_base};
    const int wv = threadIdx.x >> 6;
    WaveStage<192> ws{sh_hi + wv * 192, sh_lo + wv * 192, sh_cnt + wv * 192, 0};
    if (!WAVE) {
        if (threadIdx.x == 0) sh_n = 0;
        __syncthreads();
    }
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    const u64 step = (u64)gridDim.x * TPB * ROWS;
    for (u64 base = (u64)blockIdx.x * TPB * ROWS; base < n; base += step) {
        U128 m[ROWS];
        u32 count[ROWS];
        u64 idx[ROWS], gate[ROWS];
        bool valid[ROWS];
        if (VEC && base + (u64)TPB * 2 <= n) { // A, whole tile inside the table (table bases are 16-byte aligned)
            typedef unsigned long long __attribute__((ext_vector_type(2))) v2u64;
            typedef unsigned int __attribute__((ext_vector_type(2))) v2u32;
            const u64 i = base + 2 * (u64)threadIdx.x;
            const v2u64 l2 = __builtin_nontemporal_load((const v2u64 *)(lo + i));
            const v2u64 h2 = __builtin_nontemporal_load((const v2u64 *)(hi + i));
            const v2u32 c2 = __builtin_nontemporal_load((const v2u32 *)(cnt + i));
            m[0] = U128{l2.x, h2.x};
            m[ROWS - 1] = U128{l2.y, h2.y};
            count[0] = c2.x;
            count[ROWS - 1] = c2.y;
            valid[0] = valid[ROWS - 1] = true;
        } else {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) { // A
                const u64 i = VEC ? base + 2 * (u64)threadIdx.x + j : base + (u64)j * TPB + threadIdx.x;
                valid[j] = i < n && (!VEC || i < base + (u64)TPB * 2);
                m[j].lo = valid[j] ? __builtin_nontemporal_load(lo + i) : 0;
                m[j].hi = valid[j] ? __builtin_nontemporal_load(hi + i) : 0;
                count[j] = valid[j] ? __builtin_nontemporal_load(cnt + i) : 0;
            }
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { // B
            U128 c = m[j];
            if (!(ablate & 8)) c = canon_sub(m[j], mform_to_lform(m[j], r), r, off, k);
            const u64 h = (ablate & 4) ? (c.lo ^ c.hi) * 0x9E3779B97F4A7C15ULL : xxh3_packed_k<KC>(c, k);
            idx[j] = mod_size(h, bf.mod);
        }
#pragma unroll
        for (int j = 0; j < ROWS; ++j) // C
            gate[j] = (ablate & 1) ? 0ULL : !bf.use_gate ? ~0ULL : bf.pregate ? bf.pregate[pre_word(bf, idx[j])] : bf.gate[gate_word(bf, idx[j])];
        bool open_j[ROWS];
        if (bf.pregate && bf.use_gate) { // coarse gate first; the fine gate's line only for the rows that pass it,
#pragma unroll                           // and all of those loads in flight together
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
            const bool take = open_j[j] && !(ablate & 2) && (gate[j] & gm) == gm;
            if (ablate) asm volatile("" ::"v"((u32)idx[j]), "v"((u32)m[j].hi));
            if (WAVE) ws.push(take, m[j], count[j], open, &counters[0]);
            else st.push(take, m[j], count[j]);
        }
        if (!WAVE) st.flush_if_above(CAP - TPB * ROWS, open, &counters[0]); // room for one more full iteration
    }
    if (WAVE) {
        if (ws.staged) ws.flush(open, &counters[0]);
    } else
        st.flush_if_above(0, open, &counters[0]);
}

template <int KC, int RC>
__global__ void __launch_bounds__(TPB) scan_probe_kernel(int k_rt, int r_rt, BFView bf, MapView map, RowList open, RowList hits,
                                                         unsigned long long *counters)
{
    constexpr int CAP = TPB + 256;
    __shared__ u64 sh_hi[CAP], sh_lo[CAP];
    __shared__ u32 sh_cnt[CAP];
    __shared__ u32 sh_n;
    __shared__ unsigned long long sh_base;
    BlockStage<CAP> st{sh_hi, sh_lo, sh_cnt, &sh_n, &sh_base};
    if (threadIdx.x == 0) sh_n = 0;
    __syncthreads();
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    const u64 n_open = counters[0];
    const u64 step = (u64)gridDim.x * TPB;
    for (u64 base = (u64)blockIdx.x * TPB; base < n_open; base += step) {
        const u64 j = base + threadIdx.x;
        bool hit = false;
        U128 m{0, 0};
        u32 count = 0;
        if (j < n_open) {
            m = U128{open.lo[j], open.hi[j]};
            count = open.cnt[j];
            const U128 c = canon_sub(m, mform_to_lform(m, r), r, off, k);
            const u64 h = xxh3_packed_k<KC>(c, k);
            const u64 idx = mod_size(h, bf.mod);
            const u64 word = bf.words[idx >> 6];
            const long long s = map_find(map, c, h);
            if (s >= 0) atomicAdd(&map.vals[map.slots[s].id], count); // ref_bf.increment (main.cpp:495)
            hit = (word >> (idx & 63)) & 1;
        }
        st.push(hit, m, count);
        st.flush_if_above(CAP - TPB, hits, &counters[1]);
    }
    st.flush_if_above(0, hits, &counters[1]);
}

template <int KC, int RC>
__global__ void __launch_bounds__(TPB) scan_hits_kernel(int k_rt, int r_rt, BFView bf, BFView ctx, RowList hits,
                                                        unsigned long long *counters)
{
    const int k = KC > 0 ? KC : k_rt, r = RC > 0 ? RC : r_rt;
    const int off = (r - k) / 2;
    const u64 nh = counters[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) counters[2] += nh;
    for (u64 j = (u64)blockIdx.x * TPB + threadIdx.x; j < nh; j += (u64)gridDim.x * TPB) {
        const U128 m{hits.lo[j], hits.hi[j]};
        const U128 l = mform_to_lform(m, r);
        const U128 cc = canon_sub(m, l, r, 0, r);
        const u64 cidx = mod_size(xxh3_packed_k<RC>(cc, r), ctx.mod);
        if (bf_bit(ctx, cidx)) continue;                                  // context_bf.test_key (main.cpp:496)
        const u64 idx = mod_size(xxh3_packed_k<KC>(canon_sub(m, l, r, off, k), k), bf.mod);
        atomicAdd(&bf.counts[bf_rank(bf, idx)], hits.cnt[j]);             // bf.increment (main.cpp:498)
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
    out[i] = mod_size(xxh3_packed(c, klen), mod);
}

