// malva_hip.hip -- kernels and C ABI of the MI355X-native malva-geno hot path.
// See include/malva_hip.h for the interface and the reference lines each entry
// point replaces; DESIGN.md for the data layout and the roofline of each kernel.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
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

constexpr int TPB = 256;

// ---- ASCII rows -------------------------------------------------------------

struct RowIn {
    const u8 *p;
    __device__ __forceinline__ u32 operator()(int i) const { return p[i]; }
};
__device__ __forceinline__ int row_len(const u8 *row, int stride)
{
    const int lim = stride < MG_MAX_KMER + 1 ? stride : MG_MAX_KMER + 1;
    int n = 0;
    while (n < lim && row[n]) ++n;
    return n;
}

// canonical form of an ASCII k-mer as the exact map keys it: regular (pure
// upper-case ACGT, no NUL => never truncated) keys pack to an L-form;
// anything else is "irregular" and is kept by the host-side overflow list.
template <class CAN> __device__ __forceinline__ bool pack_regular(const CAN &c, int k, int klen, U128 *out)
{
    U128 v{0, 0};
    if (k != klen || k > MG_MAX_PACKED_K) return false;
    for (int i = 0; i < k; ++i) {
        const u32 code = code_of(c(i));
        if (code > 3) return false;
        if (i < 32) v.lo |= (u64)code << (2 * i);
        else v.hi |= (u64)code << (2 * (i - 32));
    }
    *out = v;
    return true;
}

enum RowOp { OP_BF_INSERT, OP_BF_TEST, OP_BF_INC, OP_BF_GET, OP_BF_INDEX, OP_MAP_TEST, OP_MAP_INC, OP_MAP_GET, OP_WEIGHT };

// One thread per row.  H4/H5/H7/H8 (bloom_filter.hpp:81-125) and H9
// (kmap.hpp:99-131) in batch form, plus the mixed lookup of set_coverages
// (main.cpp:166-170).  out type depends on the op.
template <int OP>
__global__ void __launch_bounds__(TPB) rows_kernel(const u8 *rows, size_t stride, size_t n, BFView bf, MapView map,
                                                   const u32 *counters, const u8 *is_ref, void *out, u8 *irregular)
{
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const u8 *row = rows + i * stride;
    const int k = row_len(row, (int)stride);
    CanonBytes<RowIn> can(RowIn{row}, k);
    bool want_map = OP == OP_MAP_TEST || OP == OP_MAP_INC || OP == OP_MAP_GET;
    if (OP == OP_WEIGHT) want_map = is_ref[i] != 0;
    if (want_map) {
        U128 key;
        long long s = -1;
        const bool regular = pack_regular(can, k, (int)map.klen, &key);
        if (regular) s = map_find(map, key, xxh3_bytes(can, k));
        if (irregular) irregular[i] = regular ? 0 : 1;
        if (OP == OP_MAP_TEST) ((u8 *)out)[i] = s >= 0;
        if (OP == OP_MAP_INC && s >= 0) atomicAdd(&map.vals[map.slots[s].id], counters[i]);
        if (OP == OP_MAP_GET || OP == OP_WEIGHT) ((i32 *)out)[i] = s >= 0 ? (i32)map.vals[map.slots[s].id] : 0;
        return;
    }
    const u64 idx = mod_size(xxh3_bytes(can, k), bf.mod);
    if (OP == OP_BF_INDEX) ((u64 *)out)[i] = idx;
    if (OP == OP_BF_INSERT) {
        atomicOr((unsigned long long *)&bf.words[idx >> 6], 1ULL << (idx & 63));
        gate_set(bf, idx);
    }
    if (OP == OP_BF_TEST) ((u8 *)out)[i] = bf_bit(bf, idx);
    if (OP == OP_BF_INC) {
        if (bf_bit(bf, idx)) atomicAdd(&bf.counts[bf_rank(bf, idx)], counters[i]);
    }
    if (OP == OP_BF_GET) ((uint16_t *)out)[i] = bf.counts && bf_bit(bf, idx) ? (uint16_t)bf.counts[bf_rank(bf, idx)] : 0;
    if (OP == OP_WEIGHT) ((i32 *)out)[i] = bf.counts && bf_bit(bf, idx) ? (i32)(uint16_t)bf.counts[bf_rank(bf, idx)] : 0;
}

// KMAP::add_key (kmap.hpp:108-112) for regular keys.  row0 = number of rows
// inserted by earlier calls (ids are global insertion rows).
__global__ void __launch_bounds__(TPB) map_insert_kernel(const u8 *rows, size_t stride, size_t n, MapView map, BFView bf,
                                                         u32 row0, u8 *irregular)
{
    const size_t i = (size_t)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const u8 *row = rows + i * stride;
    const int k = row_len(row, (int)stride);
    CanonBytes<RowIn> can(RowIn{row}, k);
    U128 key;
    const bool regular = pack_regular(can, k, (int)map.klen, &key);
    irregular[i] = regular ? 0 : 1;
    if (!regular) return;
    const u64 h = xxh3_bytes(can, k);
    gate_set(bf, mod_size(h, bf.mod));
    const u32 tag = map_tag(h);
    const u64 mask = (1ULL << map.cap_log2) - 1;
    u64 s = map_slot(map, h);
    const u32 my_id = row0 + (u32)i;
    bool done = false;
    // every lane retries inside one common loop, so a lane that owns a slot in
    // the "being written" state always finishes its publish before anyone spins on it
    for (int guard = 0; !done && guard < (1 << 30); ++guard) {
        u32 t = __hip_atomic_load(&map.slots[s].tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == 0) {
            t = atomicCAS(&map.slots[s].tag, 0u, 1u);
            if (t == 0) {
                map.slots[s].klo = key.lo;
                map.slots[s].khi = key.hi;
                atomicMin(&map.slots[s].id, my_id);
                __threadfence();
                __hip_atomic_store(&map.slots[s].tag, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                done = true;
                continue;
            }
        }
        if (t == 1) continue; // owner is publishing: look again
        if (t == tag) {
            __threadfence();
            const u64 a = __hip_atomic_load(&map.slots[s].klo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const u64 b = __hip_atomic_load(&map.slots[s].khi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (a == key.lo && b == key.hi) {
                const u32 old = atomicMin(&map.slots[s].id, my_id);
                if (old < row0) map.vals[old] = 0; // kmers[ckmer] = 0 on a key from an earlier call
                done = true;
                continue;
            }
        }
        s = (s + 1) & mask;
    }
}

__global__ void __launch_bounds__(TPB) map_clear_kernel(MapSlot *slots, u64 cap)
{
    const u64 s = (u64)blockIdx.x * TPB + threadIdx.x;
    if (s < cap) slots[s] = MapSlot{0u, 0xFFFFFFFFu, 0, 0, 0};
}
// move every published entry of an old table into a new (larger, empty) one
__global__ void __launch_bounds__(TPB) map_rehash_kernel(MapView oldm, MapView newm)
{
    const u64 s0 = (u64)blockIdx.x * TPB + threadIdx.x;
    if (s0 >= (1ULL << oldm.cap_log2)) return;
    if (oldm.slots[s0].tag < 2) return;
    U128 key{oldm.slots[s0].klo, oldm.slots[s0].khi};
    const u64 h = xxh3_lform(key, (int)oldm.klen);
    const u64 mask = (1ULL << newm.cap_log2) - 1;
    u64 s = map_slot(newm, h);
    while (atomicCAS(&newm.slots[s].tag, 0u, map_tag(h)) != 0u) s = (s + 1) & mask;
    newm.slots[s].klo = key.lo;
    newm.slots[s].khi = key.hi;
    newm.slots[s].id = oldm.slots[s0].id;
}
// gate bits of every published key (after a filter import rebuilt the gate from the bits alone)
__global__ void __launch_bounds__(TPB) map_gate_kernel(MapView m, BFView bf)
{
    const u64 s = (u64)blockIdx.x * TPB + threadIdx.x;
    if (s >= (1ULL << m.cap_log2) || m.slots[s].tag < 2) return;
    gate_set(bf, mod_size(xxh3_lform(U128{m.slots[s].klo, m.slots[s].khi}, (int)m.klen), bf.mod));
}

// list of published (key, id) for export
__global__ void __launch_bounds__(TPB) map_dump_kernel(MapView m, u64 *klo, u64 *khi, u32 *ids, unsigned long long *count)
{
    const u64 s = (u64)blockIdx.x * TPB + threadIdx.x;
    if (s >= (1ULL << m.cap_log2) || m.slots[s].tag < 2) return;
    const unsigned long long j = atomicAdd(count, 1ULL);
    klo[j] = m.slots[s].klo;
    khi[j] = m.slots[s].khi;
    ids[j] = m.slots[s].id;
}

// ---- finalize: rank directory, counters, summary ---------------------------

// per 512-bit block popcount, exclusive scan inside a tile of TPB blocks
__global__ void __launch_bounds__(TPB) blk_pop_kernel(const u64 *words, u64 nwords, u64 n_blk, u32 *blk, u32 *tile_sums)
{
    __shared__ u32 sh[TPB];
    const u64 b = (u64)blockIdx.x * TPB + threadIdx.x;
    u32 pop = 0;
    if (b < n_blk) {
        const u64 w0 = b * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (w0 + j < nwords) pop += (u32)__popcll(words[w0 + j]);
    }
    sh[threadIdx.x] = pop;
    __syncthreads();
    for (int d = 1; d < TPB; d <<= 1) {
        const u32 v = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    if (b < n_blk) blk[b] = sh[threadIdx.x] - pop;
    if (threadIdx.x == TPB - 1) tile_sums[blockIdx.x] = sh[TPB - 1];
}
// single workgroup: exclusive scan of tile sums in place; total to *total (u64)
__global__ void __launch_bounds__(1024) tile_scan_kernel(u32 *tile_sums, u64 n_tiles, unsigned long long *total)
{
    __shared__ unsigned long long sh[1024];
    const u64 per = (n_tiles + 1023) / 1024;
    const u64 lo = threadIdx.x * per, hi = lo + per < n_tiles ? lo + per : n_tiles;
    unsigned long long s = 0;
    for (u64 i = lo; i < hi; ++i) s += tile_sums[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const unsigned long long v = threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
        __syncthreads();
        sh[threadIdx.x] += v;
        __syncthreads();
    }
    unsigned long long run = sh[threadIdx.x] - s;
    for (u64 i = lo; i < hi; ++i) {
        const u32 v = tile_sums[i];
        tile_sums[i] = (u32)run; // valid while the grand total fits 32 bits (checked by the host)
        run += v;
    }
    if (threadIdx.x == 1023) *total = sh[1023];
}
__global__ void __launch_bounds__(TPB) blk_add_kernel(u32 *blk, u64 n_blk, const u32 *tile_sums, u32 total)
{
    const u64 b = (u64)blockIdx.x * TPB + threadIdx.x;
    if (b < n_blk) blk[b] += tile_sums[blockIdx.x];
    if (b == n_blk) blk[b] = total; // rank(size) (bloom_filter.hpp:97)
}
// gate entries of every set filter bit (used when a filter is imported rather than built by inserts)
__global__ void __launch_bounds__(TPB) gate_from_bits_kernel(BFView bf, u64 nwords)
{
    const u64 w = (u64)blockIdx.x * TPB + threadIdx.x;
    if (w >= nwords) return;
    u64 x = bf.words[w];
    while (x) {
        const int b = __ffsll((unsigned long long)x) - 1;
        gate_set(bf, w * 64 + b);
        x &= x - 1;
    }
}
// positions of the set bits in ascending (= counter) order; needs the rank directory
__global__ void __launch_bounds__(TPB) bit_positions_kernel(BFView bf, u64 nwords, u64 *out)
{
    const u64 w = (u64)blockIdx.x * TPB + threadIdx.x;
    if (w >= nwords) return;
    u64 x = bf.words[w];
    if (!x) return;
    u64 r = bf_rank(bf, w * 64);
    while (x) {
        out[r++] = w * 64 + (u64)(__ffsll((unsigned long long)x) - 1);
        x &= x - 1;
    }
}
__global__ void __launch_bounds__(TPB) set_bits_kernel(BFView bf, const u64 *pos, u64 n, u64 size, int *bad)
{
    const u64 i = (u64)blockIdx.x * TPB + threadIdx.x;
    if (i >= n) return;
    const u64 p = pos[i];
    if (p >= size || (i && pos[i - 1] >= p)) {
        *bad = 1; // out of range or not strictly ascending
        return;
    }
    atomicOr((unsigned long long *)&bf.words[p >> 6], 1ULL << (p & 63));
    gate_set(bf, p);
}
__global__ void __launch_bounds__(TPB) mask_u16_kernel(const u32 *in, uint16_t *out, u64 n)
{
    const u64 i = (u64)blockIdx.x * TPB + threadIdx.x;
    if (i < n) out[i] = (uint16_t)in[i];
}
__global__ void __launch_bounds__(TPB) widen_u16_kernel(const uint16_t *in, u32 *out, u64 n)
{
    const u64 i = (u64)blockIdx.x * TPB + threadIdx.x;
    if (i < n) out[i] = in[i];
}

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
//   C  load ROWS gate words                 -- random 8-byte loads from a 4 MiB bitmap (L2)
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
            gate[j] = (ablate & 1) ? 0ULL : bf.use_gate ? bf.gate[gate_word(bf, idx[j])] : ~0ULL;
#pragma unroll
        for (int j = 0; j < ROWS; ++j) { // D
            const u64 gm = gate_mask(bf, idx[j]);
            const bool open_j = valid[j] && !(ablate & 2) && (gate[j] & gm) == gm;
            if (ablate) asm volatile("" ::"v"((u32)idx[j]), "v"((u32)m[j].hi));
            if (WAVE) ws.push(open_j, m[j], count[j], open, &counters[0]);
            else st.push(open_j, m[j], count[j]);
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
            const long long s = map_find(map, key, xxh3_bytes(can, k));
            if (s >= 0) return (i32)map.vals[map.slots[s].id];
        }
        return 0;
    }
    const u64 idx = mod_size(xxh3_bytes(can, k), bf.mod);
    return bf_bit(bf, idx) ? (i32)(uint16_t)bf.counts[bf_rank(bf, idx)] : 0;
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
__device__ __forceinline__ void pack_span(const u8 *p, int n, u64 *codes, u64 *bad)
{
    const u64 addr = (u64)p;
    const u32 *q = (const u32 *)(addr & ~3ULL);
    const u32 sh = (u32)(addr & 3);
    u64 c = 0, b = 0;
    u32 prev = q[0];
    for (int j = 0; 4 * j < n; ++j) {
        const u32 next = q[j + 1];
        const u32 d = sh ? __builtin_amdgcn_alignbyte(next, prev, sh) : prev;
        u32 t = (d >> 1) & 0x03030303u; // per byte: A0 C1 G3 T2
        t ^= (t >> 1) & 0x01010101u;    //           A0 C1 G2 T3
        const u32 c8 = (t * 0x01041040u) >> 24;
        const int left = n - 4 * j;
        const u32 m = left >= 4 ? 0xFFFFFFFFu : ((1u << (8 * left)) - 1);
        if ((expand4(c8) ^ d) & m) b |= 0xFULL << (4 * j);
        c |= (u64)(left >= 4 ? c8 : (c8 & ((1u << (2 * left)) - 1))) << (8 * j);
        prev = next;
    }
    *codes = c;
    *bad = b;
}
__device__ __forceinline__ U128 shl128(U128 v, int s) // 0 <= s < 128
{
    U128 r;
    if (s == 0) return v;
    if (s < 64) {
        r.hi = (v.hi << s) | (v.lo >> (64 - s));
        r.lo = v.lo << s;
    } else {
        r.hi = v.lo << (s - 64);
        r.lo = 0;
    }
    return r;
}
// Fast path: both flanks and the allele are pure ACGT, so the signature is assembled in
// 2-bit form from flanks packed once per variant (shared by its alleles), canonicalised
// with integer compares and hashed with the register-resident XXH3 -- the same code the
// scan uses.  Anything else (N / IUPAC in the window, k outside 17..64) takes weight_bytes.
__global__ void __launch_bounds__(TPB) call_isolated_kernel(const u8 *reference, u64 n_vars, const u64 *pos,
                                                            const u32 *var_allele_off, const u32 *allele_off,
                                                            const u8 *pool, const float *freq, const u64 *present_mask,
                                                            const u8 *flags, int k, BFView bf, MapView map, GenoParams p,
                                                            u32 *cov_out, i32 *gt1, i32 *gt2, i32 *gq, u8 *status,
                                                            double *probs, const u64 *var_gt_off)
{
    const u64 v = (u64)blockIdx.x * TPB + threadIdx.x;
    if (v >= n_vars) return;
    const u32 a0 = var_allele_off[v], A = var_allele_off[v + 1] - a0;
    const u32 ref_size = allele_off[a0 + 1] - allele_off[a0];
    u32 *cov = cov_out + a0;
    for (u32 a = 0; a < A; ++a) cov[a] = 0;
    if (flags[v] & 1) {
        const u64 pm = present_mask[v];
        const u8 *site = reference + pos[v];
        const int lmax = k / 2, rmax = (k + 1) / 2;
        const bool packed_ok = k >= 17 && k <= MG_MAX_PACKED_K;
        // flanks as L-forms: left = ref[pos-lmax, pos), right = ref[pos+ref_size, +rmax)  (<= 32 bases each)
        u64 lf = 0, rf = 0, lbad = 0, rbad = 0;
        if (packed_ok) {
            pack_span(site - lmax, lmax, &lf, &lbad);
            pack_span(site + ref_size, rmax, &rf, &rbad);
        }
        for (u32 a = 0; a < A && a < 64; ++a) {
            if (!((pm >> a) & 1)) continue;
            const int alen = (int)(allele_off[a0 + a + 1] - allele_off[a0 + a]);
            const int mp = k / 2 - alen / 2, ms = (k + 1) / 2 - (alen - alen / 2);
            if (mp < 0 || ms < 0) continue; // alleles >= k take the general path (host contract)
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
            i32 w;
            if (fast) {
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
                const u64 h = xxh3_packed(key, k);
                if (a == 0) {
                    const long long s = map_find(map, key, h);
                    w = s >= 0 ? (i32)map.vals[map.slots[s].id] : 0;
                } else {
                    const u64 idx = mod_size(h, bf.mod);
                    w = bf_bit(bf, idx) ? (i32)(uint16_t)bf.counts[bf_rank(bf, idx)] : 0;
                }
            } else {
                w = weight_bytes(SigIn{site - mp, al, site + ref_size, mp, alen}, k, a == 0, bf, map);
            }
            if (w > 0) cov[a] = (u32)(float)(u32)w;
        }
    }
    genotype_one(cov, freq + a0, (int)A, p, gt1 + v, gt2 + v, gq + v, status + v, probs ? probs + var_gt_off[v] : nullptr);
}

// ---- general blocks on the device: chains, haplotype picks, signature assembly, lookup, coverage -----------
// VB::extract_kmers (var_block.hpp:95-219) with get_combs_on_the_right/left (:436-624), combine_combs (:630-677),
// get_ref_subs (:682-702) and build_alleles_combs / combine_haplotypes (:709-786), fused with set_coverages
// (main.cpp:151-184).  One workgroup per variant.  The reference builds the SET of distinct haplotype picks per
// chain and takes, per allele, the max over signatures; a max does not care about duplicates, so here every
// (chain, panel sample, haplotype pick) is simply evaluated and max-reduced -- thousands of redundant hashes
// are cheaper on this machine than a device-side set.
// Fixed capacities (chains per side, chain length, unphased fan-out): a variant that exceeds one is flagged in
// `overflow` and its block is redone by the host enumerator + mg_lookup_cover, so results never depend on them.
struct BlockBatch {
    const u8 *reference;      // concatenated contigs (mg_reference_upload)
    const u64 *blk_ref_base;  // per block: offset of the contig the block is evaluated against
    const u32 *blk_ref_len;   //            and its length
    const u32 *blk_var_off;   // [n_blocks + 1]
    const u32 *var_block;     // [n_vars] block of each variant
    const i32 *pos;           // 0-based position in the contig
    const u32 *ref_size, *min_size;
    const u8 *present;
    const u32 *var_allele_off; // [n_vars + 1] allele slots
    const u32 *allele_off;     // [n_slots + 1] into pool
    const u8 *pool;
    const u8 *canon;           // [n_slots] first allele index of the variant with the same text
    const uint16_t *gt;        // [n_vars][n_samples]: a1 | a2 << 7 | phased << 14
    u32 n_samples;
    int haploid, k;
};
constexpr int BK_MAXC = 8;   // chains per side
constexpr int BK_MAXL = 12;  // members per chain
constexpr int BK_MAXCOMB = 2 * BK_MAXL + 1;
constexpr int BK_MAXU = 10;  // unphased chain length (2^10 picks)

struct BkChains {
    int n;
    int len[BK_MAXC];
    int sum[BK_MAXC];
    int mem[BK_MAXC][BK_MAXL];
};

// get_combs_on_the_right (step +1) / _left (step -1); indices are batch-global variant indices inside [b0, b1)
__device__ bool bk_chains(const BlockBatch &B, int b0, int b1, int i, int step, BkChains *out)
{
    const int k = B.k;
    auto ov = [&](int x, int y) { // overlapping(left, right) with (x, y) given in scan order
        const int l = step > 0 ? x : y, r = step > 0 ? y : x;
        return B.pos[l] <= B.pos[r] && B.pos[r] < B.pos[l] + (int)B.ref_size[l];
    };
    auto nr = [&](int x, int y, int extra) {
        const int l = step > 0 ? x : y, r = step > 0 ? y : x;
        return B.pos[l] + (int)B.ref_size[l] - (int)B.min_size[l] - 1 + extra + (k + 1) / 2 >= B.pos[r];
    };
    out->n = 0;
    bool halt = false;
    for (int j = i + step; j >= b0 && j < b1 && !halt; j += step) {
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
                const long long s = map_find(map, key, h);
                return s >= 0 ? (i32)map.vals[map.slots[s].id] : 0;
            }
            const u64 idx = mod_size(h, bf.mod);
            return bf_bit(bf, idx) ? (i32)(uint16_t)bf.counts[bf_rank(bf, idx)] : 0;
        }
    }
    return weight_bytes(LdsBytes{buf}, len, is_ref, bf, map);
}

__global__ void __launch_bounds__(TPB) cover_blocks_kernel(BlockBatch B, u64 n_vars, BFView bf, MapView map, u32 *cov_out, u8 *overflow)
{
    __shared__ BkChains sh_left, sh_right;
    __shared__ int sh_comb[BK_MAXC * BK_MAXC][BK_MAXCOMB];
    __shared__ int sh_comb_len[BK_MAXC * BK_MAXC], sh_comb_mid[BK_MAXC * BK_MAXC];
    __shared__ int sh_ncomb, sh_bad;
    __shared__ u32 sh_cov[128];
    __shared__ u32 sh_slide[4]; // alleles (bit mask, 128 bits) that some sample carries alone and whole (len >= k)
    __shared__ u8 sh_buf[TPB][MG_MAX_PACKED_K];
    const int g = blockIdx.x;
    if ((u64)g >= n_vars) return;
    const u32 a0 = B.var_allele_off[g], A = B.var_allele_off[g + 1] - a0;
    const u32 blk = B.var_block[g];
    const int b0 = (int)B.blk_var_off[blk], b1 = (int)B.blk_var_off[blk + 1];
    const u8 *ref = B.reference + B.blk_ref_base[blk];
    const i32 ref_len = (i32)B.blk_ref_len[blk];
    const int k = B.k;
    for (u32 a = threadIdx.x; a < 128; a += TPB) sh_cov[a] = 0;
    if (threadIdx.x < 4) sh_slide[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        sh_bad = 0;
        sh_ncomb = 0;
        const bool eligible = B.present[g] && B.pos[g] >= k && B.pos[g] <= ref_len - k; // var_block.hpp:104
        if (A > 127 || k > MG_MAX_PACKED_K) sh_bad = 1;
        else if (eligible) {
            if (!bk_chains(B, b0, b1, g, -1, &sh_left) || !bk_chains(B, b0, b1, g, +1, &sh_right)) sh_bad = 1;
            else { // combine_combs
                const int nl = sh_left.n ? sh_left.n : 1, nrr = sh_right.n ? sh_right.n : 1;
                for (int l = 0; l < nl; ++l)
                    for (int r = 0; r < nrr; ++r) {
                        int *comb = sh_comb[sh_ncomb];
                        int len = 0;
                        if (sh_left.n)
                            for (int q = sh_left.len[l] - 1; q >= 0; --q) comb[len++] = sh_left.mem[l][q];
                        sh_comb_mid[sh_ncomb] = len;
                        comb[len++] = g;
                        if (sh_right.n)
                            for (int q = 0; q < sh_right.len[r]; ++q) comb[len++] = sh_right.mem[r][q];
                        sh_comb_len[sh_ncomb++] = len;
                    }
            }
        }
    }
    __syncthreads();
    if (sh_bad) {
        if (threadIdx.x == 0) overflow[g] = 1;
        for (u32 a = threadIdx.x; a < A; a += TPB) cov_out[a0 + a] = 0;
        return;
    }
    u8 *buf = sh_buf[threadIdx.x];
    bool bad = false;
    for (int c = 0; c < sh_ncomb; ++c) {
        const int *comb = sh_comb[c];
        const int m = sh_comb_len[c], jm = sh_comb_mid[c];
        const int first_pos = B.pos[comb[0]];
        const int last_end = B.pos[comb[m - 1]] + (int)B.ref_size[comb[m - 1]];
        for (u32 s = threadIdx.x; s < B.n_samples; s += TPB) {
            bool phased = true;
            if (!B.haploid)
                for (int j = 0; j < m; ++j) phased = phased && ((B.gt[(u64)comb[j] * B.n_samples + s] >> 14) & 1);
            u32 npick = B.haploid ? 1u : phased ? 2u : (1u << m);
            if (!B.haploid && !phased && m > BK_MAXU) {
                bad = true;
                continue;
            }
            for (u32 pick = 0; pick < npick; ++pick) {
                // allele of member j under this pick
                auto allele_of = [&](int j) -> u32 {
                    const u32 gt = B.gt[(u64)comb[j] * B.n_samples + s];
                    const u32 a1 = gt & 127, a2 = (gt >> 7) & 127;
                    if (B.haploid) return a1;
                    if (phased) return pick ? a2 : a1;
                    return (pick >> j) & 1 ? a2 : a1;
                };
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
                    continue;
                }
                const int first_part = mid_pos + mid_len / 2;
                const int mp = k / 2 - first_part;                  // missing_prefix (negative: cut)
                const int ms = (k + 1) / 2 - (len_v - first_part);  // missing_suffix
                if (first_pos - (mp > 0 ? mp : 0) < 0 || last_end + (ms > 0 ? ms : 0) > ref_len) {
                    bad = true; // the reference clips or throws here: leave it to the host path
                    continue;
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
                const i32 w = bk_weight(buf, k, mid_canon == 0, bf, map);
                if (w > 0) atomicMax(&sh_cov[mid_canon], (u32)w);
            }
        }
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
            const i32 w = bk_weight(buf, k, a == 0, bf, map);
            if (w > 0) {
                curr = (curr * (u32)n + (u32)w) / (u32)(n + 1);
                ++n;
            }
        }
        atomicMax(&sh_cov[a], curr);
    }
    __syncthreads();
    if (threadIdx.x == 0) overflow[g] = sh_bad ? 1 : 0;
    for (u32 a = threadIdx.x; a < A; a += TPB) cov_out[a0 + a] = sh_bad ? 0 : (u32)(float)sh_cov[a];
}

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
    Scratch s_rows, s_aux, s_out, s_irr, s_open[3], s_hit[3], s_misc[8];
    unsigned long long *d_hit_count = nullptr;
    double *d_ln = nullptr;
    float *d_eps = nullptr; // [2 * MG_EPS_TABLE]
    float eps_for = -1.f;
    u8 *d_ref = nullptr;
    size_t ref_len = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool stats_valid = false;
    u32 *joined = nullptr; // when set: one allocation holding [bf counters | map counters] (mg_counters_view)
    int use_summary = 1;
    bool gate_dirty = false; // something has been inserted into `bf`
    bool gate_fixed = false; // gate_log2 was set by the caller: do not resize at finalize
    int scan_rows = 2;    // table rows per thread per iteration of the filter kernel (swept: 2 is best)
    int scan_grid = 8192; // workgroups of the filter kernel (32 per CU; swept 2048..8192)
    int scan_ablate = 0;  // timing-only diagnostic, see scan_filter_kernel
    int scan_variant = 2; // filter-kernel VAR bits (staging / load width): 16-byte loads measured best
    int gate_k = 4;     // gate bits per entry (blocked Bloom filter inside one 64-bit word; swept 2..4)
    int gate_log2 = 25; // gate of at most 2^gate_log2 bits = 4 MiB (swept 24..26: 25 gives the best whole-scan time)
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
    v.use_gate = (c->use_summary && b.gate) ? 1 : 0;
    return v;
}
MapView view(const mg_ctx *c)
{
    const MapState &m = c->map;
    MapView v{};
    v.slots = m.slots;
    v.vals = m.vals;
    v.cap_log2 = m.cap_log2;
    v.klen = c->k;
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
    u32 want = 10;
    while ((1ULL << want) < need_rows * 4) ++want;
    if (!m.slots) return map_alloc(c, m, want);
    if (want > m.cap_log2) {
        MapView ov{};
        ov.slots = m.slots;
        ov.cap_log2 = m.cap_log2;
        ov.klen = c->k;
        m.slots = nullptr;
        TRY(map_alloc(c, m, want));
        MapView nv = view(c);
        hipLaunchKernelGGL(map_rehash_kernel, dim3(nblocks(1ULL << ov.cap_log2)), dim3(TPB), 0, c->stream, ov, nv);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        hipFree(ov.slots);
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
        if (hipMalloc(&b.words, b.nwords * 8) != hipSuccess || hipMemsetAsync(b.words, 0, b.nwords * 8, c->stream) != hipSuccess) {
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
    if (!c) return MG_OK;
    hipSetDevice(c->device);
    hipDeviceSynchronize();
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
    }
    map_free_table(c->map);
    hipFree(c->map.vals);
    for (Scratch *s : {&c->s_rows, &c->s_aux, &c->s_out, &c->s_irr}) hipFree(s->p);
    for (auto &s : c->s_open) hipFree(s.p);
    for (auto &s : c->s_hit) hipFree(s.p);
    for (auto &s : c->s_misc) hipFree(s.p);
    hipFree(c->d_hit_count);
    hipFree(c->d_ln);
    hipFree(c->d_eps);
    hipFree(c->d_ref);
    for (auto &e : c->ev)
        if (e) hipEventDestroy(e);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
    return MG_OK;
}

MG_EXPORT const char *mg_last_error(const mg_ctx *c) { return c ? c->err.c_str() : "null context"; }

MG_EXPORT int mg_set_stream(mg_ctx *c, void *s)
{
    if (!c) return MG_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    c->stream = s ? (hipStream_t)s : c->own_stream;
    return MG_OK;
}
MG_EXPORT int mg_synchronize(mg_ctx *c)
{
    if (!c) return MG_ERR_ARG;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}
MG_EXPORT int mg_set_option(mg_ctx *c, const char *name, int64_t value)
{
    if (!c || !name) return MG_ERR_ARG;
    if (!strcmp(name, "use_summary")) c->use_summary = value != 0;
    else if (!strcmp(name, "scan_rows")) c->scan_rows = (int)value;
    else if (!strcmp(name, "scan_ablate")) c->scan_ablate = (int)value;
    else if (!strcmp(name, "scan_variant")) c->scan_variant = (int)value & 3;
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

// ---- BF ---------------------------------------------------------------------------

MG_EXPORT int mg_bf_insert(mg_ctx *c, int which, const char *rows, size_t stride, size_t n)
{
    TRY(check_which(c, which));
    TRY(check_rows(c, rows, stride, n));
    // the reference lets add_key run in read mode too (the bit is set, the rank goes stale); refuse that
    if (c->bf[which].mode) return fail(c, MG_ERR_STATE, "mg_bf_insert after mg_bf_finalize");
    if (which == MG_BF_ALT) c->gate_dirty = true;
    return run_rows<OP_BF_INSERT>(c, which, rows, stride, n, nullptr, nullptr, nullptr, 0, nullptr);
}
MG_EXPORT int mg_bf_test(mg_ctx *c, int which, const char *rows, size_t stride, size_t n, uint8_t *out)
{
    TRY(check_which(c, which));
    TRY(check_rows(c, rows, stride, n));
    if (n && !out) return fail(c, MG_ERR_ARG, "out is NULL");
    return run_rows<OP_BF_TEST>(c, which, rows, stride, n, nullptr, nullptr, out, 1, nullptr);
}
MG_EXPORT int mg_debug_bf_index(mg_ctx *c, int which, const char *rows, size_t stride, size_t n, uint64_t *out)
{
    TRY(check_which(c, which));
    TRY(check_rows(c, rows, stride, n));
    if (n && !out) return fail(c, MG_ERR_ARG, "out is NULL");
    return run_rows<OP_BF_INDEX>(c, which, rows, stride, n, nullptr, nullptr, out, 8, nullptr);
}

MG_EXPORT int mg_bf_finalize(mg_ctx *c, int which)
{
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
    hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, c->stream, (u32 *)d_tiles, n_tiles, d_total);
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
        if (want != c->gate_log2) {
            c->gate_log2 = want;
            TRY(alloc_gate(c));
            hipLaunchKernelGGL(gate_from_bits_kernel, dim3(nblocks(b.nwords)), dim3(TPB), 0, c->stream, view(c, MG_BF_ALT), b.nwords);
            if (c->map.slots)
                hipLaunchKernelGGL(map_gate_kernel, dim3(nblocks(1ULL << c->map.cap_log2)), dim3(TPB), 0, c->stream, view(c),
                                   view(c, MG_BF_ALT));
            HIP_TRY(c, hipGetLastError());
            HIP_TRY(c, hipStreamSynchronize(c->stream));
        }
    }
    return MG_OK;
}

MG_EXPORT int mg_bf_increment(mg_ctx *c, int which, const char *rows, size_t stride, size_t n, const uint32_t *counters)
{
    TRY(check_which(c, which));
    TRY(check_rows(c, rows, stride, n));
    if (!c->bf[which].mode) return fail(c, MG_ERR_STATE, "BF::increment in write mode returns false");
    if (n && !counters) return fail(c, MG_ERR_ARG, "counters is NULL");
    return run_rows<OP_BF_INC>(c, which, rows, stride, n, counters, nullptr, nullptr, 0, nullptr);
}
MG_EXPORT int mg_bf_get_count(mg_ctx *c, int which, const char *rows, size_t stride, size_t n, uint16_t *out)
{
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
    TRY(check_which(c, which));
    if (size_bits) *size_bits = c->bf[which].size;
    if (n_set) *n_set = c->bf[which].nset;
    if (mode) *mode = c->bf[which].mode;
    return MG_OK;
}

// ---- KMAP ---------------------------------------------------------------------------

MG_EXPORT int mg_map_insert(mg_ctx *c, const char *rows, size_t stride, size_t n)
{
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
    if (!c || !n_keys) return MG_ERR_ARG;
    std::vector<u64> a, b;
    std::vector<u32> ids;
    TRY(map_dump(c, &a, &b, &ids));
    *n_keys = a.size() + c->map.irregular.size();
    return MG_OK;
}

// ---- reference scan --------------------------------------------------------------------

MG_EXPORT int mg_ref_scan(mg_ctx *c, const char *contig, size_t len)
{
    if (!c) return MG_ERR_ARG;
    if (len && !contig) return fail(c, MG_ERR_ARG, "contig is NULL");
    if (!c->bf[MG_BF_ALT].mode) return fail(c, MG_ERR_STATE, "mg_ref_scan needs `bf` finalised (main.cpp:378 precedes :383)");
    if (c->bf[MG_BF_CTX].mode) return fail(c, MG_ERR_STATE, "context filter already finalised");
    const size_t off = (c->ref_k - c->k) / 2;
    if (off > len) return fail(c, MG_ERR_ARG, "contig shorter than (ref_k-k)/2: the reference throws std::out_of_range here");
    if (len < c->ref_k) {
        // main.cpp:386-389 with both strings clipped by std::string(reference, pos, n): one test, no loop
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
    const u64 n_windows = len - c->ref_k + 1;
    // stream the contig through the device in slices (a human chromosome is a few hundred MB)
    const size_t slice = 256u << 20;
    for (u64 w0 = 0; w0 < n_windows; w0 += slice) {
        const u64 nw = n_windows - w0 < slice ? n_windows - w0 : slice;
        void *d;
        TRY(upload(c, c->s_rows, contig + w0, nw + c->ref_k - 1, &d));
        hipLaunchKernelGGL(ref_scan_kernel, dim3(nblocks(nw)), dim3(TPB), 0, c->stream, (const u8 *)d, w0, nw, (int)c->k,
                           (int)c->ref_k, view(c, MG_BF_ALT), view(c, MG_BF_CTX));
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return MG_OK;
}

// ---- KMC scan ----------------------------------------------------------------------------

namespace {
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
    switch ((c->scan_variant & 1) | ((c->scan_variant & 2) && vec_ok ? 2 : 0)) {
    case 1: launch_filter_var<KC, RC, ROWS, 1>(c, d_hi, d_lo, d_cnt, n, open); break;
    case 2: launch_filter_var<KC, RC, ROWS, 2>(c, d_hi, d_lo, d_cnt, n, open); break;
    case 3: launch_filter_var<KC, RC, ROWS, 3>(c, d_hi, d_lo, d_cnt, n, open); break;
    default: launch_filter_var<KC, RC, ROWS, 0>(c, d_hi, d_lo, d_cnt, n, open); break;
    }
}
template <int KC, int RC>
void launch_scan_chunk(mg_ctx *c, const u64 *d_hi, const u64 *d_lo, const u32 *d_cnt, u64 n, RowList open, RowList hits, bool timed)
{
    if (timed) hipEventRecord(c->ev[0], c->stream);
    switch (c->scan_rows) {
    case 1: launch_filter_rows<KC, RC, 1>(c, d_hi, d_lo, d_cnt, n, open); break;
    case 4: launch_filter_rows<KC, RC, 4>(c, d_hi, d_lo, d_cnt, n, open); break;
    default: launch_filter_rows<KC, RC, 2>(c, d_hi, d_lo, d_cnt, n, open); break;
    }
    if (timed) hipEventRecord(c->ev[1], c->stream);
    // the list lengths live on the device; fixed grids walk them with a stride, so no host round trip
    const unsigned grid = (unsigned)std::min<u64>(nblocks(n), 2048u);
    hipLaunchKernelGGL((scan_probe_kernel<KC, RC>), dim3(grid), dim3(TPB), 0, c->stream, (int)c->k, (int)c->ref_k, view(c, MG_BF_ALT),
                       view(c), open, hits, c->d_hit_count);
    if (timed) hipEventRecord(c->ev[2], c->stream);
    hipLaunchKernelGGL((scan_hits_kernel<KC, RC>), dim3(std::min(grid, 1024u)), dim3(TPB), 0, c->stream, (int)c->k, (int)c->ref_k,
                       view(c, MG_BF_ALT), view(c, MG_BF_CTX), hits, c->d_hit_count);
    if (timed) hipEventRecord(c->ev[3], c->stream);
}
} // namespace

MG_EXPORT int mg_kmc_scan_device(mg_ctx *c, const void *d_hi, const void *d_lo, const void *d_cnt, size_t n)
{
    if (!c) return MG_ERR_ARG;
    if (!c->bf[0].mode || !c->bf[1].mode) return fail(c, MG_ERR_STATE, "mg_kmc_scan needs both filters finalised");
    if (c->k < 17 || c->ref_k > MG_MAX_PACKED_K)
        return fail(c, MG_ERR_LIMIT, "packed scan supports 17 <= k <= ref_k <= 64 (k=%u ref_k=%u)", c->k, c->ref_k);
    if (n == 0) return MG_OK;
    if (!d_hi || !d_lo || !d_cnt) return fail(c, MG_ERR_ARG, "NULL table pointer");
    if (!c->map.slots) TRY(map_reserve(c, 0));
    const u64 chunk = 1ULL << 27; // rows per launch triple (bounds the two lists' worst-case size)
    const u64 cap = n < chunk ? n : chunk; // worst case (gate disabled): every row is listed
    void *p[6];
    Scratch *sc[6] = {&c->s_open[0], &c->s_open[1], &c->s_open[2], &c->s_hit[0], &c->s_hit[1], &c->s_hit[2]};
    for (int i = 0; i < 6; ++i) TRY(scratch(c, *sc[i], cap * (i % 3 == 2 ? 4 : 8), &p[i]));
    const RowList open{(u64 *)p[0], (u64 *)p[1], (u32 *)p[2]}, hits{(u64 *)p[3], (u64 *)p[4], (u32 *)p[5]};
    const bool d35_43 = c->k == 35 && c->ref_k == 43;
    c->stats_valid = false;
    HIP_TRY(c, hipMemsetAsync(c->d_hit_count, 0, 32, c->stream));
    for (u64 r0 = 0; r0 < n; r0 += chunk) {
        const u64 nr = n - r0 < chunk ? n - r0 : chunk;
        const u64 *ph = (const u64 *)d_hi + r0, *pl = (const u64 *)d_lo + r0;
        const u32 *pc = (const u32 *)d_cnt + r0;
        if (r0) HIP_TRY(c, hipMemsetAsync(c->d_hit_count, 0, 16, c->stream));
        if (d35_43) launch_scan_chunk<35, 43>(c, ph, pl, pc, nr, open, hits, r0 == 0);
        else launch_scan_chunk<0, 0>(c, ph, pl, pc, nr, open, hits, r0 == 0);
        HIP_TRY(c, hipGetLastError());
    }
    c->stats_valid = true;
    return MG_OK;
}

MG_EXPORT int mg_kmc_scan(mg_ctx *c, const uint64_t *hi, const uint64_t *lo, const uint32_t *cnt, size_t n)
{
    if (!c) return MG_ERR_ARG;
    if (n == 0) return MG_OK;
    if (!hi || !lo || !cnt) return fail(c, MG_ERR_ARG, "NULL table pointer");
    const size_t piece = 1u << 26; // host table streamed through the device in 64M-row pieces
    for (size_t r0 = 0; r0 < n; r0 += piece) {
        const size_t nr = n - r0 < piece ? n - r0 : piece;
        void *dh, *dl, *dc;
        TRY(upload(c, c->s_misc[5], hi + r0, nr * 8, &dh));
        TRY(upload(c, c->s_misc[6], lo + r0, nr * 8, &dl));
        TRY(upload(c, c->s_misc[7], cnt + r0, nr * 4, &dc));
        TRY(mg_kmc_scan_device(c, dh, dl, dc, nr));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
    return MG_OK;
}

MG_EXPORT int mg_scan_stats(mg_ctx *c, float *ms_out, uint64_t *n_hits)
{
    if (!c) return MG_ERR_ARG;
    if (!c->stats_valid) return fail(c, MG_ERR_STATE, "no scan has run");
    HIP_TRY(c, hipEventSynchronize(c->ev[3]));
    if (ms_out) {
        HIP_TRY(c, hipEventElapsedTime(&ms_out[0], c->ev[0], c->ev[1]));
        HIP_TRY(c, hipEventElapsedTime(&ms_out[1], c->ev[1], c->ev[2]));
        HIP_TRY(c, hipEventElapsedTime(&ms_out[2], c->ev[2], c->ev[3]));
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
    TRY(check_which(c, which));
    if (klen < 17 || klen > MG_MAX_PACKED_K) return fail(c, MG_ERR_LIMIT, "packed k-mers: 17 <= k <= 64");
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
    if (!c) return MG_ERR_ARG;
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    if (n_bf) *n_bf = c->bf[0].nset;
    if (n_map) *n_map = c->map.rows_total;
    return MG_OK;
}
MG_EXPORT int mg_counters_view(mg_ctx *c, void **d_ptr, uint64_t *n_bf, uint64_t *n_map)
{
    if (!c || !d_ptr) return MG_ERR_ARG;
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
    *d_ptr = c->joined;
    if (n_bf) *n_bf = b.nset;
    if (n_map) *n_map = m.rows_total;
    return MG_OK;
}
MG_EXPORT int mg_counters_export_device(mg_ctx *c, void *d_out)
{
    if (!c || !d_out) return MG_ERR_ARG;
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    const u64 nb = c->bf[0].nset, nm = c->map.rows_total;
    if (nb) HIP_TRY(c, hipMemcpyAsync(d_out, c->bf[0].counts, nb * 4, hipMemcpyDeviceToDevice, c->stream));
    if (nm) HIP_TRY(c, hipMemcpyAsync((u32 *)d_out + nb, c->map.vals, nm * 4, hipMemcpyDeviceToDevice, c->stream));
    return MG_OK;
}
MG_EXPORT int mg_counters_import_device(mg_ctx *c, const void *d_in)
{
    if (!c || !d_in) return MG_ERR_ARG;
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    const u64 nb = c->bf[0].nset, nm = c->map.rows_total;
    if (nb) HIP_TRY(c, hipMemcpyAsync(c->bf[0].counts, d_in, nb * 4, hipMemcpyDeviceToDevice, c->stream));
    if (nm) HIP_TRY(c, hipMemcpyAsync(c->map.vals, (const u32 *)d_in + nb, nm * 4, hipMemcpyDeviceToDevice, c->stream));
    return MG_OK;
}
MG_EXPORT int mg_counters_reset(mg_ctx *c)
{
    if (!c) return MG_ERR_ARG;
    if (c->bf[0].mode && c->bf[0].nset) HIP_TRY(c, hipMemsetAsync(c->bf[0].counts, 0, c->bf[0].nset * 4, c->stream));
    if (c->map.rows_total) HIP_TRY(c, hipMemsetAsync(c->map.vals, 0, c->map.rows_total * 4, c->stream));
    for (auto &kv : c->map.irregular) kv.second = 0;
    return MG_OK;
}

// ---- per-variant path -----------------------------------------------------------------------

MG_EXPORT int mg_lookup_cover(mg_ctx *c, const char *rows, size_t stride, size_t n_rows, const uint8_t *is_ref,
                              const uint64_t *sig_kmer_off, size_t n_sigs, const uint64_t *allele_sig_off, size_t n_alleles,
                              uint32_t *cov_out)
{
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
        hipLaunchKernelGGL(rows_kernel<OP_WEIGHT>, dim3(nblocks(n_rows)), dim3(TPB), 0, c->stream, (const u8 *)d_rows, stride,
                           n_rows, view(c, MG_BF_ALT), view(c), (const u32 *)nullptr, (const u8 *)d_isref, d_w,
                           (u8 *)nullptr);
        HIP_TRY(c, hipGetLastError());
    }
    hipLaunchKernelGGL(cover_kernel, dim3(nblocks(n_alleles)), dim3(TPB), 0, c->stream, (const i32 *)d_w, (const u64 *)d_so,
                       (const u64 *)d_ao, (u64)n_alleles, (u32 *)d_cov);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(cov_out, d_cov, 4 * n_alleles, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}

MG_EXPORT int mg_genotype(mg_ctx *c, const uint32_t *cov, const float *freq, const uint32_t *var_allele_off, size_t n_vars,
                          float error_rate, int max_cov, int haploid, int32_t *gt1, int32_t *gt2, int32_t *gq, uint8_t *status,
                          double *probs, const uint64_t *var_gt_off)
{
    if (!c) return MG_ERR_ARG;
    if (n_vars == 0) return MG_OK;
    if (!cov || !freq || !var_allele_off || !gt1 || !gt2 || !gq || !status) return fail(c, MG_ERR_ARG, "NULL argument");
    if (probs && !var_gt_off) return fail(c, MG_ERR_ARG, "probs needs var_gt_off");
    GenoParams p;
    TRY(fill_geno_params(c, error_rate, max_cov, haploid, &p));
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
    hipLaunchKernelGGL(genotype_kernel, dim3(nblocks(n_vars)), dim3(TPB), 0, c->stream, (const u32 *)d_cov, (const float *)d_freq,
                       (const u32 *)d_off, (u64)n_vars, p, (i32 *)d_g1, (i32 *)d_g2, (i32 *)d_gq, (u8 *)d_st, (double *)d_pr,
                       (const u64 *)d_go);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(gt1, d_g1, 4 * n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(gt2, d_g2, 4 * n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(gq, d_gq, 4 * n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(status, d_st, n_vars, hipMemcpyDeviceToHost, c->stream));
    if (probs && ng) HIP_TRY(c, hipMemcpyAsync(probs, d_pr, 8 * ng, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}

MG_EXPORT int mg_cover_blocks(mg_ctx *c, size_t n_blocks, const uint64_t *blk_ref_base, const uint32_t *blk_ref_len,
                              const uint32_t *blk_var_off, size_t n_vars, const int32_t *pos, const uint32_t *ref_size,
                              const uint32_t *min_size, const uint8_t *present, const uint32_t *var_allele_off,
                              const uint32_t *allele_off, const char *pool, size_t pool_len, const uint8_t *canon, const uint16_t *gt,
                              uint32_t n_samples, int haploid, uint32_t *cov_out, uint8_t *overflow_out)
{
    if (!c) return MG_ERR_ARG;
    if (n_vars == 0) return MG_OK;
    if (!blk_ref_base || !blk_ref_len || !blk_var_off || !pos || !ref_size || !min_size || !present || !var_allele_off || !allele_off ||
        !pool || !canon || (n_samples && !gt) || !cov_out || !overflow_out)
        return fail(c, MG_ERR_ARG, "NULL argument");
    if (!c->d_ref) return fail(c, MG_ERR_STATE, "mg_reference_upload first");
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    if (blk_var_off[n_blocks] != n_vars) return fail(c, MG_ERR_ARG, "block offsets do not close");
    if (!c->map.slots) TRY(map_reserve(c, 0));
    const size_t na = var_allele_off[n_vars];
    if (allele_off[na] > pool_len) return fail(c, MG_ERR_ARG, "allele offsets exceed the pool");
    std::vector<u32> var_block(n_vars);
    for (size_t b = 0; b < n_blocks; ++b) {
        if (blk_ref_base[b] + blk_ref_len[b] > c->ref_len) return fail(c, MG_ERR_ARG, "block %zu lies outside the uploaded reference", b);
        for (u32 v = blk_var_off[b]; v < blk_var_off[b + 1]; ++v) var_block[v] = (u32)b;
    }
    BlockBatch B{};
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
    TRY(upload(c, c->s_open[2], gt, 2 * (size_t)n_vars * n_samples, &d[12]));
    void *d_cov, *d_ovf;
    TRY(scratch(c, c->s_out, 4 * na, &d_cov));
    TRY(scratch(c, c->s_irr, n_vars, &d_ovf));
    B.reference = c->d_ref;
    B.blk_ref_base = (const u64 *)d[0]; B.blk_ref_len = (const u32 *)d[1]; B.blk_var_off = (const u32 *)d[2];
    B.var_block = (const u32 *)d[3]; B.pos = (const i32 *)d[4]; B.ref_size = (const u32 *)d[5]; B.min_size = (const u32 *)d[6];
    B.present = (const u8 *)d[7]; B.var_allele_off = (const u32 *)d[8]; B.allele_off = (const u32 *)d[9]; B.pool = (const u8 *)d[10];
    B.canon = (const u8 *)d[11]; B.gt = (const uint16_t *)d[12];
    B.n_samples = n_samples; B.haploid = haploid; B.k = (int)c->k;
    hipLaunchKernelGGL(cover_blocks_kernel, dim3((unsigned)n_vars), dim3(TPB), 0, c->stream, B, (u64)n_vars, view(c, MG_BF_ALT), view(c),
                       (u32 *)d_cov, (u8 *)d_ovf);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipMemcpyAsync(cov_out, d_cov, 4 * na, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipMemcpyAsync(overflow_out, d_ovf, n_vars, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return MG_OK;
}

MG_EXPORT int mg_reference_upload(mg_ctx *c, const char *ascii, size_t len)
{
    if (!c || (len && !ascii)) return MG_ERR_ARG;
    if (c->d_ref) hipFree(c->d_ref);
    c->d_ref = nullptr;
    c->ref_len = 0;
    HIP_TRY(c, hipMalloc(&c->d_ref, len + 64)); // padded: pack_span reads whole aligned dwords around a window
    HIP_TRY(c, hipMemset(c->d_ref, 0, len + 64));
    if (len) HIP_TRY(c, hipMemcpy(c->d_ref, ascii, len, hipMemcpyHostToDevice));
    c->ref_len = len;
    return MG_OK;
}

MG_EXPORT int mg_call_isolated_device(mg_ctx *c, size_t n_vars, const void *d_pos, const void *d_var_allele_off,
                                      const void *d_allele_off, const void *d_allele_pool, const void *d_freq,
                                      const void *d_present_mask, const void *d_flags, float error_rate, int max_cov, int haploid,
                                      void *d_cov_out, void *d_gt1, void *d_gt2, void *d_gq, void *d_status, void *d_probs,
                                      const void *d_var_gt_off)
{
    if (!c) return MG_ERR_ARG;
    if (n_vars == 0) return MG_OK;
    if (!c->d_ref) return fail(c, MG_ERR_STATE, "mg_reference_upload first");
    if (!c->bf[0].mode) return fail(c, MG_ERR_STATE, "`bf` not finalised");
    if (!c->map.slots) TRY(map_reserve(c, 0));
    GenoParams p;
    TRY(fill_geno_params(c, error_rate, max_cov, haploid, &p));
    hipLaunchKernelGGL(call_isolated_kernel, dim3(nblocks(n_vars)), dim3(TPB), 0, c->stream, (const u8 *)c->d_ref, (u64)n_vars,
                       (const u64 *)d_pos, (const u32 *)d_var_allele_off, (const u32 *)d_allele_off, (const u8 *)d_allele_pool,
                       (const float *)d_freq, (const u64 *)d_present_mask, (const u8 *)d_flags, (int)c->k, view(c, MG_BF_ALT),
                       view(c), p, (u32 *)d_cov_out, (i32 *)d_gt1, (i32 *)d_gt2, (i32 *)d_gq, (u8 *)d_status, (double *)d_probs,
                       (const u64 *)d_var_gt_off);
    HIP_TRY(c, hipGetLastError());
    return MG_OK;
}

MG_EXPORT int mg_call_isolated(mg_ctx *c, size_t n_vars, const uint64_t *pos, const uint32_t *var_allele_off,
                               const uint32_t *allele_off, const char *allele_pool, size_t pool_len, const float *freq,
                               const uint64_t *present_mask, const uint8_t *flags, float error_rate, int max_cov, int haploid,
                               uint32_t *cov_out, int32_t *gt1, int32_t *gt2, int32_t *gq, uint8_t *status, double *probs,
                               const uint64_t *var_gt_off)
{
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
            if (pos[v] < c->k || pos[v] + rs + c->k > c->ref_len)
                return fail(c, MG_ERR_ARG, "variant %zu flagged eligible but within k of the reference buffer end", v);
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

MG_EXPORT int mg_bf_export(mg_ctx *c, int which, uint64_t *words_out, uint16_t *counts_out)
{
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
    TRY(check_rows(c, rows, stride, n));
    if (n == 0) return MG_OK;
    const u64 row0 = c->map.rows_total;
    TRY(mg_map_insert(c, rows, stride, n));
    if (vals) {
        // imported keys are distinct, so row i keeps id row0 + i; irregular rows go to the overflow list
        HIP_TRY(c, hipMemcpy(c->map.vals + row0, vals, n * 4, hipMemcpyHostToDevice));
        for (size_t i = 0; i < n; ++i) {
            std::string key(rows + i * stride, strnlen(rows + i * stride, stride));
            auto it = c->map.irregular.find(key);
            if (it != c->map.irregular.end()) it->second = vals[i];
        }
    }
    return MG_OK;
}
