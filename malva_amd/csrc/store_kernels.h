// store_kernels.h -- batch forms of every BF / KMAP call on ASCII rows, exact-map insert / rehash / dump, and the finalize pass (rank directory, gate rebuild)
// Part of the malva_hip translation unit: included by malva_hip.hip inside its anonymous namespace, after
// geno_dev.h (which brings xxh3_dev.h and kmer_dev.h).  See DESIGN.md section 4 for the kernels' rooflines.
#pragma once

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

enum RowOp { OP_BF_INSERT, OP_BF_TEST, OP_BF_INC, OP_BF_GET, OP_BF_INDEX, OP_MAP_TEST, OP_MAP_INC, OP_MAP_GET, OP_MAP_SET, OP_WEIGHT };

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
    bool want_map = OP == OP_MAP_TEST || OP == OP_MAP_INC || OP == OP_MAP_GET || OP == OP_MAP_SET;
    if (OP == OP_WEIGHT) want_map = is_ref[i] != 0;
    if (want_map) {
        U128 key;
        long long s = -1;
        const bool regular = pack_regular(can, k, (int)map.klen, &key);
        if (regular) { // counter id, or -1
            const u64 h = xxh3_bytes(can, k);
            s = map_find_id(map, key, h, mod_size(h, bf.mod));
        }
        if (irregular) irregular[i] = regular ? 0 : 1;
        if (OP == OP_MAP_TEST) ((u8 *)out)[i] = s >= 0;
        if (OP == OP_MAP_INC && s >= 0) atomicAdd(&map.vals[s], counters[i]);
        if (OP == OP_MAP_SET && s >= 0) map.vals[s] = counters[i]; // index load: the stored value of an imported key
        if (OP == OP_MAP_GET || OP == OP_WEIGHT) ((i32 *)out)[i] = s >= 0 ? (i32)map.vals[s] : 0;
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
        u32 r;
        if (bf_bit_rank(bf, idx, &r)) atomicAdd(&bf.counts[r], counters[i]);
    }
    if (OP == OP_BF_GET) ((uint16_t *)out)[i] = bf.counts ? (uint16_t)bf_count_at(bf, idx) : 0;
    if (OP == OP_WEIGHT) ((i32 *)out)[i] = bf.counts ? (i32)(uint16_t)bf_count_at(bf, idx) : 0;
}

// KMAP::add_key (kmap.hpp:108-112) for one regular key (canonical L-form, its XXH3): my_id = this insertion's row
// number, row0 = first row number of the current batch (a key met again from an EARLIER batch has its value reset,
// `kmers[ckmer] = 0`).  Callable from any kernel: a lane that finds a slot "being written" waits for the owner, which is
// always a lane inside this same loop (it won the CAS in this iteration and publishes before the iteration ends).
__device__ __forceinline__ void map_insert_key(const MapView &map, const BFView &bf, U128 key, u64 h, u32 my_id, u32 row0)
{
    gate_set(bf, mod_size(h, bf.mod));
    const u32 tag = map_tag(h);
    const u64 mask = (1ULL << map.cap_log2) - 1;
    u64 s = map_home(map, mod_size(h, bf.mod));
    bool done = false;
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

// KMAP::add_key for a batch of ASCII rows.  row0 = number of rows
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
    map_insert_key(map, bf, key, xxh3_bytes(can, k), row0 + (u32)i, row0);
}

__global__ void __launch_bounds__(TPB) map_clear_kernel(MapSlot *slots, u64 cap)
{
    const u64 s = (u64)blockIdx.x * TPB + threadIdx.x;
    if (s < cap) slots[s] = MapSlot{0u, 0xFFFFFFFFu, 0, 0, {0u, 0u}, {0ULL, 0ULL}, 0};
}
// move every published entry of an old table into a new (larger, empty) one
__global__ void __launch_bounds__(TPB) map_rehash_kernel(MapView oldm, MapView newm, ModDesc mod)
{
    const u64 s0 = (u64)blockIdx.x * TPB + threadIdx.x;
    if (s0 >= (1ULL << oldm.cap_log2)) return;
    if (oldm.slots[s0].tag < 2) return;
    U128 key{oldm.slots[s0].klo, oldm.slots[s0].khi};
    const u64 h = xxh3_lform(key, (int)oldm.klen);
    const u64 mask = (1ULL << newm.cap_log2) - 1;
    u64 s = map_home(newm, mod_size(h, mod));
    while (atomicCAS(&newm.slots[s].tag, 0u, map_tag(h)) != 0u) s = (s + 1) & mask;
    newm.slots[s].klo = key.lo;
    newm.slots[s].khi = key.hi;
    newm.slots[s].id = oldm.slots[s0].id;
}
// The filter's directory inside the records (MapSlot::bidx / brank): cleared, then one entry per set bit of the
// finalised filter.  Bit positions are distinct, so an entry is claimed with one CAS and never looked at again here.
__global__ void __launch_bounds__(TPB) bf_entries_clear_kernel(MapSlot *slots, u64 cap)
{
    const u64 s = (u64)blockIdx.x * TPB + threadIdx.x;
    if (s >= cap) return;
    slots[s].bidx[0] = slots[s].bidx[1] = 0;
    slots[s].brank[0] = slots[s].brank[1] = 0;
}
__global__ void __launch_bounds__(TPB) bf_entries_build_kernel(BFView bf, MapView m, u64 nwords)
{
    const u64 w = (u64)blockIdx.x * TPB + threadIdx.x;
    if (w >= nwords) return;
    u64 bits = bf.words[w];
    if (!bits) return;
    u32 rank = bf_rank(bf, w * 64);
    const u64 mask = (1ULL << m.cap_log2) - 1;
    while (bits) {
        const u64 pos = w * 64 + (u64)__builtin_ctzll(bits);
        bits &= bits - 1;
        u64 s = map_home(m, pos);
        for (;;) {
            int j = -1;
            if (atomicCAS((unsigned long long *)&m.slots[s].bidx[0], 0ULL, (unsigned long long)(pos + 1)) == 0ULL) j = 0;
            else if (atomicCAS((unsigned long long *)&m.slots[s].bidx[1], 0ULL, (unsigned long long)(pos + 1)) == 0ULL) j = 1;
            if (j >= 0) {
                m.slots[s].brank[j] = rank;
                break;
            }
            s = (s + 1) & mask;
        }
        ++rank;
    }
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

// positions of a filter's set bits into an open-addressing set (BFView::pos_set); `set` zeroed, capacity 2^log2 >= 2 x the bits
__global__ void __launch_bounds__(TPB) pos_set_build_kernel(const u64 *__restrict__ words, u64 nwords, u64 *set, u32 log2)
{
    const u64 w = (u64)blockIdx.x * TPB + threadIdx.x;
    if (w >= nwords) return;
    u64 x = words[w];
    const u64 mask = (1ULL << log2) - 1;
    while (x) {
        const u64 pos = w * 64 + (u64)(__ffsll((unsigned long long)x) - 1);
        x &= x - 1;
        u64 s = (pos * 0x9E3779B97F4A7C15ULL) >> (64 - log2);
        while (atomicCAS((unsigned long long *)&set[s], 0ULL, (unsigned long long)(pos + 1)) != 0ULL) s = (s + 1) & mask;
    }
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
// Exclusive scan of tile sums in place, total to *total (u64), in three launches (launch_tile_scan): per chunk of SCAN_CHUNK sums a
// workgroup adds them up; one workgroup scans the chunk totals; every workgroup rescans its chunk from its base.  (The single
// workgroup this replaces, each thread walking its own stretch, took 0.94 ms for the 3.1e5 tile sums of an 8e7-record cut: the
// loads of a wave were 1.2 KB apart and a thousand of them came one after the other.)
constexpr int SCAN_TPB = 1024, SCAN_PER = 8, SCAN_CHUNK = SCAN_TPB * SCAN_PER;
__device__ __forceinline__ unsigned long long block_sum_1024(unsigned long long v, unsigned long long *sh16)
{
    for (int d = 32; d; d >>= 1) v += __shfl_xor((long long)v, d, 64);
    if ((threadIdx.x & 63) == 0) sh16[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long t = 0;
    for (int w = 0; w < SCAN_TPB / 64; ++w) t += sh16[w];
    __syncthreads();
    return t;
}
__global__ void __launch_bounds__(SCAN_TPB) tile_reduce_kernel(const u32 *__restrict__ x, u64 n, unsigned long long *part)
{
    __shared__ unsigned long long sh16[SCAN_TPB / 64];
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK;
    unsigned long long v = 0;
    for (int j = 0; j < SCAN_PER; ++j) {
        const u64 i = base + (u64)j * SCAN_TPB + threadIdx.x;
        if (i < n) v += x[i];
    }
    const unsigned long long t = block_sum_1024(v, sh16);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void __launch_bounds__(SCAN_TPB) part_scan_kernel(unsigned long long *part, u64 n_part, unsigned long long *total)
{
    __shared__ unsigned long long sh[SCAN_TPB];
    unsigned long long carry = 0;
    for (u64 b = 0; b < n_part; b += SCAN_TPB) { // (one round up to 8.4e6 tile sums = 2.1e9 records)
        const u64 i = b + threadIdx.x;
        const unsigned long long v = i < n_part ? part[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int d = 1; d < SCAN_TPB; d <<= 1) {
            const unsigned long long a = (int)threadIdx.x >= d ? sh[threadIdx.x - d] : 0;
            __syncthreads();
            sh[threadIdx.x] += a;
            __syncthreads();
        }
        if (i < n_part) part[i] = carry + sh[threadIdx.x] - v;
        const unsigned long long round_total = sh[SCAN_TPB - 1];
        __syncthreads();
        carry += round_total;
    }
    if (threadIdx.x == 0) *total = carry;
}
__global__ void __launch_bounds__(SCAN_TPB) tile_rescan_kernel(u32 *x, u64 n, const unsigned long long *__restrict__ part)
{
    __shared__ unsigned long long sh16[SCAN_TPB / 64];
    const u64 base = (u64)blockIdx.x * SCAN_CHUNK;
    unsigned long long carry = part[blockIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = 0; j < SCAN_PER; ++j) {
        const u64 i = base + (u64)j * SCAN_TPB + threadIdx.x;
        const u32 v = i < n ? x[i] : 0u;
        unsigned long long incl = v;
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long up = (unsigned long long)__shfl_up((long long)incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) sh16[wave] = incl;
        __syncthreads();
        unsigned long long before = 0, round_total = 0;
        for (int w = 0; w < SCAN_TPB / 64; ++w) {
            if (w < wave) before += sh16[w];
            round_total += sh16[w];
        }
        if (i < n) x[i] = (u32)(carry + before + incl - v); // valid while the grand total fits 32 bits (checked by the host)
        carry += round_total;
        __syncthreads();
    }
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


// The other direction (vectors_current): the vectors set from the records' copies, after scans that added to the copies alone
// (MapView::lazy).  Every counter a lookup, an export or an exchange can ask for belongs to exactly one record -- vals[id] to
// the record of its key, counts[rank] to the record entry of its set bit -- so one pass over the table writes them all; a copy
// of an older epoch is zero.  The filter's counters come back as the reference's u16 cells (the vector's upper halves, which
// only ever matter modulo 2^16, restart at zero).
__global__ void __launch_bounds__(TPB) rec_collect_kernel(MapView m, u32 *__restrict__ counts, u32 epoch)
{
    const u64 n = 1ULL << m.cap_log2;
    for (u64 s = (u64)blockIdx.x * TPB + threadIdx.x; s < n; s += (u64)gridDim.x * TPB) {
        const uint4 *p = reinterpret_cast<const uint4 *>(&m.slots[s]);
        const uint4 a = p[0], b = p[1], c = p[2];
        const u64 b0 = c.x | (u64)c.y << 32, b1 = c.z | (u64)c.w << 32;
        if (a.x < 2 && !b0) continue; // an empty record
        const uint4 d = p[3]; // {cval, cbf}
        if (a.x >= 2) m.vals[a.y] = d.y == epoch ? d.x : 0u;
        if (counts) {
            const u32 both = d.w == epoch ? d.z : 0u;
            if (b0) counts[b.z] = both & 0xFFFFu;
            if (b1) counts[b.w] = both >> 16;
        }
    }
}

// Every record's counter copies set from the vectors (MapSlot::cval / cbf, epoch `epoch`): see records_current.
__global__ void __launch_bounds__(TPB) rec_publish_kernel(MapView m, const u32 *__restrict__ counts, u32 epoch)
{
    const u64 n = 1ULL << m.cap_log2;
    for (u64 s = (u64)blockIdx.x * TPB + threadIdx.x; s < n; s += (u64)gridDim.x * TPB) {
        MapSlot *rec = &m.slots[s];
        const uint4 *p = reinterpret_cast<const uint4 *>(rec);
        const uint4 a = p[0], b = p[1], c = p[2];
        const u64 b0 = c.x | (u64)c.y << 32, b1 = c.z | (u64)c.w << 32;
        if (a.x < 2 && !b0) continue; // an empty record: its copies are of epoch 0 and are never read
        const u32 val = a.x >= 2 ? m.vals[a.y] : 0u;
        u32 both = 0;
        if (counts) {
            if (b0) both |= counts[b.z] & 0xFFFFu;
            if (b1) both |= (counts[b.w] & 0xFFFFu) << 16;
        }
        rec->cval = (unsigned long long)epoch << 32 | val;
        rec->cbf = (unsigned long long)epoch << 32 | both;
    }
}
