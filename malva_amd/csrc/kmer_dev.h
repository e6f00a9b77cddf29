// Device-side k-mer primitives: the RCN complement table and canonical form of
// the reference (bloom_filter.hpp:36-65, kmap.hpp:86-97) for ASCII text, the same
// on 2-bit packed ACGT strings, hash -> slot reduction, and read-only views of
// the two stores (Bloom filter + rank directory, exact map).
#pragma once
#include "xxh3_dev.h"

namespace mg {

typedef int32_t i32;

// ---- ASCII ------------------------------------------------------------------

// RCN[128] of bloom_filter.hpp:36-50; every byte not listed (and every byte
// >= 128, which the reference would index out of bounds) complements to 0.
__device__ __forceinline__ u32 rcn(u32 c)
{
    switch (c) {
    case 'A': return 'T';
    case 'C': return 'G';
    case 'G': return 'C';
    case 'T': return 'A';
    case 'N': return 'N';
    case 'a': return 'T';
    case 'c': return 'G';
    case 'g': return 'G'; // sic (table index 103)
    case 't': return 'A';
    case 'n': return 'N';
    default: return 0;
    }
}

// A k-mer seen through a byte accessor IN(i), i in [0,k), oriented as
// BF::_canonical leaves it: strcmp(kmer, rc) < 0 ? kmer : rc.  A NUL in rc ends
// strcmp's walk with kmer's byte greater, so rc is kept.
template <class IN> struct CanonBytes {
    IN in;
    int k;
    bool fwd;
    __device__ __forceinline__ CanonBytes(IN in_, int k_) : in(in_), k(k_), fwd(false)
    {
        for (int i = 0; i < k; ++i) {
            const u32 a = in(i), b = rcn(in(k - 1 - i));
            if (a != b) {
                fwd = a < b;
                break;
            }
        }
    }
    __device__ __forceinline__ u32 operator()(int i) const { return fwd ? in(i) : rcn(in(k - 1 - i)); }
};

// 2-bit code of an upper-case base, 4 for anything else
__device__ __forceinline__ u32 code_of(u32 c)
{
    return c == 'A' ? 0u : c == 'C' ? 1u : c == 'G' ? 2u : c == 'T' ? 3u : 4u;
}

// ---- packed -----------------------------------------------------------------
// Two layouts of a string of n <= 64 bases in a 128-bit value (hi:lo):
//   M-form: base i at bits 2(n-1-i)   (MSB-first, right aligned: the ABI's table format)
//   L-form: base i at bits 2i         (LSB-first: byte order of the ASCII rendering)
// For a k-mer x with reverse complement rc:  L(rc) = ~M(x) & mask, and
// strcmp(x, rc) < 0  <=>  L(x) < L(rc) as integers, so the canonical string is the
// integer minimum of the two L-forms.

struct U128 {
    u64 lo, hi;
};

// order of the 32 two-bit codes reversed: bit reversal (v_bfrev_b32 per half), then the two bits of every code
// swapped back, per 32-bit half so that it is two shifts and one v_bfi_b32 each
__device__ __forceinline__ u32 pairswap32(u32 v) { return ((v >> 1) & 0x55555555u) | ((v << 1) & ~0x55555555u); }
__device__ __forceinline__ u64 pairrev64(u64 x)
{
    const u32 lo = pairswap32(__brev((u32)(x >> 32))), hi = pairswap32(__brev((u32)x));
    return (u64)lo | ((u64)hi << 32);
}
__device__ __forceinline__ U128 shr128(U128 v, int s) // 0 <= s < 128
{
    U128 r;
    if (s == 0) return v;
    if (s < 64) {
        r.lo = (v.lo >> s) | (v.hi << (64 - s));
        r.hi = v.hi >> s;
    } else {
        r.lo = v.hi >> (s - 64);
        r.hi = 0;
    }
    return r;
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
__device__ __forceinline__ U128 mask128(int bits) // 0 < bits <= 128
{
    U128 m;
    m.lo = bits >= 64 ? ~0ULL : ((1ULL << bits) - 1);
    m.hi = bits >= 128 ? ~0ULL : (bits > 64 ? ((1ULL << (bits - 64)) - 1) : 0);
    return m;
}
__device__ __forceinline__ bool lt128(U128 a, U128 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }

// L-form of the forward string from its M-form (n bases)
__device__ __forceinline__ U128 mform_to_lform(U128 m, int n)
{
    U128 r;
    r.lo = pairrev64(m.hi);
    r.hi = pairrev64(m.lo);
    return shr128(r, 2 * (64 - n));
}
// canonical L-form of the sub-string [off, off+k) of an n-base string given both forms
__device__ __forceinline__ U128 canon_sub(U128 mform, U128 lform, int n, int off, int k)
{
    const U128 mk = mask128(2 * k);
    U128 f = shr128(lform, 2 * off);
    f.lo &= mk.lo;
    f.hi &= mk.hi;
    U128 m = shr128(mform, 2 * (n - k - off));
    U128 r;
    r.lo = ~m.lo & mk.lo;
    r.hi = ~m.hi & mk.hi;
    return lt128(f, r) ? f : r;
}
__device__ __forceinline__ u64 xxh3_packed(U128 c, int len)
{
    return len > 32 ? xxh3_packed_33to64(c.lo, c.hi, len) : xxh3_packed_17to32(c.lo, c.hi, len);
}
// XXH3 of the ASCII rendering of a packed canonical key (any length 1..64)
struct LformIn {
    U128 v;
    __device__ __forceinline__ u32 operator()(int i) const
    {
        const u32 c = (u32)((i < 32 ? v.lo >> (2 * i) : v.hi >> (2 * (i - 32))) & 3);
        return (0x54474341u >> (8 * c)) & 0xFF;
    }
};
__device__ __forceinline__ u64 xxh3_lform(U128 key, int len)
{
    return len >= 17 ? xxh3_packed(key, len) : xxh3_bytes(LformIn{key}, len);
}
// LEN known at compile time (33..64): the fixed-length form; LEN == 0: runtime length (any 1..64)
template <int LEN> __device__ __forceinline__ u64 xxh3_packed_k(U128 c, int len_rt)
{
    if constexpr (LEN >= 33 && LEN <= 64) return xxh3_packed_fixed<LEN>(c.lo, c.hi);
    else return xxh3_lform(c, len_rt);
}
// the same with the ASCII table of ascii_lut_fill in LDS (used where LEN is fixed; ignored otherwise)
template <int LEN> __device__ __forceinline__ u64 xxh3_packed_k(U128 c, int len_rt, const u32 *lut)
{
    if constexpr (LEN >= 33 && LEN <= 64) return xxh3_packed_fixed<LEN, true>(c.lo, c.hi, lut);
    else return xxh3_lform(c, len_rt);
}

// ---- hash -> bit index (hash % _size, bloom_filter.hpp:84) --------------------
// size = odd * 2^shift.  x mod size = ((x >> shift) mod odd) << shift | (x & (2^shift - 1)).
struct ModDesc {
    u64 size;
    u64 odd;
    u32 shift;
    u32 kind; // 0: power of two, 1: (x >> shift) and odd fit 32 bits, 2: generic
};
__device__ __forceinline__ u64 mod_size(u64 h, const ModDesc &m)
{
    if (m.kind == 0) return h & (m.size - 1);
    if (m.kind == 1) {
        const u32 q = (u32)(h >> m.shift) % (u32)m.odd;
        return ((u64)q << m.shift) | (h & ((1ULL << m.shift) - 1));
    }
    return h % m.size;
}

// ---- Bloom filter view ---------------------------------------------------------
struct BFView {
    u64 *words;        // size bits
    const u32 *blk;    // ones before each 512-bit block (valid once finalised)
    u32 *counts;       // one wrapping u32 per set bit; the u16 cell of the reference is its low half
    // The gate: one cache-resident blocked Bloom filter in front of BOTH stores of the
    // call-time scan, keyed by the filter slot idx = XXH3 % size (the exact map is
    // addressed by the same XXH3, so one hash and one probe serve both).  Every set
    // bit of `bf` and every exact-map key sets gate_k bits inside ONE 64-bit word:
    // the word and the first bit from idx >> gate_shift, the others from the low
    // gate_shift bits of idx.  All of them are functions of idx alone, so a k-mer that
    // merely COLLIDES with a set bf bit (a Bloom false positive, which the reference
    // counts) passes too.  A closed gate proves that neither bf.increment nor
    // ref_bf.increment can do anything, so ~97 % of table rows finish after one
    // L2-resident probe.  Only the `bf` view carries it.
    u64 *gate;
    // When the index is too large for the gate to stay in L2 (> 4 MiB), a second, coarse gate of the L2-resident
    // size sits in front of it: same construction over the same idx, fewer bits per entry.  Only rows that pass
    // the coarse gate pay the (HBM / Infinity Cache) line of the fine one.
    u64 *pregate;
    u64 *pregate_fill; // the same array whenever it is allocated: inserts always fill it, `pregate` says whether probes use it
    // `context_bf` only, call time: the positions of its set bits as an open-addressing set (position + 1, 0 = free).  A
    // whole-genome context filter is 16 GiB of bits of which a few million are set: the scan's hit kernel asks the set
    // (tens of MB: no page of it ever leaves the TLBs) instead of a random word of the bit array (a page walk per row)
    const u64 *pos_set;
    u32 pos_set_log2;
    ModDesc mod;
    u32 gate_shift;
    u32 gate_k;   // bits per entry, 1..4
    u32 pre_shift;
    u32 pre_k;
    u32 use_gate;
};
__device__ __forceinline__ bool bf_bit(const BFView &b, u64 idx) { return (b.words[idx >> 6] >> (idx & 63)) & 1; }
// the same answer from the set of set positions, where the view carries one
__device__ __forceinline__ bool bf_bit_via_set(const BFView &b, u64 idx)
{
    if (!b.pos_set) return bf_bit(b, idx);
    const u64 mask = (1ULL << b.pos_set_log2) - 1;
    u64 s = (idx * 0x9E3779B97F4A7C15ULL) >> (64 - b.pos_set_log2);
    for (;;) {
        const u64 v = b.pos_set[s];
        if (v == idx + 1) return true;
        if (v == 0) return false;
        s = (s + 1) & mask;
    }
}
// low 32 bits of the product of the low 24 bits of a and c (full rate; the compiler keeps v_mul_lo_u32 for
// __umul24 when it cannot see the operand's width)
__device__ __forceinline__ u32 mul24(u32 a, u32 c)
{
    u32 r;
    asm("v_mul_u32_u24_e32 %0, %1, %2" : "=v"(r) : "s"(c), "v"(a));
    return r;
}
__device__ __forceinline__ u64 gate_mask_sk(u64 idx, u32 shift, u32 k)
{
    u64 m = 1ULL << ((idx >> shift) & 63);
    // the other positions: bits 26..31 of 24-bit products (v_mul_u32_u24 runs at full rate, v_mul_lo_u32 at a quarter) of
    // the low shift + 6 bits of idx -- everything in which two entries of the same gate word can differ.  (Taking only the
    // `shift` bits below the first position's six made the three extra positions a function of 6 bits at whole-genome
    // gates, shift = 6: 5.6 % of random rows passed a 13-bit-per-entry gate instead of ~1 %.)
    const u32 t = (u32)idx & (u32)((1ULL << (shift + 6 < 24 ? shift + 6 : 24)) - 1);
    if (k > 1) m |= 1ULL << (mul24(t, 0x9E3779u) >> 26);
    if (k > 2) m |= 1ULL << (mul24(t, 0xEBCA77u) >> 26);
    if (k > 3) m |= 1ULL << (mul24(t, 0xB2AE3Du) >> 26);
    return m;
}
__device__ __forceinline__ u64 gate_word(const BFView &b, u64 idx) { return idx >> (b.gate_shift + 6); }
__device__ __forceinline__ u64 gate_mask(const BFView &b, u64 idx) { return gate_mask_sk(idx, b.gate_shift, b.gate_k); }
__device__ __forceinline__ u64 pre_word(const BFView &b, u64 idx) { return idx >> (b.pre_shift + 6); }
__device__ __forceinline__ u64 pre_mask(const BFView &b, u64 idx) { return gate_mask_sk(idx, b.pre_shift, b.pre_k); }
__device__ __forceinline__ bool gate_open(const BFView &b, u64 idx)
{
    if (!b.use_gate) return true;
    if (b.pregate) {
        const u64 pm = pre_mask(b, idx);
        if ((b.pregate[pre_word(b, idx)] & pm) != pm) return false;
    }
    const u64 m = gate_mask(b, idx);
    return (b.gate[gate_word(b, idx)] & m) == m;
}
__device__ __forceinline__ void gate_set(const BFView &b, u64 idx)
{
    if (!b.gate) return;
    atomicOr((unsigned long long *)&b.gate[gate_word(b, idx)], gate_mask(b, idx));
    if (b.pregate_fill) atomicOr((unsigned long long *)&b.pregate_fill[pre_word(b, idx)], pre_mask(b, idx));
}
// rank(idx) = ones in [0, idx)   (rank_support_v<1>, bloom_filter.hpp:108)
// The 512-bit block that holds idx is one aligned 64-byte piece of `words` (the array is padded to whole blocks):
// it is read whole, four 16-byte loads issued together with the directory entry, and both the bit and the ones in
// front of it come out of registers -- one round trip instead of a word-by-word walk behind the bit test.
__device__ __forceinline__ bool bf_bit_rank(const BFView &b, u64 idx, u32 *rank)
{
    const u64 blk = idx >> 9;
    const u32 wi = (u32)(idx >> 6) & 7, bi = (u32)idx & 63;
    const uint4 *p = reinterpret_cast<const uint4 *>(b.words + blk * 8);
    const uint4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    u32 r = b.blk ? b.blk[blk] : 0;
    const u64 w[8] = {q0.x | (u64)q0.y << 32, q0.z | (u64)q0.w << 32, q1.x | (u64)q1.y << 32, q1.z | (u64)q1.w << 32,
                      q2.x | (u64)q2.y << 32, q2.z | (u64)q2.w << 32, q3.x | (u64)q3.y << 32, q3.z | (u64)q3.w << 32};
    bool bit = false;
#pragma unroll
    for (u32 j = 0; j < 8; ++j) {
        const u64 m = j < wi ? ~0ULL : j == wi ? (1ULL << bi) - 1 : 0ULL;
        r += (u32)__popcll(w[j] & m);
        if (j == wi) bit = (w[j] >> bi) & 1;
    }
    *rank = r;
    return bit;
}
__device__ __forceinline__ u32 bf_rank(const BFView &b, u64 idx)
{
    u32 r;
    bf_bit_rank(b, idx, &r);
    return r;
}
// counter of slot idx as the reference reads it (0 when the bit is clear)   bloom_filter.hpp:115-125
__device__ __forceinline__ u32 bf_count_at(const BFView &b, u64 idx)
{
    u32 r;
    return bf_bit_rank(b, idx, &r) ? b.counts[r] : 0;
}

// ---- exact map view --------------------------------------------------------------
// Open addressing, linear probing, power-of-two capacity.  tags: 0 empty, 1 being
// written, >= 2 fingerprint of a published key.  Keys are canonical L-forms.
// ids[slot] = smallest insertion row that carried the key; vals[id] is its counter,
// so the counter vector has the same layout on every GPU that replays the inserts.
// One slot = one 64-byte record, so a probe touches ONE cache line whether it ends at an empty slot, a foreign tag
// or the key itself (four parallel arrays cost up to four random 128-byte lines per hit).  The record also has
// room for two entries of the FILTER's directory: a set bit of `bf` (position + 1) and its rank, i.e. the index of
// its counter.  Both stores are addressed from the filter slot idx = XXH3 % size (map_home), so the scan's probe
// of a table row -- "is this k-mer a key of the exact map?" and "is bit idx of bf set, and which counter is it?" --
// reads one record instead of a filter word, a rank-directory entry and a map slot on three different lines.
// The bit array and its rank directory stay what the ABI's per-k-mer calls, export and the reference scan use.
struct __attribute__((aligned(64))) MapSlot {
    u32 tag; // 0 empty, 1 being written, >= 2 fingerprint of a published key
    u32 id;  // smallest insertion row that carried the key: index of its counter in vals[]
    u64 klo, khi;
    u32 brank[2]; // filter entries (written once `bf` is finalised): counter index of set bit bidx[j] - 1
    u64 bidx[2];  // 0 = free, else position of a set bit of `bf` + 1; slot 0 fills before slot 1
    // The record's own copy of its three counters, so that a call-time lookup ends on the line it started on (key ->
    // vals[id] and bit -> counts[rank] are each a second random line: four per biallelic SNP, two with the copies).
    // Each word carries the EPOCH it was last written in; a word of another epoch reads as zero, so resetting the
    // counters (every sample starts from zero) is one increment of the context's epoch, not a pass over the table.
    // vals[] / counts[] stay the counters of record (exchange between GPUs, export, the per-k-mer calls); the scan's
    // probe and hit kernels, which hold the record's line anyway, keep the copies equal to them (MapView::epoch != 0).
    unsigned long long cval; // epoch << 32 | the key's counter (wrapping u32, kmap.hpp:114-122)
    unsigned long long cbf;  // epoch << 32 | filter entry 1's counter << 16 | entry 0's (wrapping u16 each, bloom_filter.hpp:100-113)
};
static_assert(sizeof(MapSlot) == 64, "one record per 64 bytes");
struct MapView {
    MapSlot *slots;
    u32 *vals;
    u32 cap_log2;
    u32 klen;  // every key held here is exactly this long (other lengths live in the host overflow list)
    u32 epoch; // != 0: the records' counter copies of this epoch are current (single GPU, every increment since the reset made by the scan)
    u32 lazy;  // (epoch != 0 only) the scan adds to the records' copies ALONE: vals[] / counts[] catch up when somebody asks for them (rec_collect_kernel)
    u64 home_mul; // see map_home
};
// the scan's side of the copies: add `c` to the record's counter of the current epoch (a word of an older epoch restarts at zero)
__device__ __forceinline__ void rec_add_val(MapSlot *rec, u32 epoch, u32 c)
{
    unsigned long long old = rec->cval;
    for (;;) {
        const u32 cur = (u32)(old >> 32) == epoch ? (u32)old : 0u;
        const unsigned long long nw = (unsigned long long)epoch << 32 | (u32)(cur + c);
        const unsigned long long prev = atomicCAS(&rec->cval, old, nw);
        if (prev == old) return;
        old = prev;
    }
}
// (`old`: what the caller last read of the word -- the first compare-and-swap needs no load of its own)
__device__ __forceinline__ void rec_add_val_from(MapSlot *rec, u32 epoch, u32 c, unsigned long long old)
{
    for (;;) {
        const u32 cur = (u32)(old >> 32) == epoch ? (u32)old : 0u;
        const unsigned long long nw = (unsigned long long)epoch << 32 | (u32)(cur + c);
        const unsigned long long prev = atomicCAS(&rec->cval, old, nw);
        if (prev == old) return;
        old = prev;
    }
}
__device__ __forceinline__ void rec_add_bf_from(MapSlot *rec, int j, u32 epoch, u32 c, unsigned long long old)
{
    for (;;) {
        const u32 cur = (u32)(old >> 32) == epoch ? (u32)old : 0u;
        const u32 half = ((cur >> (16 * j)) + c) & 0xFFFFu;
        const unsigned long long nw = (unsigned long long)epoch << 32 | ((cur & ~(0xFFFFu << (16 * j))) | half << (16 * j));
        const unsigned long long prev = atomicCAS(&rec->cbf, old, nw);
        if (prev == old) return;
        old = prev;
    }
}
__device__ __forceinline__ void rec_add_bf(MapSlot *rec, int j, u32 epoch, u32 c)
{
    unsigned long long old = rec->cbf;
    for (;;) {
        const u32 cur = (u32)(old >> 32) == epoch ? (u32)old : 0u;
        const u32 half = ((cur >> (16 * j)) + c) & 0xFFFFu;
        const unsigned long long nw = (unsigned long long)epoch << 32 | ((cur & ~(0xFFFFu << (16 * j))) | half << (16 * j));
        const unsigned long long prev = atomicCAS(&rec->cbf, old, nw);
        if (prev == old) return;
        old = prev;
    }
}
// The table is addressed with the same XXH3 value the Bloom filter uses for the
// k-mer (one hash per table row serves both stores): home record from the filter slot
// idx = h % size (so that a bit of the filter, which knows only idx, lands in the same
// record as the keys that hash to it), tag from the middle of h.
__device__ __forceinline__ u32 map_tag(u64 h) { return (u32)(h >> 8) | 0x80000000u; }
// home_mul = floor((2^64 - 1) / size): the home record is idx scaled onto the table, so records lie IN ORDER OF THE FILTER SLOT.
// idx is already an XXH3 value reduced mod size -- as uniform as a second hash of it -- and the order buys locality where it
// decides everything: the ticket form of the scan files its rows by gate slice, i.e. by ranges of idx, so the rows a wave of
// the probe kernel holds land in one 1/128th of a 17-34 GB table (tens of pages) instead of on 64 pages of 16 thousand, and
// the kernel stops waiting for address translation (profiles/r03_pmc_c4share_before.txt).  (The multiplicative hash it
// replaces, home_mul = 0x9E3779B97F4A7C15, stays available as option map_ordered = 0.)
__device__ __forceinline__ u64 map_home(const MapView &m, u64 idx) { return (idx * m.home_mul) >> (64 - m.cap_log2); }
// Counter id of a published key (index into vals[]), or -1.  Read side only (every insert has completed: the host
// orders the kernels), so a record is read whole -- 16-byte loads issued together, one round trip -- instead of
// tag, then key low, then key high, then id, each waiting for the one before on the same line.
__device__ __forceinline__ long long map_find_id(const MapView &m, U128 key, u64 h, u64 idx)
{
    const u64 mask = (1ULL << m.cap_log2) - 1;
    u64 s = map_home(m, idx);
    const u32 tag = map_tag(h);
    for (;;) {
        const uint4 *p = reinterpret_cast<const uint4 *>(&m.slots[s]);
        const uint4 a = p[0], b = p[1]; // {tag, id, klo}, {khi, brank}
        if (a.x == 0) return -1;
        if (a.x == tag && a.z == (u32)key.lo && a.w == (u32)(key.lo >> 32) && b.x == (u32)key.hi && b.y == (u32)(key.hi >> 32)) return (long long)a.y;
        s = (s + 1) & mask;
    }
}
// Counter index (rank) of bit idx of the finalised filter, or -1 when the bit is clear.
__device__ __forceinline__ long long bucket_rank(const MapView &m, u64 idx)
{
    const u64 mask = (1ULL << m.cap_log2) - 1, want = idx + 1;
    u64 s = map_home(m, idx);
    for (;;) {
        const uint4 *p = reinterpret_cast<const uint4 *>(&m.slots[s]);
        const uint4 b = p[1], c = p[2]; // {khi, brank0, brank1}, {bidx0, bidx1}
        const u64 b0 = c.x | (u64)c.y << 32, b1 = c.z | (u64)c.w << 32;
        if (b0 == want) return (long long)b.z;
        if (b1 == want) return (long long)b.w;
        if (b0 == 0 || b1 == 0) return -1;
        s = (s + 1) & mask;
    }
}
// KMAP::get_count of a key (kmap.hpp:124-131): 0 when absent.  With the records' copies current the answer sits in the
// record (its fourth 16 bytes, requested with the first two); otherwise vals[id], a second random line.
__device__ __forceinline__ i32 map_value(const MapView &m, U128 key, u64 h, u64 idx)
{
    if (!m.epoch) {
        const long long id = map_find_id(m, key, h, idx);
        return id >= 0 ? (i32)m.vals[id] : 0;
    }
    const u64 mask = (1ULL << m.cap_log2) - 1;
    u64 s = map_home(m, idx);
    const u32 tag = map_tag(h);
    for (;;) {
        const uint4 *p = reinterpret_cast<const uint4 *>(&m.slots[s]);
        const uint4 a = p[0], b = p[1], d = p[3]; // {tag, id, klo}, {khi, brank}, {cval, cbf}
        if (a.x == 0) return 0;
        if (a.x == tag && a.z == (u32)key.lo && a.w == (u32)(key.lo >> 32) && b.x == (u32)key.hi && b.y == (u32)(key.hi >> 32))
            return d.y == m.epoch ? (i32)d.x : 0;
        s = (s + 1) & mask;
    }
}
// BF::get_count of filter slot idx as the genotyping reads it (bloom_filter.hpp:115-125: the u16 cell, 0 when the bit is clear)
__device__ __forceinline__ u32 bucket_count(const MapView &m, const u32 *counts, u64 idx)
{
    if (!m.epoch) {
        const long long rank = bucket_rank(m, idx);
        return rank >= 0 ? (u32)(uint16_t)counts[rank] : 0u;
    }
    const u64 mask = (1ULL << m.cap_log2) - 1, want = idx + 1;
    u64 s = map_home(m, idx);
    for (;;) {
        const uint4 *p = reinterpret_cast<const uint4 *>(&m.slots[s]);
        const uint4 c = p[2], d = p[3]; // {bidx0, bidx1}, {cval, cbf}
        const u64 b0 = c.x | (u64)c.y << 32, b1 = c.z | (u64)c.w << 32;
        const u32 both = d.w == m.epoch ? d.z : 0u;
        if (b0 == want) return both & 0xFFFFu;
        if (b1 == want) return both >> 16;
        if (b0 == 0 || b1 == 0) return 0u;
        s = (s + 1) & mask;
    }
}
// A wave's 64 home records, fetched by ROUNDS of four lanes per record (16 bytes each, three of the four) instead of three
// 16-byte loads per lane: 4 wave-wide load instructions touching 16 records each, against 3 touching 64.  At whole-genome
// scale the record table is 17-34 GB, every lane of a wave lands on another page, and each load instruction of each lane
// costs a translation request that misses the per-CU TLB (profiles/r03_pmc_c4share_before.txt: UTCL1 miss rate 0.65, UTCL2 busy
// 93 % of scan_probe_kernel's time -- the kernel was bound by address translation, not by HBM).  Every lane of the wave must
// call this; `s` = the lane's record, ignored where !active.  Returns the record's first 48 bytes.
__device__ __forceinline__ uint4 shfl_u4(uint4 v, int src)
{
    return uint4{(u32)__shfl((int)v.x, src, 64), (u32)__shfl((int)v.y, src, 64), (u32)__shfl((int)v.z, src, 64), (u32)__shfl((int)v.w, src, 64)};
}
__device__ __forceinline__ void records_load_coop(const MapView &m, u64 s, bool active, uint4 *a, uint4 *b, uint4 *c)
{
    const int lane = threadIdx.x & 63;
    const int part = lane & 3, q = lane >> 2;
    uint4 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { // all four rounds' loads are requested before any is used
        const int src = 16 * t + q; // the lane whose record this lane helps fetch in round t
        const u32 lo = (u32)__shfl((int)(u32)s, src, 64), hi = (u32)__shfl((int)(u32)(s >> 32), src, 64);
        const bool act = __shfl((int)active, src, 64) != 0;
        v[t] = uint4{0u, 0u, 0u, 0u};
        if (act && part < 3) v[t] = reinterpret_cast<const uint4 *>(&m.slots[(u64)lo | ((u64)hi << 32)])[part];
    }
    uint4 ra{0u, 0u, 0u, 0u}, rb = ra, rc = ra;
    const int mine = 4 * (lane & 15); // as an owner: my record's parts sit in lanes mine .. mine + 2 of round lane / 16
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint4 p0 = shfl_u4(v[t], mine), p1 = shfl_u4(v[t], mine + 1), p2 = shfl_u4(v[t], mine + 2);
        if ((lane >> 4) == t) {
            ra = p0;
            rb = p1;
            rc = p2;
        }
    }
    *a = ra;
    *b = rb;
    *c = rc;
}
// the same, whole records: the fourth 16 bytes (the record's counter copies, cval | cbf) come with the others -- the lane that sat idle
// in each group of four fetches them -- so that a scan kernel can compare-and-swap a copy without reading it first
__device__ __forceinline__ void records_load_coop4(const MapView &m, u64 s, bool active, uint4 *a, uint4 *b, uint4 *c, uint4 *d)
{
    const int lane = threadIdx.x & 63;
    const int part = lane & 3, q = lane >> 2;
    uint4 v[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) { // all four rounds' loads are requested before any is used
        const int src = 16 * t + q; // the lane whose record this lane helps fetch in round t
        const u32 lo = (u32)__shfl((int)(u32)s, src, 64), hi = (u32)__shfl((int)(u32)(s >> 32), src, 64);
        const bool act = __shfl((int)active, src, 64) != 0;
        v[t] = uint4{0u, 0u, 0u, 0u};
        if (act) v[t] = reinterpret_cast<const uint4 *>(&m.slots[(u64)lo | ((u64)hi << 32)])[part];
    }
    uint4 ra{0u, 0u, 0u, 0u}, rb = ra, rc = ra, rd = ra;
    const int mine = 4 * (lane & 15); // as an owner: my record's parts sit in lanes mine .. mine + 3 of round lane / 16
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint4 p0 = shfl_u4(v[t], mine), p1 = shfl_u4(v[t], mine + 1), p2 = shfl_u4(v[t], mine + 2), p3 = shfl_u4(v[t], mine + 3);
        if ((lane >> 4) == t) {
            ra = p0;
            rb = p1;
            rc = p2;
            rd = p3;
        }
    }
    *a = ra;
    *b = rb;
    *c = rc;
    *d = rd;
}
// bucket_probe with the home records fetched cooperatively (every lane of the wave calls it; `active` = the lane has a row)
// map_slot / bf_ent (may be NULL): the record that held the key, and record * 2 + entry of the filter bit -- where the scan keeps the counters' copies
__device__ __forceinline__ void bucket_probe_coop(const MapView &m, U128 key, u64 h, u64 idx, bool active, long long *map_id, long long *rank,
                                                  u64 *map_slot = nullptr, u64 *bf_ent = nullptr)
{
    const u64 mask = (1ULL << m.cap_log2) - 1, want = idx + 1;
    u64 s = map_home(m, idx);
    const u32 tag = map_tag(h);
    bool map_open = active, bf_open = active;
    *map_id = -1;
    *rank = -1;
    uint4 a, b, c;
    records_load_coop(m, s, active, &a, &b, &c);
    for (;;) {
        if (map_open) {
            if (a.x == 0) map_open = false;
            else if (a.x == tag && a.z == (u32)key.lo && a.w == (u32)(key.lo >> 32) && b.x == (u32)key.hi && b.y == (u32)(key.hi >> 32)) {
                *map_id = (long long)a.y;
                if (map_slot) *map_slot = s;
                map_open = false;
            }
        }
        if (bf_open) {
            const u64 b0 = c.x | (u64)c.y << 32, b1 = c.z | (u64)c.w << 32;
            if (b0 == want) {
                *rank = (long long)b.z;
                if (bf_ent) *bf_ent = 2 * s;
                bf_open = false;
            } else if (b1 == want) {
                *rank = (long long)b.w;
                if (bf_ent) *bf_ent = 2 * s + 1;
                bf_open = false;
            } else if (b0 == 0 || b1 == 0)
                bf_open = false;
        }
        if (!map_open && !bf_open) return;
        s = (s + 1) & mask; // (one lane in five goes on to the next record: nearly always the same page)
        const uint4 *p = reinterpret_cast<const uint4 *>(&m.slots[s]);
        a = p[0];
        b = p[1];
        c = p[2];
    }
}
// bucket_rank likewise
__device__ __forceinline__ long long bucket_rank_coop(const MapView &m, u64 idx, bool active, u64 *bf_ent = nullptr)
{
    const u64 mask = (1ULL << m.cap_log2) - 1, want = idx + 1;
    u64 s = map_home(m, idx);
    uint4 a, b, c;
    records_load_coop(m, s, active, &a, &b, &c);
    if (!active) return -1;
    for (;;) {
        const u64 b0 = c.x | (u64)c.y << 32, b1 = c.z | (u64)c.w << 32;
        if (b0 == want) {
            if (bf_ent) *bf_ent = 2 * s;
            return (long long)b.z;
        }
        if (b1 == want) {
            if (bf_ent) *bf_ent = 2 * s + 1;
            return (long long)b.w;
        }
        if (b0 == 0 || b1 == 0) return -1;
        s = (s + 1) & mask;
        const uint4 *p = reinterpret_cast<const uint4 *>(&m.slots[s]);
        b = p[1];
        c = p[2];
    }
}
// Both questions of the scan's probe in one walk (the two chains share their records).
__device__ __forceinline__ void bucket_probe(const MapView &m, U128 key, u64 h, u64 idx, long long *map_id, long long *rank)
{
    const u64 mask = (1ULL << m.cap_log2) - 1, want = idx + 1;
    u64 s = map_home(m, idx);
    const u32 tag = map_tag(h);
    bool map_open = true, bf_open = true;
    *map_id = -1;
    *rank = -1;
    for (;;) {
        const uint4 *p = reinterpret_cast<const uint4 *>(&m.slots[s]);
        const uint4 a = p[0], b = p[1], c = p[2];
        if (map_open) {
            if (a.x == 0) map_open = false;
            else if (a.x == tag && a.z == (u32)key.lo && a.w == (u32)(key.lo >> 32) && b.x == (u32)key.hi && b.y == (u32)(key.hi >> 32)) {
                *map_id = (long long)a.y;
                map_open = false;
            }
        }
        if (bf_open) {
            const u64 b0 = c.x | (u64)c.y << 32, b1 = c.z | (u64)c.w << 32;
            if (b0 == want) {
                *rank = (long long)b.z;
                bf_open = false;
            } else if (b1 == want) {
                *rank = (long long)b.w;
                bf_open = false;
            } else if (b0 == 0 || b1 == 0)
                bf_open = false;
        }
        if (!map_open && !bf_open) return;
        s = (s + 1) & mask;
    }
}

} // namespace mg
