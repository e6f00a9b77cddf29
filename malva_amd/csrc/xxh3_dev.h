// XXH3_64bits (seed 0, default secret) for gfx950, written for two input forms:
//   xxh3_bytes   -- any byte string of 0..128 bytes behind a byte accessor
//                   (reference text k-mers, which may hold N / NUL bytes)
//   xxh3_packed  -- a 2-bit packed ACGT string of 33..64 bases, expanded to its
//                   ASCII bytes in registers (the KMC stream)
// Follows the published algorithm as used by the reference:
//   XXH3_64bits            xxhash.h:5037 -> XXH3_64bits_internal :5011-5031
//   XXH3_len_0to16_64b     xxhash.h:3820-3907
//   XXH3_len_17to128_64b   xxhash.h:3946-3980,  XXH3_mix16B :3913-3944
//   XXH3_mul128_fold64     xxhash.h:3747,  XXH3_avalanche :3764,  secret :3548-3561
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mg {

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

// the default 192-byte secret read as little-endian u64 at every byte offset the
// <=128-byte paths touch (offsets 0..127 step 8 for mix16B, plus the short-input ones)
__device__ __constant__ const u8 kSecret[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
    0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8,
    0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d,
    0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64,
    0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb,
    0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce,
    0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e,
};

// the same bytes as a host/device compile-time table: sec64_at<OFF>() is the
// little-endian u64 at byte offset OFF, folded to an immediate
constexpr u8 kSecretC[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c,
    0xde, 0xd4, 0x6d, 0xe9, 0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f,
    0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78, 0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21,
    0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6, 0x81, 0x3a, 0x26, 0x4c,
    0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8,
    0xa8, 0xfa, 0x76, 0x3f, 0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d,
    0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31, 0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64,
    0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff, 0xfa, 0x13, 0x63, 0xeb,
    0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce,
    0x45, 0xcb, 0x3a, 0x8f, 0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e,
};
constexpr u64 sec64_c(int off)
{
    u64 v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | kSecretC[off + i];
    return v;
}
template <int OFF> struct Sec64 { static constexpr u64 value = sec64_c(OFF); };
template <int OFF> __host__ __device__ constexpr u64 sec64_at() { return Sec64<OFF>::value; }

constexpr u64 P64_1 = 0x9E3779B185EBCA87ULL;
constexpr u64 P64_2 = 0xC2B2AE3D27D4EB4FULL;
constexpr u64 P64_3 = 0x165667B19E3779F9ULL;
constexpr u64 P_MX1 = 0x165667919E3779F9ULL;
constexpr u64 P_MX2 = 0x9FB21C651E98DF25ULL;

// 64x64 -> 128 multiply folded to lo ^ hi, as four 32x32+64 multiply-adds
// (v_mad_u64_u32) so no partial product is computed twice.
__device__ __forceinline__ u64 mul128_fold64(u64 a, u64 b)
{
    const u32 a0 = (u32)a, a1 = (u32)(a >> 32), b0 = (u32)b, b1 = (u32)(b >> 32);
    const u64 t0 = (u64)a0 * b0;
    const u64 t1 = (u64)a0 * b1 + (t0 >> 32);
    const u64 t2 = (u64)a1 * b0 + (u32)t1;
    const u64 hi = (u64)a1 * b1 + (t1 >> 32) + (t2 >> 32);
    const u64 lo = (t2 << 32) | (u32)t0;
    return lo ^ hi;
}
__device__ __forceinline__ u64 xxh3_avalanche(u64 h)
{
    h ^= h >> 37;
    h *= P_MX1;
    h ^= h >> 32;
    return h;
}
__device__ __forceinline__ u64 xxh64_avalanche(u64 h)
{
    h ^= h >> 33;
    h *= P64_2;
    h ^= h >> 29;
    h *= P64_3;
    h ^= h >> 32;
    return h;
}
__device__ __forceinline__ u64 rotl64(u64 x, int r) { return (x << r) | (x >> (64 - r)); }
__device__ __forceinline__ u64 bswap64(u64 x) { return __builtin_bswap64(x); }

__device__ __forceinline__ u64 sec64_rt(int off)
{
    u64 v = 0;
#pragma unroll
    for (int i = 7; i >= 0; --i) v = (v << 8) | kSecret[off + i];
    return v;
}

// ---- byte-string form ------------------------------------------------------
// GET(i) returns byte i of the (already canonical) string.
template <class GET> __device__ __forceinline__ u64 rd64(GET get, int off)
{
    u64 v = 0;
#pragma unroll
    for (int i = 7; i >= 0; --i) v = (v << 8) | (u64)get(off + i);
    return v;
}
template <class GET> __device__ __forceinline__ u32 rd32(GET get, int off)
{
    u32 v = 0;
#pragma unroll
    for (int i = 3; i >= 0; --i) v = (v << 8) | (u32)get(off + i);
    return v;
}

template <class GET> __device__ u64 xxh3_bytes(GET get, int len)
{
    if (len > 16) { // 17..128
        u64 acc = (u64)len * P64_1;
        int i = (len - 1) / 32;
        do {
            acc += mul128_fold64(rd64(get, 16 * i) ^ sec64_rt(32 * i), rd64(get, 16 * i + 8) ^ sec64_rt(32 * i + 8));
            acc += mul128_fold64(rd64(get, len - 16 * (i + 1)) ^ sec64_rt(32 * i + 16),
                                 rd64(get, len - 16 * (i + 1) + 8) ^ sec64_rt(32 * i + 24));
        } while (i-- != 0);
        return xxh3_avalanche(acc);
    }
    if (len > 8) {
        const u64 lo = rd64(get, 0) ^ (sec64_rt(24) ^ sec64_rt(32));
        const u64 hi = rd64(get, len - 8) ^ (sec64_rt(40) ^ sec64_rt(48));
        const u64 acc = (u64)len + bswap64(lo) + hi + mul128_fold64(lo, hi);
        return xxh3_avalanche(acc);
    }
    if (len >= 4) {
        const u32 i1 = rd32(get, 0), i2 = rd32(get, len - 4);
        const u64 bitflip = sec64_rt(8) ^ sec64_rt(16);
        u64 h = ((u64)i2 + ((u64)i1 << 32)) ^ bitflip;
        h ^= rotl64(h, 49) ^ rotl64(h, 24);
        h *= P_MX2;
        h ^= (h >> 35) + (u64)len;
        h *= P_MX2;
        return h ^ (h >> 28);
    }
    if (len > 0) {
        const u32 c1 = get(0), c2 = get(len >> 1), c3 = get(len - 1);
        const u32 combined = (c1 << 16) | (c2 << 24) | c3 | ((u32)len << 8);
        const u64 bitflip = (u64)((u32)sec64_rt(0) ^ (u32)(sec64_rt(0) >> 32));
        return xxh64_avalanche((u64)combined ^ bitflip);
    }
    return xxh64_avalanche(sec64_rt(56) ^ sec64_rt(64));
}

// ---- packed form -----------------------------------------------------------
// 4 two-bit codes (8 bits, base j at bits 2j) -> 4 ASCII bytes (base j in byte j).
// Spread the codes to one per byte, then let v_perm_b32 pick 'A','C','G','T'
// from the LUT word by selector value.
__device__ __forceinline__ u32 expand4(u32 c8)
{
    u32 t = (c8 | (c8 << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    return __builtin_amdgcn_perm(0u, 0x54474341u, t);
}
// bases [o, o+8) of the LSB-first packed string lo:hi (base i at bits 2i) as the
// little-endian u64 of their ASCII bytes
__device__ __forceinline__ u64 ascii8(u64 lo, u64 hi, int o)
{
    const int sh = 2 * o;
    u32 c16;
    if (sh == 0) c16 = (u32)lo;
    else if (sh < 64) c16 = (u32)((lo >> sh) | (sh > 48 ? (hi << (64 - sh)) : 0));
    else c16 = (u32)(hi >> (sh - 64));
    return (u64)expand4(c16 & 0xFF) | ((u64)expand4((c16 >> 8) & 0xFF) << 32);
}

// XXH3 of the ASCII rendering of a packed string of LEN bases, 33 <= LEN <= 64.
// With LEN a compile-time constant every shift and secret word is an immediate.
__device__ __forceinline__ u64 xxh3_packed_33to64(u64 lo, u64 hi, int len)
{
    u64 acc = (u64)len * P64_1;
    acc += mul128_fold64(ascii8(lo, hi, 16) ^ sec64_at<32>(), ascii8(lo, hi, 24) ^ sec64_at<40>());
    acc += mul128_fold64(ascii8(lo, hi, len - 32) ^ sec64_at<48>(), ascii8(lo, hi, len - 24) ^ sec64_at<56>());
    acc += mul128_fold64(ascii8(lo, hi, 0) ^ sec64_at<0>(), ascii8(lo, hi, 8) ^ sec64_at<8>());
    acc += mul128_fold64(ascii8(lo, hi, len - 16) ^ sec64_at<16>(), ascii8(lo, hi, len - 8) ^ sec64_at<24>());
    return xxh3_avalanche(acc);
}
// The same for a compile-time length 33..64: the ASCII rendering is expanded ONCE into
// ceil(LEN/4) dwords (one v_perm_b32 each) and the eight overlapping 8-byte reads of
// XXH3's 33..64-byte path are cut out of them with v_alignbyte_b32, instead of expanding
// sixteen dwords (the two read sets overlap almost entirely).
// LUT == true: `lut` is a 256-entry table in LDS, lut[c8] = expand4(c8) (ascii_lut_fill).  The table-streaming
// kernels are bound by VALU issue, not by HBM: an expansion costs four VALU operations per dword in arithmetic
// and one (the byte select) plus an LDS read, which issues beside the VALU, through the table.
__device__ __forceinline__ void ascii_lut_fill(u32 *lut) // every thread of a >= 256-thread workgroup; barrier after
{
    if (threadIdx.x < 256) lut[threadIdx.x] = expand4(threadIdx.x);
}
template <int LEN, bool LUT = false> __device__ __forceinline__ u64 xxh3_packed_fixed(u64 lo, u64 hi, const u32 *lut = nullptr)
{
    constexpr int NW = (LEN + 3) / 4;
    u32 w[NW + 1];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
        const u32 c8 = (u32)((j < 8 ? lo >> (8 * j) : hi >> (8 * (j - 8))) & 0xFF);
        w[j] = LUT ? lut[c8] : expand4(c8);
    }
    w[NW] = 0;
    auto rd = [&](int o) -> u64 { // little-endian u64 at byte offset o of the ASCII string
        const int i = o >> 2, s = o & 3;
        const u32 a = s ? __builtin_amdgcn_alignbyte(w[i + 1], w[i], (u32)s) : w[i];
        const u32 b = s ? __builtin_amdgcn_alignbyte(w[i + 2 <= NW ? i + 2 : NW], w[i + 1], (u32)s) : w[i + 1];
        return (u64)a | ((u64)b << 32);
    };
    u64 acc = (u64)LEN * P64_1;
    acc += mul128_fold64(rd(16) ^ sec64_at<32>(), rd(24) ^ sec64_at<40>());
    acc += mul128_fold64(rd(LEN - 32) ^ sec64_at<48>(), rd(LEN - 24) ^ sec64_at<56>());
    acc += mul128_fold64(rd(0) ^ sec64_at<0>(), rd(8) ^ sec64_at<8>());
    acc += mul128_fold64(rd(LEN - 16) ^ sec64_at<16>(), rd(LEN - 8) ^ sec64_at<24>());
    return xxh3_avalanche(acc);
}
// 17 <= len <= 32
__device__ __forceinline__ u64 xxh3_packed_17to32(u64 lo, u64 hi, int len)
{
    u64 acc = (u64)len * P64_1;
    acc += mul128_fold64(ascii8(lo, hi, 0) ^ sec64_at<0>(), ascii8(lo, hi, 8) ^ sec64_at<8>());
    acc += mul128_fold64(ascii8(lo, hi, len - 16) ^ sec64_at<16>(), ascii8(lo, hi, len - 8) ^ sec64_at<24>());
    return xxh3_avalanche(acc);
}

} // namespace mg
