/*
 * malva_oracle.c -- CPU restatement of malva-geno's k-mer matching and
 * genotype-likelihood path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the *checker* for the HIP path in malva_amd/csrc.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The
 * product (libmalva_hip.so, malva-geno) never links or calls anything in here
 * and has no CPU fallback.
 *
 * Every function restates, in plain C, the behaviour of the reference at the
 * cited file:line (paths relative to the reference checkout).  Nothing is
 * copied: the reference is C++ over sdsl/std containers, this is flat C over
 * arrays.  Parity pins: see oracle/README.md (XXH3 against the reference's own
 * vendored xxhash.c built into oracle/_ref, the survey's known-answer vectors,
 * and the reference's haploid golden VCF end to end).
 *
 * Build: gcc -O2 -ffp-contract=off -fPIC -shared -o libmalva_oracle.so malva_oracle.c -lm
 * (-ffp-contract=off: the genotype arithmetic below must round each product
 * exactly where the reference's x86-64 build does.)
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------- */
/* XXH3_64bits, seed 0, default secret: xxhash.h:5037-5040 -> :5011-5031.     */
/* ------------------------------------------------------------------------- */

/* xxhash.h:3548-3561 (public constant of the algorithm, from FARSH) */
static const uint8_t kSecret[192] = {
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

#define P64_1 0x9E3779B185EBCA87ULL /* xxhash.h:2684 */
#define P64_2 0xC2B2AE3D27D4EB4FULL /* xxhash.h:2685 */
#define P64_3 0x165667B19E3779F9ULL /* xxhash.h:2686 */
#define P_MX1 0x165667919E3779F9ULL /* xxhash.h:3767 (XXH3_avalanche multiplier) */
#define P_MX2 0x9FB21C651E98DF25ULL /* xxhash.h:3781 (rrmxmx multiplier) */

static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; } /* little-endian host */
static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rotl64(uint64_t x, int r) { return (x << r) | (x >> (64 - r)); }
static inline uint64_t bswap64(uint64_t x) { return __builtin_bswap64(x); }

/* xxhash.h:3747-3751: low64(product) ^ high64(product) */
static inline uint64_t mul128_fold64(uint64_t a, uint64_t b)
{
    unsigned __int128 p = (unsigned __int128)a * b;
    return (uint64_t)p ^ (uint64_t)(p >> 64);
}
/* xxhash.h:3764-3770 */
static inline uint64_t xxh3_avalanche(uint64_t h)
{
    h ^= h >> 37; h *= P_MX1; h ^= h >> 32; return h;
}
/* xxhash.h:2716-2724 */
static inline uint64_t xxh64_avalanche(uint64_t h)
{
    h ^= h >> 33; h *= P64_2; h ^= h >> 29; h *= P64_3; h ^= h >> 32; return h;
}
/* xxhash.h:3913-3944, seed 0 */
static inline uint64_t mix16B(const uint8_t *in, const uint8_t *sec)
{
    return mul128_fold64(rd64(in) ^ rd64(sec), rd64(in + 8) ^ rd64(sec + 8));
}

MO_API uint64_t mo_xxh3_64(const void *data, size_t len)
{
    const uint8_t *in = (const uint8_t *)data;
    const uint8_t *s = kSecret;
    if (len == 0) /* xxhash.h:3905 */
        return xxh64_avalanche(rd64(s + 56) ^ rd64(s + 64));
    if (len <= 3) { /* xxhash.h:3820-3841 */
        uint32_t c1 = in[0], c2 = in[len >> 1], c3 = in[len - 1];
        uint32_t combined = (c1 << 16) | (c2 << 24) | c3 | ((uint32_t)len << 8);
        uint64_t bitflip = (uint64_t)(rd32(s) ^ rd32(s + 4));
        return xxh64_avalanche((uint64_t)combined ^ bitflip);
    }
    if (len <= 8) { /* xxhash.h:3843-3858, rrmxmx :3777-3785 */
        uint32_t i1 = rd32(in), i2 = rd32(in + len - 4);
        uint64_t bitflip = rd64(s + 8) ^ rd64(s + 16);
        uint64_t h = ((uint64_t)i2 + ((uint64_t)i1 << 32)) ^ bitflip;
        h ^= rotl64(h, 49) ^ rotl64(h, 24);
        h *= P_MX2;
        h ^= (h >> 35) + len;
        h *= P_MX2;
        return h ^ (h >> 28);
    }
    if (len <= 16) { /* xxhash.h:3860-3876 */
        uint64_t lo = rd64(in) ^ (rd64(s + 24) ^ rd64(s + 32));
        uint64_t hi = rd64(in + len - 8) ^ (rd64(s + 40) ^ rd64(s + 48));
        uint64_t acc = len + bswap64(lo) + hi + mul128_fold64(lo, hi);
        return xxh3_avalanche(acc);
    }
    if (len <= 128) { /* xxhash.h:3946-3980 */
        uint64_t acc = len * P64_1;
        size_t i = (len - 1) / 32;
        do {
            acc += mix16B(in + 16 * i, s + 32 * i);
            acc += mix16B(in + len - 16 * (i + 1), s + 32 * i + 16);
        } while (i-- != 0);
        return xxh3_avalanche(acc);
    }
    if (len <= 240) { /* xxhash.h:3984-4038 */
        uint64_t acc = len * P64_1;
        int nb = (int)len / 16, i;
        for (i = 0; i < 8; i++) acc += mix16B(in + 16 * i, s + 16 * i);
        acc = xxh3_avalanche(acc);
        for (i = 8; i < nb; i++) acc += mix16B(in + 16 * i, s + 16 * (i - 8) + 3);
        acc += mix16B(in + len - 16, s + 136 - 17);
        return xxh3_avalanche(acc);
    }
    fprintf(stderr, "mo_xxh3_64: len %zu > 240 is outside the path (k-mers are <= 240 bytes)\n", len);
    abort();
}

/* ------------------------------------------------------------------------- */
/* RCN complement table and canonical form: bloom_filter.hpp:36-50, 58-65.    */
/* ------------------------------------------------------------------------- */

static inline unsigned char rcn(unsigned char c)
{
    switch (c) { /* every byte not listed maps to 0 (bloom_filter.hpp:36-50) */
    case 'A': return 'T'; case 'C': return 'G'; case 'G': return 'C'; case 'T': return 'A';
    case 'N': return 'N';
    case 'a': return 'T'; case 'c': return 'G'; case 'g': return 'G'; /* sic: index 103 holds 'G' */
    case 't': return 'A'; case 'n': return 'N';
    default: return 0;
    }
}

/* out receives k bytes + NUL.  kmer must hold k non-NUL bytes.
 * strcmp(kmer, rc) < 0 ? kmer : rc  -- rc may contain NULs, which end strcmp's
 * walk with kmer's byte > 0, i.e. rc is kept (bloom_filter.hpp:63-64). */
MO_API void mo_canonical(const char *kmer, int k, char *out)
{
    int i;
    for (i = 0; i < k; i++) out[i] = (char)rcn((unsigned char)kmer[k - 1 - i]);
    out[k] = 0;
    if (strcmp(kmer, out) < 0) memmove(out, kmer, (size_t)k);
}

/* ------------------------------------------------------------------------- */
/* BF: bloom_filter.hpp:52-157.  bit_vector + rank_support_v<1> + int_vector<16>
 * restated as u64 words + 512-bit-blocked prefix popcounts + u16 cells.      */
/* ------------------------------------------------------------------------- */

typedef struct mo_bf {
    int mode;          /* 0 = write, 1 = read (bloom_filter.hpp:152) */
    uint64_t size;     /* bits */
    uint64_t nwords;
    uint64_t *words;
    uint64_t *blk;     /* ones before each 512-bit block (rank directory) */
    uint64_t nset;
    uint16_t *counts;
} mo_bf;

MO_API mo_bf *mo_bf_new(uint64_t size_bits)
{
    mo_bf *b = (mo_bf *)calloc(1, sizeof(mo_bf));
    b->size = size_bits;
    b->nwords = (size_bits + 63) / 64;
    b->words = (uint64_t *)calloc(b->nwords ? b->nwords : 1, 8);
    if (!b->words) { free(b); return NULL; }
    return b;
}
MO_API void mo_bf_free(mo_bf *b)
{
    if (!b) return;
    free(b->words); free(b->blk); free(b->counts); free(b);
}

/* bloom_filter.hpp:67-74: strlen, canonical, XXH3 over k bytes of the canonical buffer */
static uint64_t bf_hash(const char *kmer)
{
    size_t k = strlen(kmer);
    char ck[k + 1];
    mo_canonical(kmer, (int)k, ck);
    return mo_xxh3_64(ck, k);
}
MO_API uint64_t mo_bf_hash(const char *kmer) { return bf_hash(kmer); }

static inline int bf_bit(const mo_bf *b, uint64_t i) { return (int)((b->words[i >> 6] >> (i & 63)) & 1); }
/* rank(i) = ones in [0, i); i == size allowed (bloom_filter.hpp:97) */
static uint64_t bf_rank(const mo_bf *b, uint64_t i)
{
    uint64_t blk = i >> 9, w = blk * 8, r = b->blk[blk], we = i >> 6;
    for (; w < we; w++) r += (uint64_t)__builtin_popcountll(b->words[w]);
    if (i & 63) r += (uint64_t)__builtin_popcountll(b->words[we] & ((1ULL << (i & 63)) - 1));
    return r;
}

MO_API void mo_bf_add_key(mo_bf *b, const char *kmer) /* bloom_filter.hpp:81-85 */
{
    uint64_t i = bf_hash(kmer) % b->size;
    b->words[i >> 6] |= 1ULL << (i & 63);
}
MO_API int mo_bf_test_key(const mo_bf *b, const char *kmer) /* bloom_filter.hpp:87-91 */
{
    return bf_bit(b, bf_hash(kmer) % b->size);
}
MO_API void mo_bf_switch_mode(mo_bf *b) /* bloom_filter.hpp:93-98 */
{
    uint64_t nblk = (b->nwords + 7) / 8 + 1, w, acc = 0;
    b->mode = 1;
    free(b->blk); free(b->counts);
    b->blk = (uint64_t *)malloc(nblk * 8);
    for (w = 0; w < b->nwords; w++) {
        if ((w & 7) == 0) b->blk[w >> 3] = acc;
        acc += (uint64_t)__builtin_popcountll(b->words[w]);
    }
    for (w = (b->nwords + 7) / 8; w < nblk; w++) b->blk[w] = acc;
    b->nset = acc;
    b->counts = (uint16_t *)calloc(acc ? acc : 1, 2);
}
MO_API int mo_bf_increment(mo_bf *b, const char *kmer, uint32_t counter) /* bloom_filter.hpp:100-113 */
{
    uint64_t i;
    if (!b->mode) return 0;
    i = bf_hash(kmer) % b->size;
    if (bf_bit(b, i)) {
        uint64_t c = bf_rank(b, i);
        uint32_t nv = (uint32_t)b->counts[c] + counter;
        b->counts[c] = (uint16_t)nv; /* int_vector<16> cell: keeps the low 16 bits */
    }
    return 1;
}
MO_API uint16_t mo_bf_get_count(const mo_bf *b, const char *kmer) /* bloom_filter.hpp:115-125 */
{
    if (b->mode) {
        uint64_t i = bf_hash(kmer) % b->size;
        if (bf_bit(b, i)) return b->counts[bf_rank(b, i)];
    }
    return 0;
}
/* inspection (parity against the device arrays) */
MO_API uint64_t mo_bf_size(const mo_bf *b) { return b->size; }
MO_API uint64_t mo_bf_nwords(const mo_bf *b) { return b->nwords; }
MO_API const uint64_t *mo_bf_words(const mo_bf *b) { return b->words; }
MO_API uint64_t mo_bf_nset(const mo_bf *b) { return b->nset; }
MO_API const uint16_t *mo_bf_counts(const mo_bf *b) { return b->counts; }
MO_API uint64_t mo_bf_popcount(const mo_bf *b)
{
    uint64_t w, acc = 0;
    for (w = 0; w < b->nwords; w++) acc += (uint64_t)__builtin_popcountll(b->words[w]);
    return acc;
}
/* positions of the set bits in ascending order (== counter index order) */
MO_API uint64_t mo_bf_set_positions(const mo_bf *b, uint64_t *out, uint64_t cap)
{
    uint64_t w, n = 0;
    for (w = 0; w < b->nwords; w++) {
        uint64_t x = b->words[w];
        while (x) {
            if (n < cap) out[n] = w * 64 + (uint64_t)__builtin_ctzll(x);
            n++; x &= x - 1;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------- */
/* KMAP: kmap.hpp:46-132.  unordered_map<string,int> restated as an open-
 * addressing table over byte strings.  Keys are the canonical string
 * *truncated at its first NUL* (kmap.hpp:95: std::string(const char*)).      */
/* ------------------------------------------------------------------------- */

typedef struct mo_kmap {
    uint64_t cap;      /* power of two */
    uint64_t n;
    int64_t *slot;     /* entry index or -1 */
    uint64_t ecap;
    uint64_t *ehash;
    uint64_t *eoff;
    uint32_t *elen;
    int32_t *eval;
    char *pool;
    uint64_t plen, pcap;
} mo_kmap;

static uint64_t km_strhash(const char *s, size_t n)
{
    uint64_t h = 0xcbf29ce484222325ULL; size_t i;
    for (i = 0; i < n; i++) { h ^= (unsigned char)s[i]; h *= 0x100000001b3ULL; }
    h ^= h >> 29; h *= 0xbf58476d1ce4e5b9ULL; h ^= h >> 32;
    return h;
}
MO_API mo_kmap *mo_kmap_new(void)
{
    mo_kmap *m = (mo_kmap *)calloc(1, sizeof(mo_kmap));
    uint64_t i;
    m->cap = 1024; m->slot = (int64_t *)malloc(m->cap * 8);
    for (i = 0; i < m->cap; i++) m->slot[i] = -1;
    m->ecap = 512;
    m->ehash = (uint64_t *)malloc(m->ecap * 8); m->eoff = (uint64_t *)malloc(m->ecap * 8);
    m->elen = (uint32_t *)malloc(m->ecap * 4); m->eval = (int32_t *)malloc(m->ecap * 4);
    m->pcap = 1 << 16; m->pool = (char *)malloc(m->pcap);
    return m;
}
MO_API void mo_kmap_free(mo_kmap *m)
{
    if (!m) return;
    free(m->slot); free(m->ehash); free(m->eoff); free(m->elen); free(m->eval); free(m->pool); free(m);
}
static int64_t km_find(const mo_kmap *m, const char *s, size_t n, uint64_t h)
{
    uint64_t i = h & (m->cap - 1);
    for (;;) {
        int64_t e = m->slot[i];
        if (e < 0) return -1;
        if (m->ehash[e] == h && m->elen[e] == n && memcmp(m->pool + m->eoff[e], s, n) == 0) return e;
        i = (i + 1) & (m->cap - 1);
    }
}
static void km_grow(mo_kmap *m)
{
    uint64_t ncap = m->cap * 2, i, e;
    free(m->slot);
    m->slot = (int64_t *)malloc(ncap * 8);
    for (i = 0; i < ncap; i++) m->slot[i] = -1;
    m->cap = ncap;
    for (e = 0; e < m->n; e++) {
        i = m->ehash[e] & (ncap - 1);
        while (m->slot[i] >= 0) i = (i + 1) & (ncap - 1);
        m->slot[i] = (int64_t)e;
    }
}
/* kmap.hpp:86-97: canonical, then std::string(ckmer) => cut at first NUL */
static size_t km_canon(const char *kmer, char *ck)
{
    size_t k = strlen(kmer);
    mo_canonical(kmer, (int)k, ck);
    return strlen(ck);
}
MO_API void mo_kmap_add_key(mo_kmap *m, const char *kmer) /* kmap.hpp:108-112: kmers[ckmer] = 0 */
{
    size_t k = strlen(kmer);
    char ck[k + 1];
    size_t n = km_canon(kmer, ck);
    uint64_t h = km_strhash(ck, n), i;
    int64_t e = km_find(m, ck, n, h);
    if (e >= 0) { m->eval[e] = 0; return; }
    if ((m->n + 1) * 2 > m->cap) km_grow(m);
    if (m->n == m->ecap) {
        m->ecap *= 2;
        m->ehash = (uint64_t *)realloc(m->ehash, m->ecap * 8); m->eoff = (uint64_t *)realloc(m->eoff, m->ecap * 8);
        m->elen = (uint32_t *)realloc(m->elen, m->ecap * 4); m->eval = (int32_t *)realloc(m->eval, m->ecap * 4);
    }
    while (m->plen + n + 1 > m->pcap) { m->pcap *= 2; m->pool = (char *)realloc(m->pool, m->pcap); }
    memcpy(m->pool + m->plen, ck, n); m->pool[m->plen + n] = 0;
    e = (int64_t)m->n++;
    m->ehash[e] = h; m->eoff[e] = m->plen; m->elen[e] = (uint32_t)n; m->eval[e] = 0;
    m->plen += n + 1;
    i = h & (m->cap - 1);
    while (m->slot[i] >= 0) i = (i + 1) & (m->cap - 1);
    m->slot[i] = e;
}
MO_API int mo_kmap_test_key(const mo_kmap *m, const char *kmer) /* kmap.hpp:99-106 */
{
    size_t k = strlen(kmer);
    char ck[k + 1];
    size_t n = km_canon(kmer, ck);
    return km_find(m, ck, n, km_strhash(ck, n)) >= 0;
}
MO_API void mo_kmap_increment(mo_kmap *m, const char *kmer, int counter) /* kmap.hpp:114-122 */
{
    size_t k = strlen(kmer);
    char ck[k + 1];
    size_t n = km_canon(kmer, ck);
    int64_t e = km_find(m, ck, n, km_strhash(ck, n));
    if (e >= 0) {
        uint32_t nv = (uint32_t)m->eval[e] + (uint32_t)counter; /* uint32 new_value = kmers[ckmer] + counter */
        m->eval[e] = (int32_t)nv;
    }
}
MO_API int mo_kmap_get_count(const mo_kmap *m, const char *kmer) /* kmap.hpp:124-131 */
{
    size_t k = strlen(kmer);
    char ck[k + 1];
    size_t n = km_canon(kmer, ck);
    int64_t e = km_find(m, ck, n, km_strhash(ck, n));
    return e >= 0 ? m->eval[e] : 0;
}
MO_API uint64_t mo_kmap_size(const mo_kmap *m) { return m->n; }
/* entry e (insertion order): key bytes, length, value */
MO_API const char *mo_kmap_entry(const mo_kmap *m, uint64_t e, uint32_t *len, int32_t *val)
{
    *len = m->elen[e]; *val = m->eval[e];
    return m->pool + m->eoff[e];
}

/* ------------------------------------------------------------------------- */
/* Batched drivers over fixed-stride, NUL-terminated ASCII rows.              */
/* ------------------------------------------------------------------------- */

/* add_kmers_to_bf body, main.cpp:122-144: allele 0 -> KMAP.add_key, others -> BF.add_key */
MO_API void mo_add_kmers(mo_bf *bf, mo_kmap *ref_bf, const char *rows, size_t stride, size_t n,
                         const uint8_t *is_ref)
{
    size_t i;
    for (i = 0; i < n; i++) {
        if (is_ref[i]) mo_kmap_add_key(ref_bf, rows + i * stride);
        else mo_bf_add_key(bf, rows + i * stride);
    }
}

/* KMC scan loop body, main.cpp:488-499, over ASCII contexts.  The caller
 * supplies what CKmerAPI::to_string would (one ref_k-mer per row). */
static inline void scan_one(mo_bf *context_bf, mo_bf *bf, mo_kmap *ref_bf, char *context, uint32_t counter,
                            int k, int ref_k)
{
    int i;
    char kmer[k + 1];
    for (i = 0; i < ref_k; i++) /* main.cpp:491 toupper */
        if (context[i] >= 'a' && context[i] <= 'z') context[i] = (char)(context[i] - 32);
    strncpy(kmer, context + ((ref_k - k) / 2), (size_t)k); /* main.cpp:493 */
    kmer[k] = 0;
    mo_kmap_increment(ref_bf, kmer, (int)counter);          /* main.cpp:495 */
    if (!mo_bf_test_key(context_bf, context))               /* main.cpp:496 */
        mo_bf_increment(bf, kmer, counter);                 /* main.cpp:498 */
}
MO_API void mo_kmc_scan(mo_bf *context_bf, mo_bf *bf, mo_kmap *ref_bf, const char *rows, size_t stride,
                        const uint32_t *counts, size_t n, int k, int ref_k)
{
    size_t i;
    char ctx[ref_k + 1];
    for (i = 0; i < n; i++) {
        memcpy(ctx, rows + i * stride, (size_t)ref_k); ctx[ref_k] = 0;
        scan_one(context_bf, bf, ref_bf, ctx, counts[i], k, ref_k);
    }
}
/* Same loop fed from the 2-bit packed table the device consumes (SoA hi/lo,
 * MSB-first, right-aligned: base i of the r-mer sits at bits 2(r-1-i)+1..2(r-1-i)
 * of the 128-bit value hi:lo).  Unpacking stands in for CKmerAPI::to_string
 * (main.cpp:490), which is part of the reference's per-k-mer cost. */
MO_API void mo_kmc_scan_packed(mo_bf *context_bf, mo_bf *bf, mo_kmap *ref_bf, const uint64_t *hi,
                               const uint64_t *lo, const uint32_t *counts, size_t n, int k, int ref_k)
{
    static const char L[4] = {'A', 'C', 'G', 'T'};
    size_t i;
    char ctx[ref_k + 1];
    for (i = 0; i < n; i++) {
        int j;
        for (j = 0; j < ref_k; j++) {
            int sh = 2 * (ref_k - 1 - j);
            uint64_t c = sh >= 64 ? (hi[i] >> (sh - 64)) : (lo[i] >> sh);
            ctx[j] = L[c & 3];
        }
        ctx[ref_k] = 0;
        scan_one(context_bf, bf, ref_bf, ctx, counts[i], k, ref_k);
    }
}

/* The same loop run by T threads over T contiguous slices of the table: the "whole box" CPU figure of
 * SURVEY 8(d)(ii), next to the single-threaded one that mirrors the reference.  Both counter updates are wrapping
 * sums (Appendix A.2), so they commute; the threads share the read-only bits / rank / keys and add atomically.
 * Results are identical to mo_kmc_scan_packed (tests/test_oracle_pins.py). */
/* ------------------------------------------------------------------------- */
/* VB::are_near, var_block.hpp:417-423.                                       */
/* The reference writes                                                       */
/*   v1.ref_pos + v1.ref_size - v1.min_size - 1 + sum_to_add                  */
/*       + ceil((float)k / 2) >= v2.ref_pos                                   */
/* with `using namespace std`: ceil(float) is the float overload, so the int   */
/* sum on its left is converted to FLOAT, the addition rounds to float, and    */
/* v2.ref_pos is converted to float for the comparison.  Below 2^24 that is    */
/* integer arithmetic; above (most of every human chromosome) positions are    */
/* rounded to multiples of 2, 4, 8, 16 and the answer can differ from the      */
/* exact one in both directions.  Written with the same operand types, so the  */
/* C compiler applies the same conversions (ceilf = C++'s std::ceil(float)).   */
MO_API int mo_are_near(int v1_ref_pos, int v1_ref_size, int v1_min_size, int sum_to_add, int k, int v2_ref_pos)
{
    return v1_ref_pos + v1_ref_size - v1_min_size - 1 + sum_to_add + ceilf((float)k / 2) >= v2_ref_pos;
}

#include <pthread.h>
typedef struct {
    mo_bf *context_bf, *bf;
    mo_kmap *ref_bf;
    const uint64_t *hi, *lo;
    const uint32_t *counts;
    size_t begin, end;
    int k, ref_k;
} mt_job;
static void *mt_scan_slice(void *arg)
{
    static const char L[4] = {'A', 'C', 'G', 'T'};
    mt_job *j = (mt_job *)arg;
    const int k = j->k, ref_k = j->ref_k;
    char ctx[ref_k + 1], kmer[k + 1];
    size_t i;
    for (i = j->begin; i < j->end; i++) {
        int q;
        for (q = 0; q < ref_k; q++) {
            int sh = 2 * (ref_k - 1 - q);
            uint64_t c = sh >= 64 ? (j->hi[i] >> (sh - 64)) : (j->lo[i] >> sh);
            ctx[q] = L[c & 3];
        }
        ctx[ref_k] = 0;
        strncpy(kmer, ctx + ((ref_k - k) / 2), (size_t)k); /* main.cpp:493 */
        kmer[k] = 0;
        { /* ref_bf.increment, main.cpp:495 */
            char ck[k + 1];
            size_t n = km_canon(kmer, ck);
            int64_t e = km_find(j->ref_bf, ck, n, km_strhash(ck, n));
            if (e >= 0) __atomic_fetch_add((uint32_t *)&j->ref_bf->eval[e], (uint32_t)j->counts[i], __ATOMIC_RELAXED);
        }
        if (!mo_bf_test_key(j->context_bf, ctx) && j->bf->mode) { /* main.cpp:496-498 */
            uint64_t b = bf_hash(kmer) % j->bf->size;
            if (bf_bit(j->bf, b)) __atomic_fetch_add(&j->bf->counts[bf_rank(j->bf, b)], (uint16_t)j->counts[i], __ATOMIC_RELAXED);
        }
    }
    return NULL;
}
MO_API int mo_kmc_scan_packed_mt(mo_bf *context_bf, mo_bf *bf, mo_kmap *ref_bf, const uint64_t *hi, const uint64_t *lo,
                                 const uint32_t *counts, size_t n, int k, int ref_k, int n_threads)
{
    pthread_t th[256];
    mt_job job[256];
    int t, started = 0;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 256) n_threads = 256;
    for (t = 0; t < n_threads; t++) {
        job[t] = (mt_job){context_bf, bf, ref_bf, hi, lo, counts, n * (size_t)t / (size_t)n_threads,
                          n * (size_t)(t + 1) / (size_t)n_threads, k, ref_k};
        if (pthread_create(&th[t], NULL, mt_scan_slice, &job[t]) != 0) break;
        started++;
    }
    for (t = 0; t < started; t++) pthread_join(th[t], NULL);
    for (t = started; t < n_threads; t++) mt_scan_slice(&job[t]); /* threads that could not start: run here */
    return started;
}

/* Reference-context scan, main.cpp:383-401, one contig.  bf must be in read
 * mode only for symmetry with the reference (test_key ignores the mode).
 * std::string(reference, pos, n) clips at the end of the contig; a contig
 * shorter than (ref_k-k)/2 would throw in the reference -- returns -1 here. */
MO_API int mo_ref_scan(const mo_bf *bf, mo_bf *context_bf, const char *reference, size_t len, int k, int ref_k)
{
    size_t off = (size_t)((ref_k - k) / 2), p;
    char ksub[k + 1], ctx[ref_k + 1];
    size_t kn, cn;
    if (off > len) return -1;
    kn = len - off < (size_t)k ? len - off : (size_t)k;
    cn = len < (size_t)ref_k ? len : (size_t)ref_k;
    memcpy(ksub, reference + off, kn); ksub[kn] = 0;
    memcpy(ctx, reference, cn); ctx[cn] = 0;
    if (mo_bf_test_key(bf, ksub)) mo_bf_add_key(context_bf, ctx);
    for (p = (size_t)ref_k; p < len; ++p) {
        /* erase(0,1) then += : a sliding window once the strings are full length */
        memmove(ctx, ctx + 1, cn - 1); ctx[cn - 1] = reference[p];
        memmove(ksub, ksub + 1, kn - 1); ksub[kn - 1] = reference[p - off];
        if (mo_bf_test_key(bf, ksub)) mo_bf_add_key(context_bf, ctx);
    }
    return 0;
}

/* Weights of signature k-mers, main.cpp:166-170 */
MO_API void mo_lookup_weights(const mo_bf *bf, const mo_kmap *ref_bf, const char *rows, size_t stride, size_t n,
                              const uint8_t *is_ref, int32_t *w)
{
    size_t i;
    for (i = 0; i < n; i++)
        w[i] = is_ref[i] ? mo_kmap_get_count(ref_bf, rows + i * stride)
                         : (int32_t)mo_bf_get_count(bf, rows + i * stride);
}

/* set_coverages arithmetic, main.cpp:159-181, over flat descriptors:
 * allele slot a owns signatures [allele_sig_off[a], allele_sig_off[a+1]),
 * signature s owns k-mer weights [sig_kmer_off[s], sig_kmer_off[s+1]).
 * The result passes through a float parameter (var_block.hpp:84) into a uint
 * (variant.hpp:242). */
MO_API void mo_set_coverages(const int32_t *w, const uint64_t *sig_kmer_off, const uint64_t *allele_sig_off,
                             uint64_t n_alleles, uint32_t *cov)
{
    uint64_t a, s, j;
    for (a = 0; a < n_alleles; a++) {
        unsigned allele_cov = 0;
        for (s = allele_sig_off[a]; s < allele_sig_off[a + 1]; s++) {
            unsigned curr_cov = 0;
            int n = 0;
            for (j = sig_kmer_off[s]; j < sig_kmer_off[s + 1]; j++) {
                int wt = w[j];
                if (wt > 0) {
                    curr_cov = (curr_cov * n + wt) / (n + 1);
                    ++n;
                }
            }
            if (curr_cov > allele_cov) allele_cov = curr_cov;
        }
        {
            float f = (float)allele_cov;
            cov[a] = (uint32_t)f;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Genotype likelihoods: var_block.hpp:224-330, log_binomial :792-797.        */
/* log(float) is the float overload, log(int) the double one; every
 * uint*float product is rounded to float before it joins the double sum.     */
/* ------------------------------------------------------------------------- */

static double log_binomial(int n, int k) /* var_block.hpp:792-797 */
{
    if (n == 0 || n == k || k == 0) return 0;
    return n * log((double)n) - k * log((double)k) - (n - k) * log((double)(n - k));
}

/* Writes the computed_gts list: (g1, g2, value) triples in the reference's
 * emission order; haploid entries have g2 = -1.  Returns the entry count,
 * or -1 if cap is too small.  Early-outs emit (0,0|-1) entries exactly as
 * var_block.hpp:236-266 does (one per over-covered allele; value 1 when the
 * variant has a single allele; value 0 when nothing is covered). */
MO_API int mo_genotype(const uint32_t *cov, const float *freq, int A, float error_rate, int max_cov,
                       int haploid, int *g1s, int *g2s, double *vals, int cap)
{
    int n = 0, g1, g2, flag = 0;
    unsigned total_sum;
    int isum = 0;
    for (g1 = 0; g1 < A; g1++)
        if ((int)cov[g1] > max_cov) { /* var_block.hpp:237-246 */
            if (n >= cap) return -1;
            g1s[n] = 0; g2s[n] = haploid ? -1 : 0; vals[n] = 0; n++;
            flag = 1;
        }
    if (flag) return n;
    if (A == 1) { /* var_block.hpp:252-257 */
        if (n >= cap) return -1;
        g1s[n] = 0; g2s[n] = haploid ? -1 : 0; vals[n] = 1; return 1;
    }
    for (g1 = 0; g1 < A; g1++) isum += (int)cov[g1]; /* accumulate(..., 0): int */
    total_sum = (unsigned)isum;
    if (total_sum == 0) { /* var_block.hpp:260-266 */
        if (n >= cap) return -1;
        g1s[n] = 0; g2s[n] = haploid ? -1 : 0; vals[n] = 0; return 1;
    }
    if (haploid) { /* var_block.hpp:268-287 */
        for (g1 = 0; g1 < A; g1++) {
            unsigned truth = cov[g1], error = total_sum - truth;
            double log_prior = (double)(2 * logf(freq[g1]));
            double log_post = log_binomial((int)(truth + error), (int)truth)
                              + (double)((float)truth * logf(1 - error_rate))
                              + (double)((float)error * logf(error_rate / (float)(unsigned long)(A - 1)));
            double lp = log_prior + log_post, prob = 0;
            if (!isinf(lp)) prob = exp(lp);
            if (n >= cap) return -1;
            g1s[n] = g1; g2s[n] = -1; vals[n] = prob; n++;
        }
        return n;
    }
    for (g1 = 0; g1 < A; g1++) /* var_block.hpp:290-327 */
        for (g2 = g1; g2 < A; g2++) {
            double log_prior, log_post, lp, prob = 0;
            if (g1 == g2) {
                unsigned truth = cov[g1], error = total_sum - truth;
                log_prior = (double)(2 * logf(freq[g1]));
                log_post = log_binomial((int)(truth + error), (int)truth)
                           + (double)((float)truth * logf(1 - error_rate))
                           + (double)((float)error * logf(error_rate / (float)(unsigned long)(A - 1)));
            } else {
                unsigned t1 = cov[g1], t2 = cov[g2], error = total_sum - t1 - t2;
                log_prior = (double)logf(2 * freq[g1] * freq[g2]);
                log_post = log_binomial((int)(t1 + t2 + error), (int)(t1 + t2))
                           + log_binomial((int)(t1 + t2), (int)t1)
                           + (double)((float)t1 * logf((1 - error_rate) / 2))
                           + (double)((float)t2 * logf((1 - error_rate) / 2));
                if (A > 2)
                    log_post += (double)((float)error * logf(error_rate / (float)(unsigned long)(A - 2)));
            }
            lp = log_prior + log_post;
            if (!isinf(lp)) prob = exp(lp);
            if (n >= cap) return -1;
            g1s[n] = g1; g2s[n] = g2; vals[n] = prob; n++;
        }
    return n;
}

/* Normalise, first-strict-max, GQ: var_block.hpp:366-394.
 * Returns the index of the winning entry, or -1 when no entry beats 0.0 (the
 * caller then prints best_geno "0/0" or "0").  norm[i] = vals[i]/total. */
MO_API int mo_select_gt(const double *vals, int n, double *norm, int *gq)
{
    double total = 0, best = 0;
    int i, bi = -1;
    for (i = 0; i < n; i++) total += vals[i];
    for (i = 0; i < n; i++) {
        double q = vals[i] / total;
        if (norm) norm[i] = q;
        if (q > best) { best = q; bi = i; }
    }
    *gq = (int)round(best * 100);
    return bi;
}

/* ------------------------------------------------------------------------- */
/* Loop B specialised to isolated variants whose alleles are all shorter than
 * k (the C3 benchmark shape: one variant per block, so the only chain is the
 * variant itself).  Restates, per variant:
 *   extract_kmers  var_block.hpp:104-112 (skip rule), :145-200 with comb={v}:
 *       missing_prefix = k/2 - len/2, missing_suffix = ceil(k/2) - (len - len/2)
 *       kmer = ref[pos-mp, pos) + allele + ref[pos+ref_size, +ms)
 *     one signature per allele that some panel haplotype carries
 *     (build_alleles_combs :734-786 -> `present` bit mask, bit a = allele a)
 *   set_coverages  main.cpp:151-184 (single k-mer: cov = w if w > 0)
 *   genotype + select as above.
 * Alleles are given as offsets into one byte pool.  Outputs per variant:
 * cov[A], best (g1,g2), GQ, and the normalised list (optional).            */
/* ------------------------------------------------------------------------- */
MO_API void mo_call_isolated(const mo_bf *bf, const mo_kmap *ref_bf, const char *reference, size_t ref_len,
                             size_t n_vars, const int64_t *pos, const uint32_t *allele_off /* n_alleles_total+1 */,
                             const uint32_t *var_allele_off /* n_vars+1 */, const char *allele_pool,
                             const float *freq /* per allele slot */, const uint64_t *present_mask,
                             const uint8_t *is_present, int k, float error_rate, int max_cov, int haploid,
                             uint32_t *cov_out, int32_t *gt1, int32_t *gt2, int32_t *gq_out)
{
    size_t v;
    for (v = 0; v < n_vars; v++) {
        uint32_t a0 = var_allele_off[v], A = var_allele_off[v + 1] - a0, a;
        uint32_t ref_size = allele_off[a0 + 1] - allele_off[a0];
        int64_t p = pos[v];
        uint32_t *cov = cov_out + a0;
        for (a = 0; a < A; a++) cov[a] = 0;
        if (is_present[v] && p >= k && p <= (int64_t)ref_len - k) {
            for (a = 0; a < A; a++) {
                uint32_t alen = allele_off[a0 + a + 1] - allele_off[a0 + a];
                int mp, ms, w;
                char kmer[4 * k + 8];
                size_t L = 0;
                if (!((present_mask[v] >> a) & 1)) continue;
                mp = k / 2 - (int)(alen / 2);
                ms = (int)ceilf((float)k / 2) - (int)(alen - alen / 2);
                if (mp < 0 || ms < 0) { fprintf(stderr, "mo_call_isolated: allele >= k not supported\n"); abort(); }
                memcpy(kmer, reference + p - mp, (size_t)mp); L += (size_t)mp;
                memcpy(kmer + L, allele_pool + allele_off[a0 + a], alen); L += alen;
                {
                    size_t s = (size_t)p + ref_size, m = (size_t)ms;
                    if (s > ref_len) s = ref_len;
                    if (s + m > ref_len) m = ref_len - s;
                    memcpy(kmer + L, reference + s, m); L += m;
                }
                kmer[L] = 0;
                w = a == 0 ? mo_kmap_get_count(ref_bf, kmer) : (int)mo_bf_get_count(bf, kmer);
                if (w > 0) { float f = (float)(unsigned)w; cov[a] = (uint32_t)f; }
            }
        }
        {
            int cap = (int)(A * (A + 1) / 2 + A + 2), n, bi, gq;
            int g1s[cap], g2s[cap];
            double vals[cap];
            n = mo_genotype(cov, freq + a0, (int)A, error_rate, max_cov, haploid, g1s, g2s, vals, cap);
            bi = mo_select_gt(vals, n, NULL, &gq);
            gt1[v] = bi < 0 ? 0 : g1s[bi];
            gt2[v] = bi < 0 ? (haploid ? -1 : 0) : g2s[bi];
            gq_out[v] = gq;
        }
    }
}

/* Load a bit array built elsewhere (e.g. exported from the device index, whose
 * own parity is tested separately) so a timing run does not have to rebuild it. */
MO_API void mo_bf_load_words(mo_bf *b, const uint64_t *words)
{
    memcpy(b->words, words, b->nwords * 8);
}
