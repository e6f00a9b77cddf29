// Microbenchmark behind DESIGN.md's bound for the scan's filter kernel: how many independent, uniformly random
// 8-byte reads per second one MI355X sustains from a table of F bytes (F = 16 KiB .. 1 GiB: L2-resident up to
// 4 MiB per XCD), alone and beside a non-temporal 20-byte-per-lane stream shaped like the k-mer table.
//   make microbench && bin/l2_gather_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e = (x);                                                                    \
        if (e != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                             \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

// rows: n; every thread takes 2 rows per iteration (as the filter kernel), reads the stream (optional) and one
// random word of the table per row; the word index is a cheap mix of the row number (or of the streamed data)
// SPOL: cache policy of the stream's loads (buffer builtins' aux: 1 sc0, 2 nt, 16 sc1), -1 = flat nt loads
template <bool STREAM, bool GATHER, int SPOL = -1>
__global__ void __launch_bounds__(256) bench(const uint64_t *__restrict__ hi, const uint64_t *__restrict__ lo,
                                             const uint32_t *__restrict__ cnt, uint64_t n, const uint64_t *__restrict__ table,
                                             uint64_t mask, unsigned long long *sink)
{
    uint64_t acc = 0;
    const uint64_t step = (uint64_t)gridDim.x * 512;
    for (uint64_t base = (uint64_t)blockIdx.x * 512; base + 512 <= n; base += step) {
        const uint64_t i = base + 2 * threadIdx.x;
        uint64_t a0 = i * 0x9E3779B97F4A7C15ULL, a1 = (i + 1) * 0x9E3779B97F4A7C15ULL;
        if (STREAM && SPOL >= 0) {
            typedef unsigned int __attribute__((ext_vector_type(4))) v4u;
            typedef unsigned int __attribute__((ext_vector_type(2))) v2u;
            const __amdgpu_buffer_rsrc_t rl = __builtin_amdgcn_make_buffer_rsrc((void *)lo, 0, 0x7FFFFFFF, 0x00020000);
            const __amdgpu_buffer_rsrc_t rh = __builtin_amdgcn_make_buffer_rsrc((void *)hi, 0, 0x7FFFFFFF, 0x00020000);
            const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void *)cnt, 0, 0x7FFFFFFF, 0x00020000);
            const v4u l4 = __builtin_amdgcn_raw_buffer_load_b128(rl, (int)(i * 8), 0, SPOL);
            const v4u h4 = __builtin_amdgcn_raw_buffer_load_b128(rh, (int)(i * 8), 0, SPOL);
            const v2u c2 = __builtin_amdgcn_raw_buffer_load_b64(rc, (int)(i * 4), 0, SPOL);
            a0 ^= l4.x + h4.y + c2.x + ((uint64_t)(l4.y ^ h4.x) << 32);
            a1 ^= l4.z + h4.w + c2.y + ((uint64_t)(l4.w ^ h4.z) << 32);
        } else if (STREAM) {
            typedef unsigned long long __attribute__((ext_vector_type(2))) v2u64;
            typedef unsigned int __attribute__((ext_vector_type(2))) v2u32;
            const v2u64 l2 = __builtin_nontemporal_load((const v2u64 *)(lo + i));
            const v2u64 h2 = __builtin_nontemporal_load((const v2u64 *)(hi + i));
            const v2u32 c2 = __builtin_nontemporal_load((const v2u32 *)(cnt + i));
            a0 ^= l2.x + h2.x + c2.x;
            a1 ^= l2.y + h2.y + c2.y;
        }
        a0 ^= a0 >> 29;
        a1 ^= a1 >> 29;
        a0 *= 0xBF58476D1CE4E5B9ULL;
        a1 *= 0xBF58476D1CE4E5B9ULL;
        if (GATHER) {
            const uint64_t g0 = table[(a0 >> 20) & mask], g1 = table[(a1 >> 20) & mask];
            acc += g0 + g1;
        } else
            acc += a0 + a1;
    }
    if (acc == 0x1234567) atomicAdd(sink, 1ULL);
}

int main()
{
    const uint64_t n = 100000000ULL / 512 * 512;
    uint64_t *hi, *lo, *table;
    uint32_t *cnt;
    unsigned long long *sink;
    CK(hipMalloc(&hi, n * 8));
    CK(hipMalloc(&lo, n * 8));
    CK(hipMalloc(&cnt, n * 4));
    CK(hipMalloc(&table, 1ULL << 30));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(hi, 1, n * 8));
    CK(hipMemset(lo, 2, n * 8));
    CK(hipMemset(cnt, 3, n * 4));
    CK(hipMemset(table, 5, 1ULL << 30));
    CK(hipMemset(sink, 0, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time = [&](auto kern, uint64_t mask) -> float {
        float best = 1e9f;
        for (int rep = 0; rep < 6; ++rep) {
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(kern, dim3(8192), dim3(256), 0, 0, hi, lo, cnt, n, table, mask, sink);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep && ms < best) best = ms;
        }
        return best;
    };
    printf("rows %llu, 2 rows per thread, grid 8192 x 256\n", (unsigned long long)n);
    const float s_only = time(bench<true, false>, 0);
    printf("stream only (20 B/row, nt):            %.3f ms  %.2f TB/s\n", s_only, n * 20.0 / s_only * 1e-9);
    const float none = time(bench<false, false>, 0);
    printf("no memory (index arithmetic only):      %.3f ms\n", none);
    {
        const uint64_t m4 = (4ULL << 20) / 8 - 1;
        printf("stream policy, alone / beside a 4 MiB gather:  flat nt %.3f / %.3f | buffer plain %.3f / %.3f | sc0 %.3f / %.3f | nt %.3f / %.3f | sc1 %.3f / %.3f |"
               " sc0 sc1 %.3f / %.3f | sc1 nt %.3f / %.3f | sc0 sc1 nt %.3f / %.3f ms\n",
               time(bench<true, false>, 0), time(bench<true, true>, m4), time(bench<true, false, 0>, 0), time(bench<true, true, 0>, m4),
               time(bench<true, false, 1>, 0), time(bench<true, true, 1>, m4), time(bench<true, false, 2>, 0), time(bench<true, true, 2>, m4),
               time(bench<true, false, 16>, 0), time(bench<true, true, 16>, m4), time(bench<true, false, 17>, 0), time(bench<true, true, 17>, m4),
               time(bench<true, false, 18>, 0), time(bench<true, true, 18>, m4), time(bench<true, false, 19>, 0), time(bench<true, true, 19>, m4));
    }
    for (int lg = 14; lg <= 30; lg += (lg < 26 ? 2 : 1)) {
        const uint64_t words = (1ULL << lg) / 8;
        const float g = time(bench<false, true>, words - 1), sg = time(bench<true, true>, words - 1);
        printf("table %6llu KiB: gather only %.3f ms = %.3g reads/s | stream + gather %.3f ms = %.3g rows/s (stream only + gather only = %.3f)\n",
               (unsigned long long)((1ULL << lg) >> 10), g, n / (g * 1e-3), sg, n / (sg * 1e-3), s_only + g);
    }
    return 0;
}
