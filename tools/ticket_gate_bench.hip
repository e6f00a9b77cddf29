// Microbenchmark behind pass two of the scan's ticket form (scan_ticket_gate_kernel): n 8-byte tickets, filed under P
// slices of SLICE bytes of a gate; XCD x (workgroups b = x mod 8) walks slices x, x + 8, ... and reads, per ticket, one
// random 8-byte word of the slice, which should sit in that XCD's L2.  What rate does the chip sustain for this pattern,
// by workgroup shape, tickets in flight per thread, slice size, and with / without a rendezvous between slices?
//   hipcc --offload-arch=gfx950 -O3 -o bin/ticket_gate_bench tools/ticket_gate_bench.hip && bin/ticket_gate_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e = (x);                                                                    \
        if (e != hipSuccess) {                                                                 \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                             \
            return 1;                                                                          \
        }                                                                                      \
    } while (0)

__global__ void fill(uint64_t *tk, uint64_t n, uint64_t per_slice, uint32_t slice_words_log2)
{
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t h = i * 0x9E3779B97F4A7C15ULL;
    h ^= h >> 29;
    h *= 0xBF58476D1CE4E5B9ULL;
    const uint64_t p = i / per_slice;
    const uint64_t word = (p << slice_words_log2) | ((h >> 20) & ((1ULL << slice_words_log2) - 1));
    tk[i] = (word << 27) | (i & ((1ULL << 27) - 1));
}

// MODE 0: gather from the ticket's own slice; 1: every round gathers from the XCD's FIRST slice (no slice change);
// 2: no gather (stream only); 3: no ticket stream (made-up tickets of the right slice)
template <int TPB, int U, int MODE>
__global__ void __launch_bounds__(TPB) walk(const uint64_t *__restrict__ tk, uint64_t per_slice, int P, uint32_t slice_words_log2,
                                            const uint64_t *__restrict__ gate, uint32_t *sync, unsigned long long *sink)
{
    const uint32_t xcd = blockIdx.x & 7, local = blockIdx.x >> 3, nlocal = gridDim.x >> 3;
    const uint64_t share = (per_slice + nlocal - 1) / nlocal;
    const uint64_t wmask = (1ULL << slice_words_log2) - 1;
    uint64_t acc = 0;
    for (int p = (int)xcd; p < P; p += 8) {
        const uint64_t b0 = (uint64_t)p * per_slice + (share * local < per_slice ? share * local : per_slice);
        const uint64_t b1 = (uint64_t)p * per_slice + (share * (local + 1) < per_slice ? share * (local + 1) : per_slice);
        const uint64_t *first = tk + b0;
        const uint32_t total = (uint32_t)(b1 - b0);
        uint64_t a[U], b[U], c[U];
        auto fetch = [&](uint32_t base, uint64_t (&t)[U]) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const uint32_t q = base + u * TPB + threadIdx.x;
                if (MODE == 3) {
                    uint64_t h = (b0 + q) * 0x9E3779B97F4A7C15ULL;
                    h ^= h >> 29;
                    t[u] = ((((uint64_t)p << slice_words_log2) | ((h >> 20) & wmask)) << 27) | q;
                } else
                    t[u] = __builtin_nontemporal_load(first + (q < total ? q : 0));
            }
        };
        fetch(0, a);
        fetch(U * TPB, b);
        for (uint32_t base = 0; base < total; base += U * TPB) {
            uint64_t w[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                uint64_t word = a[u] >> 27;
                if (MODE == 1) word = ((uint64_t)xcd << slice_words_log2) | (word & wmask);
                w[u] = MODE == 2 ? a[u] : gate[word];
            }
            asm volatile("" ::: "memory");
            fetch(base + 2 * U * TPB, c);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < U; ++u) acc += w[u] ^ a[u];
#pragma unroll
            for (int u = 0; u < U; ++u) a[u] = b[u], b[u] = c[u];
        }
        if (sync && p + 8 < P) {
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t *const arrived = sync + xcd * 64 + (p >> 3);
                atomicAdd(arrived, 1u);
                for (int spin = 0; spin < 4096 && __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nlocal; ++spin)
                    __builtin_amdgcn_s_sleep(2);
            }
            __syncthreads();
        }
    }
    if (acc == 0x1234567) atomicAdd(sink, 1ULL);
}

int main()
{
    const uint64_t n = 1ULL << 27;
    uint64_t *tk, *gate;
    uint32_t *sync;
    unsigned long long *sink;
    CK(hipMalloc(&tk, n * 8 + 4096));
    CK(hipMalloc(&gate, 256ULL << 20));
    CK(hipMalloc(&sync, 8 * 64 * 4));
    CK(hipMalloc(&sink, 8));
    CK(hipMemset(gate, 0, 256ULL << 20));
    CK(hipMemset(sink, 0, 8));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int slice_log2 = 20; slice_log2 <= 21; ++slice_log2) { // slice bytes: 1 MiB, 2 MiB (gate 256 MiB -> 256 / 128 slices)
        const int P = (int)((256ULL << 20) >> slice_log2);
        const uint32_t swl = (uint32_t)slice_log2 - 3;
        const uint64_t per_slice = n / P;
        hipLaunchKernelGGL(fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, tk, n, per_slice, swl);
        CK(hipDeviceSynchronize());
        auto time = [&](auto kern, int grid, int tpb, bool with_sync) -> float {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemsetAsync(sync, 0, 8 * 64 * 4, 0);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(kern, dim3(grid), dim3(tpb), 0, 0, tk, per_slice, P, swl, gate, with_sync ? sync : (uint32_t *)nullptr, sink);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            return best;
        };
        printf("== %d slices of %d KiB, %llu tickets ==\n", P, 1 << (slice_log2 - 10), (unsigned long long)n);
#define ROW(TPB, U)                                                                                                                          \
    for (int per_cu = 1; per_cu * TPB <= 2048; per_cu *= 2) {                                                                                \
        const int grid = 256 * per_cu;                                                                                                       \
        printf("TPB %4d U %d wg/CU %d: own slice %.3f (sync %.3f) | fixed slice %.3f | stream only %.3f | gather only %.3f ms\n", TPB, U,   \
               per_cu, time(walk<TPB, U, 0>, grid, TPB, false), time(walk<TPB, U, 0>, grid, TPB, true), time(walk<TPB, U, 1>, grid, TPB, false), \
               time(walk<TPB, U, 2>, grid, TPB, false), time(walk<TPB, U, 3>, grid, TPB, false));                                            \
    }
        ROW(1024, 4)
        ROW(1024, 8)
        ROW(512, 4)
        ROW(256, 4)
        ROW(256, 8)
        ROW(256, 2)
    }
    return 0;
}
