// valu_rate_bench.hip -- issue cost of the integer instructions the XXH3 / canonical-form code is made of, on gfx950:
// every CU runs 16 waves (4 per SIMD) of a loop of 64 independent instances of ONE instruction; cycles per wave-instruction
// per SIMD = elapsed clock x SIMDs / instructions issued.  Build: hipcc --offload-arch=gfx950 -O3 -o bin/valu_rate_bench tools/valu_rate_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
template <int OP> __global__ void __launch_bounds__(1024) k(uint32_t *out, int iters)
{
    uint32_t a[8], b[8];
    uint64_t q[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i, b[i] = a[i] ^ 0x9E3779B9u, q[i] = ((uint64_t)a[i] << 32) | b[i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 1) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
                if (OP == 2) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 3) asm volatile("v_mul_hi_u32 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 4) asm volatile("v_mul_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 5) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 7]));
                if (OP == 6) asm volatile("v_alignbyte_b32 %0, %0, %1, 3" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 7) asm volatile("v_lshlrev_b64 %0, 5, %0" : "+v"(q[i]));
                if (OP == 8) asm volatile("v_mul_hi_u32_u24 %0, %1, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 9) asm volatile("v_mad_u32_u24 %0, %1, %0, %0" : "+v"(a[i]) : "v"(b[i]));
                if (OP == 10) asm volatile("v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %2, vcc, %3, %2, vcc" : "+v"(a[i]), "+v"(b[i]) : "v"(a[(i + 1) & 7]), "v"(b[(i + 1) & 7]) : "vcc");
                if (OP == 11) asm volatile("v_bfrev_b32 %0, %0" : "+v"(a[i]));
                if (OP == 12) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(b[(i + 1) & 7]));
                if (OP == 13) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(q[i]) : "v"(a[i]), "v"(b[i]) : "vcc");
            }
        }
    }
    uint32_t r = 0;
    for (int i = 0; i < 8; ++i) r ^= a[i] ^ b[i] ^ (uint32_t)q[i] ^ (uint32_t)(q[i] >> 32);
    out[blockIdx.x * 1024 + threadIdx.x] = r;
}
template <int OP> void run(const char *name, int per)
{
    uint32_t *d;
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipMalloc(&d, (size_t)cus * 1024 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<OP><<<cus, 1024>>>(d, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<OP><<<cus, 1024>>>(d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per SIMD: 4 waves x iters x 64 instances x per instructions
    const double instr = 4.0 * iters * 64 * per;
    printf("%-28s %8.3f ms   %6.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", name, ms, ms * 1e-3 * 2.4e9 / instr);
    hipFree(d);
}
int main()
{
    run<0>("v_xor_b32", 1);
    run<1>("v_mad_u64_u32 (acc)", 1);
    run<13>("v_mad_u64_u32 (+0)", 1);
    run<2>("v_mul_lo_u32", 1);
    run<3>("v_mul_hi_u32", 1);
    run<4>("v_mul_u32_u24", 1);
    run<8>("v_mul_hi_u32_u24", 1);
    run<9>("v_mad_u32_u24", 1);
    run<5>("v_lshl_add_u64", 1);
    run<6>("v_alignbyte_b32", 1);
    run<7>("v_lshlrev_b64", 1);
    run<10>("v_add_co + v_addc_co", 2);
    run<11>("v_bfrev_b32", 1);
    run<12>("v_perm_b32", 1);
    return 0;
}
