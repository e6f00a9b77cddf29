// Genotype likelihoods on the device: VB::genotype (var_block.hpp:224-330),
// log_binomial (:792-797) and the normalise / first-strict-max / GQ step of
// VB::output_variants (:366-394).
//
// Numerics follow SURVEY Appendix A.4: log(float) is the float overload, every
// uint*float product is rounded to float before it joins the double sum, sums
// run left to right, and nothing may be contracted into an FMA (this file is
// compiled with -ffp-contract=off).  Values that only depend on the error rate
// and the allele count -- logf(1-e), logf(e/(A-1)), logf((1-e)/2), logf(e/(A-2))
// -- and ln(n) for small n come from tables the host fills with its own libm
// (parameter preprocessing, like a RoPE table): a 1-ulp difference in those
// floats would be multiplied by the coverage.
#pragma once
#include "kmer_dev.h"

namespace mg {

#define MG_LN_TABLE 65536 // ln(n) for n < this comes from the host table
#define MG_EPS_TABLE 256  // per-allele-count constants for A < this

struct GenoParams {
    const double *ln_tab; // [MG_LN_TABLE], ln_tab[0] unused
    const float *c_err1;  // [MG_EPS_TABLE] logf(e / (A-1)), index A
    const float *c_err2;  // [MG_EPS_TABLE] logf(e / (A-2)), index A
    float c_hom;          // logf(1 - e)
    float c_het;          // logf((1 - e) / 2)
    float error_rate;
    int max_cov;
    int haploid;
};

// logf as the reference's libm computes it.  glibc >= 2.28 uses the table-driven
// algorithm of ARM's optimized-routines (math/logf.c, MIT licence): 16-entry
// (1/c, ln c) table, degree-3 polynomial, all in double, one final rounding.
// Restating that algorithm (rather than calling the device's own logf, which
// differs from it in the last bit for a few percent of inputs) keeps log priors
// bit-identical to the host's.  Checked against glibc logf on 1.1e8 inputs with
// zero mismatches before being moved here.
__device__ __constant__ const double kLogfInvC[16] = {
    0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010bp+0,  0x1.3c995b0b80385p+0,
    0x1.30d190c8864a5p+0, 0x1.25e227b0b8eap+0,  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0,
    0x1.0953f419900a7p+0, 0x1p+0,               0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aap-1,
    0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1, 0x1.767dcf5534862p-1};
__device__ __constant__ const double kLogfLogC[16] = {
    -0x1.57bf7808caadep-2, -0x1.2bef0a7c06ddbp-2, -0x1.01eae7f513a67p-2, -0x1.b31d8a68224e9p-3,
    -0x1.6574f0ac07758p-3, -0x1.1aa2bc79c81p-3,   -0x1.a4e76ce8c0e5ep-4, -0x1.1973c5a611cccp-4,
    -0x1.252f438e10c1ep-5, 0x0p+0,                0x1.aa5aa5df25984p-5,  0x1.c5e53aa362eb4p-4,
    0x1.526e57720db08p-3,  0x1.bc2860d22477p-3,   0x1.1058bc8a07ee1p-2,  0x1.4043057b6ee09p-2};
__device__ __forceinline__ float logf_ref(float x)
{
    u32 ix = __float_as_uint(x);
    if (ix == 0x3f800000u) return 0.f;
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u) {
        if (ix * 2 == 0) return -INFINITY;                          // log(+-0)
        if (ix == 0x7f800000u) return x;                             // log(inf)
        if ((ix & 0x80000000u) || ix * 2 >= 0xff000000u) return NAN; // negative or NaN
        ix = __float_as_uint(x * 0x1p23f) - (23u << 23);             // subnormal: normalise
    }
    const u32 tmp = ix - 0x3f330000u;
    const int i = (tmp >> 19) & 15;
    const int k = (int)tmp >> 23;
    const u32 iz = ix - (tmp & 0xff800000u);
    const double z = (double)__uint_as_float(iz);
    const double r = z * kLogfInvC[i] - 1.0;
    const double y0 = kLogfLogC[i] + (double)k * 0x1.62e42fefa39efp-1;
    const double r2 = r * r;
    double y = 0x1.5575b0be00b6ap-2 * r + -0x1.ffffef20a4123p-2;
    y = -0x1.00ea348b88334p-2 * r2 + y;
    y = y * r2 + (y0 + r);
    return (float)y;
}

__device__ __forceinline__ double ln_int(int n, const GenoParams &p)
{
    return n < MG_LN_TABLE ? p.ln_tab[n] : log((double)n);
}
__device__ __forceinline__ double log_binomial(int n, int k, const GenoParams &p)
{
    if (n == 0 || n == k || k == 0) return 0.0;
    const double a = n * ln_int(n, p);
    const double b = k * ln_int(k, p);
    const double c = (n - k) * ln_int(n - k, p);
    return a - b - c;
}
__device__ __forceinline__ float c_err1(int A, const GenoParams &p)
{
    return A < MG_EPS_TABLE ? p.c_err1[A] : logf_ref(p.error_rate / (float)(unsigned long)(A - 1));
}
__device__ __forceinline__ float c_err2(int A, const GenoParams &p)
{
    return A < MG_EPS_TABLE ? p.c_err2[A] : logf_ref(p.error_rate / (float)(unsigned long)(A - 2));
}

// unnormalised probability of genotype (g1, g2); g2 < 0 or g1 == g2: homozygous / haploid form
__device__ __forceinline__ double gt_value(const u32 *cov, const float *freq, int A, u32 total, int g1, int g2,
                                           const GenoParams &p)
{
    double log_prior, log_post;
    if (g2 < 0 || g1 == g2) {
        const u32 truth = cov[g1], error = total - truth;
        log_prior = (double)(2 * logf_ref(freq[g1]));
        const float t1 = (float)truth * p.c_hom;
        const float t2 = (float)error * c_err1(A, p);
        log_post = log_binomial((int)(truth + error), (int)truth, p) + (double)t1 + (double)t2;
    } else {
        const u32 t1c = cov[g1], t2c = cov[g2], error = total - t1c - t2c;
        const float pr = 2 * freq[g1] * freq[g2];
        log_prior = (double)logf_ref(pr);
        const float t1 = (float)t1c * p.c_het;
        const float t2 = (float)t2c * p.c_het;
        log_post = log_binomial((int)(t1c + t2c + error), (int)(t1c + t2c), p) + log_binomial((int)(t1c + t2c), (int)t1c, p)
                   + (double)t1 + (double)t2;
        if (A > 2) {
            const float t3 = (float)error * c_err2(A, p);
            log_post += (double)t3;
        }
    }
    const double lp = log_prior + log_post;
    return isinf(lp) ? 0.0 : exp(lp);
}

// One variant.  Writes gt1/gt2/gq/status and, if probs != nullptr, the
// normalised list in the reference's emission order.
__device__ inline void genotype_one(const u32 *cov, const float *freq, int A, const GenoParams &p, i32 *gt1, i32 *gt2,
                                    i32 *gq, u8 *status, double *probs)
{
    const int dflt2 = p.haploid ? -1 : 0;
    *gt1 = 0;
    *gt2 = dflt2;
    *gq = 0;
    bool over = false;
    int isum = 0;
    for (int a = 0; a < A; ++a) {
        over |= (int)cov[a] > p.max_cov;
        isum += (int)cov[a];
    }
    if (over) { // var_block.hpp:236-248: (best_geno, 0) per over-covered allele; 0/0 = NaN never wins
        *status = 1;
        return;
    }
    if (A == 1) { // var_block.hpp:252-257: (best_geno, 1) -> q = 1 -> GQ 100
        *status = 2;
        *gq = 100;
        return;
    }
    const u32 total = (u32)isum;
    if (total == 0) { // var_block.hpp:260-266
        *status = 3;
        return;
    }
    *status = 0;
    // pass 1: raw values (kept in `probs` when the caller gave room for them), total in list order
    double sum = 0.0;
    int n = 0;
    if (p.haploid) {
        for (int g = 0; g < A; ++g, ++n) {
            const double val = gt_value(cov, freq, A, total, g, -1, p);
            if (probs) probs[n] = val;
            sum += val;
        }
    } else {
        for (int g1 = 0; g1 < A; ++g1)
            for (int g2 = g1; g2 < A; ++g2, ++n) {
                const double val = gt_value(cov, freq, A, total, g1, g2, p);
                if (probs) probs[n] = val;
                sum += val;
            }
    }
    // pass 2: normalise, first strictly greater wins
    double best = 0.0;
    n = 0;
    if (p.haploid) {
        for (int g = 0; g < A; ++g, ++n) {
            const double q = (probs ? probs[n] : gt_value(cov, freq, A, total, g, -1, p)) / sum;
            if (probs) probs[n] = q;
            if (q > best) {
                best = q;
                *gt1 = g;
                *gt2 = -1;
            }
        }
    } else {
        for (int g1 = 0; g1 < A; ++g1)
            for (int g2 = g1; g2 < A; ++g2, ++n) {
                const double q = (probs ? probs[n] : gt_value(cov, freq, A, total, g1, g2, p)) / sum;
                if (probs) probs[n] = q;
                if (q > best) {
                    best = q;
                    *gt1 = g1;
                    *gt2 = g2;
                }
            }
    }
    *gq = (i32)round(best * 100.0);
}

} // namespace mg
