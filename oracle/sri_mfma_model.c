/*
 * sri_mfma_model.c — TEST INFRASTRUCTURE (like the rest of oracle/): CPU models of the arithmetic the gfx950 apply
 * kernel performs, used to (1) check on the GPU box that the kernel's summation is understood bit for bit and
 * (2) price alternative summation orders against an extended-precision product without spending GPU time.
 * Nothing here is reference code or product code; the product never links it.
 *
 * The kernel (csrc/sri_apply.hip) accumulates every output element with v_mfma_f64_16x16x4_f64: one instruction adds
 * four products a(k) * x(k) to the accumulator.  k is consumed chunk by chunk (16 k per LDS stage); inside a chunk the
 * instruction order is (k pair j = 0,1) x (element e = 0,1), and the four lanes' k slots g = 0..3 of one instruction
 * hold k = 16*chunk + 8*j + 2*g + e (sri_internal.h: k_of) — a consequence of reading X with 16-byte loads from the
 * caller's column-major panel.  `mode` selects the model:
 *   0  kernel order, each instruction a chain of four fused multiply-adds in ascending g          (the hypothesis)
 *   1  kernel order, chain in descending g
 *   2  kernel order, the four products and the accumulator summed exactly (binary128), one rounding per instruction
 *   3  kernel order, pairwise inside the instruction: (p0 + p1) + (p2 + p3) in binary128, then + c, one rounding
 *   4  strictly ascending k, one fused multiply-add per term            (what 8-byte X-fragment reads would give)
 *   5  strictly ascending k, multiply then add (two roundings per term)  == sri_oracle_apply (main.cpp:172 as Eigen's
 *      default traversal would do it term by term)
 * Zero padding of the operator (k >= cols) contributes fma(0, x, c) = c and is skipped.
 */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

static inline double mfma4(int mode, const double a[4], const double x[4], int n, double c)
{
    if (mode == 0) {
        for (int g = 0; g < n; ++g) c = fma(a[g], x[g], c);
        return c;
    }
    if (mode == 1) {
        for (int g = n - 1; g >= 0; --g) c = fma(a[g], x[g], c);
        return c;
    }
    __float128 p[4] = {0, 0, 0, 0};
    for (int g = 0; g < n; ++g) p[g] = (__float128)a[g] * (__float128)x[g];   /* exact: 106 bits fit 113 */
    if (mode == 2) return (double)((((__float128)c + p[0]) + p[1]) + p[2] + p[3]);
    return (double)((__float128)c + ((p[0] + p[1]) + (p[2] + p[3])));
}

int sri_model_apply(int mode, uint32_t r, uint32_t k, const double *A, size_t lda, const double *X, size_t ldx,
                    double *Y, size_t ldy, size_t M)
{
    if (!A || !X || !Y || lda < r || ldx < k || ldy < r || mode < 0 || mode > 5) return -1;
#pragma omp parallel for schedule(static)
    for (long jj = 0; jj < (long)M; ++jj) {
        const size_t j = (size_t)jj;
        const double *xj = X + j * ldx;
        for (uint32_t i = 0; i < r; ++i) {
            double c = 0.0;
            if (mode == 4) {
                for (uint32_t kk = 0; kk < k; ++kk) c = fma(A[i + (size_t)kk * lda], xj[kk], c);
            } else if (mode == 5) {
                for (uint32_t kk = 0; kk < k; ++kk) {
                    volatile double prod = A[i + (size_t)kk * lda] * xj[kk];   /* no contraction */
                    c = c + prod;
                }
            } else {
                for (uint32_t k0 = 0; k0 < k; k0 += 16)
                    for (int jp = 0; jp < 2; ++jp)
                        for (int e = 0; e < 2; ++e) {
                            double a4[4], x4[4];
                            int n = 0;
                            for (int g = 0; g < 4; ++g) {
                                const uint32_t kk = k0 + 8 * jp + 2 * g + e;
                                if (kk < k) { a4[n] = A[i + (size_t)kk * lda]; x4[n] = xj[kk]; ++n; }
                            }
                            if (n) c = mfma4(mode, a4, x4, n, c);
                        }
            }
            Y[i + j * ldy] = c;
        }
    }
    return 0;
}

/* Extended-precision reference product: binary128 accumulation of exact products, rounded once. */
int sri_model_apply_exact(uint32_t r, uint32_t k, const double *A, size_t lda, const double *X, size_t ldx,
                          double *Y, size_t ldy, size_t M)
{
    if (!A || !X || !Y || lda < r || ldx < k || ldy < r) return -1;
#pragma omp parallel for schedule(static)
    for (long jj = 0; jj < (long)M; ++jj) {
        const size_t j = (size_t)jj;
        for (uint32_t i = 0; i < r; ++i) {
            __float128 s = 0;
            for (uint32_t kk = 0; kk < k; ++kk) s += (__float128)A[i + (size_t)kk * lda] * (__float128)X[kk + j * ldx];
            Y[i + j * ldy] = (double)s;
        }
    }
    return 0;
}
