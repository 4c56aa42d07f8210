/*
 * sri_oracle_mt.c — CPU ORACLE, all-cores variant of the apply (test infrastructure only).
 *
 * Same arithmetic as sri_oracle_apply() in sri_oracle.c (main.cpp:172: k-ascending
 * multiply-then-add per output element), with the independent columns (rods) split across
 * OpenMP threads.  Used only by bench.py's cpu_baseline leg to report the "honest best CPU"
 * figure next to the single-thread one; results are bit-identical to the
 * single-thread oracle because every column is computed by exactly the same instruction
 * sequence.
 */
#include <stddef.h>
#include <stdint.h>
#include <omp.h>

int sri_oracle_apply(uint32_t r, uint32_t k, const double *A, size_t lda,
                     const double *X, size_t ldx, double *Y, size_t ldy, size_t M);

int sri_oracle_apply_mt(uint32_t r, uint32_t k, const double *A, size_t lda,
                        const double *X, size_t ldx, double *Y, size_t ldy, size_t M, int threads)
{
    if (threads < 1) threads = omp_get_max_threads();
    int rc = 0;
    const size_t chunk = 64;
    const long nchunks = (long)((M + chunk - 1) / chunk);
#pragma omp parallel for num_threads(threads) schedule(dynamic, 4)
    for (long c = 0; c < nchunks; ++c) {
        const size_t j0 = (size_t)c * chunk;
        const size_t m = (j0 + chunk <= M) ? chunk : (M - j0);
        int r1 = sri_oracle_apply(r, k, A, lda, X + j0 * ldx, ldx, Y + j0 * ldy, ldy, m);
        if (r1 != 0) {
#pragma omp critical
            rc = r1;
        }
    }
    return rc;
}

int sri_oracle_max_threads(void) { return omp_get_max_threads(); }
