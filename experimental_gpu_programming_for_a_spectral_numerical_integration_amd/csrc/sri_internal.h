// sri_internal.h — declarations shared by the translation units of libsri_hip.so (product code).
#ifndef SRI_INTERNAL_H
#define SRI_INTERNAL_H

#include <cstddef>
#include <cstdint>

#include "sri_capi.h"

namespace sri {

// ---- host formers (sri_host.cpp) -------------------------------------------------------------
int cheb_points(uint32_t n, double L, double* x);
int cheb_coeffs(uint32_t n, double* c);
int cheb_dn(uint32_t n, double* D, size_t ld);
int cheb_blocks(uint32_t n, const double* D, size_t ld, double* Dnn, size_t ldnn, double* Din);
int dense_inverse(uint32_t m, const double* A, size_t lda, double* Ainv, size_t ldi);
int cheb_integration(uint32_t n, double* Dnn_inv, size_t ld, double* Din);
int ivp(uint32_t m, uint32_t s, const double* Din, const double* x_base, double* out, size_t ld);

// ---- fragment geometry shared by the host packer and the kernel (sri_apply.hip) ---------------
// One v_mfma_f64_16x16x4_f64 consumes a 16x4 slice of A.  Lane l feeds row-slot i = l & 15 and
// k-slot g = l >> 4, and receives output rows (g + 4*reg), reg = 0..3, of column l & 15.
// We permute the 16 rows of every tile so that the two registers (2q, 2q+1) of lane (c, g) are
// the ADJACENT rows 8q + 2g and 8q + 2g + 1: each lane then stores 16 contiguous bytes and the
// four g-lanes of a column cover one 64-byte segment of Y.
//   slot i = g + 4*reg = g + 8q + 4e   <->   row offset 8q + 2g + e
constexpr int kChunkK = 16;   // k extent of one pipeline stage (BK)

inline constexpr int row_of_slot(int i) { return 8 * (i >> 3) + 2 * (i & 3) + ((i >> 2) & 1); }
// k index inside a chunk for lane k-slot g, k-pair j, element e of the 16-byte fragment read:
inline constexpr int k_of(int j, int g, int e) { return 8 * j + 2 * g + e; }

}  // namespace sri

#endif  // SRI_INTERNAL_H
