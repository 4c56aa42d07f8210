// sri_apply_geometries.h — what the per-geometry translation units (sri_apply_{flat,mid,tall,wide,strip}.hip) give sri_apply.hip:
// each compiles the variants of sri::apply_kernel for its geometry (so that they build in parallel) and exports how to
// prepare them on a device (attributes, residency), launch the right one for a call, and pack an operator in its fragment order.
#ifndef SRI_APPLY_GEOMETRIES_H
#define SRI_APPLY_GEOMETRIES_H

#include <vector>

namespace sri {

struct ApplyArgs;

#define SRI_GEOMETRY_DECL(name)                                                                                                    \
  int prepare_##name(int dev);                                                                                                     \
  int resident_##name(int dev);   /* workgroups of the plain variant one CU keeps resident, as prepare() measured (0: not prepared) */ \
  int dispatch_##name(sri_apply_t* h, const ApplyArgs& args, bool dma, bool rank1, hipStream_t stream, uint32_t* partials, bool lean); \
  void pack_##name(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc, \
                   std::vector<double>& out);
SRI_GEOMETRY_DECL(flat)
SRI_GEOMETRY_DECL(mid)
SRI_GEOMETRY_DECL(tall)
#undef SRI_GEOMETRY_DECL

int prepare_g96(int dev);
int dispatch_g96(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream);
void pack_g96(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
              std::vector<double>& out);
int prepare_g192(int dev);
int dispatch_g192(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream);
void pack_g192(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
               std::vector<double>& out);
int prepare_g224(int dev);
int dispatch_g224(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream);
void pack_g224(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
               std::vector<double>& out);
int prepare_g160(int dev);
int dispatch_g160(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream);
void pack_g160(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
               std::vector<double>& out);
int prepare_wide64(int dev);
int prepare_wide32(int dev);
int dispatch_wide64(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream);
int dispatch_wide32(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream);
void pack_wide64(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
                 std::vector<double>& out);
void pack_wide32(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
                 std::vector<double>& out);

int resident_g96(int dev);
int resident_g192(int dev);
int resident_g224(int dev);
int resident_g160(int dev);
int resident_wide64(int dev);
int resident_wide32(int dev);
int resident_strip(int dev);
int prepare_flatmini(int dev);
int resident_flatmini(int dev);
int dispatch_flatmini(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream, bool lean);
void pack_flatmini(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
                   std::vector<double>& out);
int prepare_midmini(int dev);
int resident_midmini(int dev);
int dispatch_midmini(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream, bool lean);
void pack_midmini(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
                  std::vector<double>& out);
int prepare_mini(int dev);
int resident_mini(int dev);
int dispatch_mini(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream, bool lean);
void pack_mini(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
               std::vector<double>& out);
int prepare_strip(int dev);
int dispatch_strip(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream, bool lean);
void pack_strip(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
                std::vector<double>& out);

}  // namespace sri

#endif  // SRI_APPLY_GEOMETRIES_H
