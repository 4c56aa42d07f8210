// sri_apply_wide.hip — the 64 x 256 and 32 x 256 geometries of an operator's last part (GeoWide, GeoWide32) and the row blocks of an
// operator's own height (Geo96, Geo160, Geo192, Geo224): plain and rank-1-epilogue variants
#include "sri_apply_kernel.h"
#include "sri_apply_geometries.h"

namespace sri {

// GeoWide serves plain and rank-1-epilogue launches of an operator's last part only
template <class GW>
static int prepare_wide(int dev) {
  int rc = prepare<GW, true, false, false>(dev);
  if (rc == SRI_OK) rc = prepare<GW, true, true, false>(dev);
  if (rc == SRI_OK) rc = prepare<GW, true, false, true>(dev);
  if (rc == SRI_OK) rc = prepare<GW, true, false, false, EPI_NONE, 0, true>(dev);
  if (rc == SRI_OK) rc = prepare<GW, true, false, true, EPI_NONE, 0, true>(dev);
  return rc;
}

template <class GeoWide>
static int dispatch_wide(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream) {
  const size_t col_tiles = (args.m + GeoWide::BN - 1) / GeoWide::BN;
  const int cus = h->num_cus, dev = h->device;
  const uint32_t cap = h->max_workgroups;
  const bool tail = args.tail_k != 0;
  if (h->ascending) {
    if (rank1) return SRI_EUNSUPPORTED;
    return tail ? launch<GeoWide, true, false, true, EPI_NONE, 0, true>(args, col_tiles, dev, cus, cap, stream)
                : launch<GeoWide, true, false, false, EPI_NONE, 0, true>(args, col_tiles, dev, cus, cap, stream);
  }
  if (rank1) return tail ? SRI_EUNSUPPORTED : launch<GeoWide, true, true, false>(args, col_tiles, dev, cus, cap, stream);   // (run() keeps such handles whole)
  return tail ? launch<GeoWide, true, false, true>(args, col_tiles, dev, cus, cap, stream)
              : launch<GeoWide, true, false, false>(args, col_tiles, dev, cus, cap, stream);
}

int prepare_g96(int dev) { return prepare_wide<Geo96>(dev); }
int resident_g96(int dev) { return (dev < 0 || dev >= kMaxDevices) ? 0 : residency<Geo96, true, false, false>(dev).load(std::memory_order_acquire); }
int dispatch_g96(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream) { return dispatch_wide<Geo96>(h, args, rank1, stream); }
void pack_g96(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
              std::vector<double>& out) {
  pack_operator<Geo96>(A, lda, rows, kdim, nk, row_blocks, w, asc, out);
}
int prepare_g192(int dev) { return prepare_wide<Geo192>(dev); }
int resident_g192(int dev) { return (dev < 0 || dev >= kMaxDevices) ? 0 : residency<Geo192, true, false, false>(dev).load(std::memory_order_acquire); }
int dispatch_g192(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream) { return dispatch_wide<Geo192>(h, args, rank1, stream); }
void pack_g192(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
               std::vector<double>& out) {
  pack_operator<Geo192>(A, lda, rows, kdim, nk, row_blocks, w, asc, out);
}
int prepare_g224(int dev) { return prepare_wide<Geo224>(dev); }
int resident_g224(int dev) { return (dev < 0 || dev >= kMaxDevices) ? 0 : residency<Geo224, true, false, false>(dev).load(std::memory_order_acquire); }
int dispatch_g224(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream) { return dispatch_wide<Geo224>(h, args, rank1, stream); }
void pack_g224(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
               std::vector<double>& out) {
  pack_operator<Geo224>(A, lda, rows, kdim, nk, row_blocks, w, asc, out);
}
int prepare_g160(int dev) { return prepare_wide<Geo160>(dev); }
int resident_g160(int dev) { return (dev < 0 || dev >= kMaxDevices) ? 0 : residency<Geo160, true, false, false>(dev).load(std::memory_order_acquire); }
int dispatch_g160(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream) { return dispatch_wide<Geo160>(h, args, rank1, stream); }
void pack_g160(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
               std::vector<double>& out) {
  pack_operator<Geo160>(A, lda, rows, kdim, nk, row_blocks, w, asc, out);
}
int prepare_wide64(int dev) { return prepare_wide<GeoWide>(dev); }
int resident_wide64(int dev) { return (dev < 0 || dev >= kMaxDevices) ? 0 : residency<GeoWide, true, false, false>(dev).load(std::memory_order_acquire); }
int prepare_wide32(int dev) { return prepare_wide<GeoWide32>(dev); }
int resident_wide32(int dev) { return (dev < 0 || dev >= kMaxDevices) ? 0 : residency<GeoWide32, true, false, false>(dev).load(std::memory_order_acquire); }
int dispatch_wide64(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream) { return dispatch_wide<GeoWide>(h, args, rank1, stream); }
int dispatch_wide32(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream) { return dispatch_wide<GeoWide32>(h, args, rank1, stream); }
void pack_wide64(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
                 std::vector<double>& out) {
  pack_operator<GeoWide>(A, lda, rows, kdim, nk, row_blocks, w, asc, out);
}
void pack_wide32(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
                 std::vector<double>& out) {
  pack_operator<GeoWide32>(A, lda, rows, kdim, nk, row_blocks, w, asc, out);
}

}  // namespace sri
