// sri_apply_strip.hip — the 256-row geometry with a 16-row strip (GeoTallStrip: the last block of an operator of 256 a + r rows, r <= 16)
// and with the mini strip (GeoTallMini: r <= 4, on v_mfma_f64_4x4x4_4b_f64)
#include "sri_apply_kernel.h"
#include "sri_apply_geometries.h"

namespace sri {

// (the mini strip carries 1-4 rows behind a multiple of 256: its row count is never a multiple of 16, so only the odd-row-count
//  variants — TAIL, with the rank-1 term folded into the product — exist for it)
template <class GS>
static int prepare_strip_geometry(int dev) {
  int rc = SRI_OK;
  if constexpr (GS::STRIP == 1) {
    rc = prepare<GS, true, false, false>(dev);
    if (rc == SRI_OK) rc = prepare<GS, true, true, false>(dev);
    if (rc == SRI_OK) rc = prepare<GS, true, false, false, EPI_NONE, 0, true>(dev);
  }
  if (rc == SRI_OK) rc = prepare<GS, true, false, true>(dev);
  if (rc == SRI_OK) rc = prepare<GS, true, true, true>(dev);
  if (rc == SRI_OK) rc = prepare<GS, true, false, true, EPI_NONE, 0, true>(dev);
  if (rc == SRI_OK) rc = prepare<GS, true, false, true, EPI_NONE, 0, false, true>(dev);
  if (rc == SRI_OK) rc = prepare<GS, true, true, true, EPI_NONE, 0, false, true>(dev);
  if (rc == SRI_OK) rc = prepare<GS, true, false, true, EPI_NONE, 0, true, true>(dev);
  return rc;
}

template <class GS>
static int dispatch_strip_geometry(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream, bool lean) {
  const size_t col_tiles = (args.m + GS::BN - 1) / GS::BN;
  const int cus = h->num_cus, dev = h->device;
  const uint32_t cap = h->max_workgroups;
  const bool tail = args.tail_k != 0;
  if (args.row_blocks != 1) return SRI_EUNSUPPORTED;                 // the strip belongs to its part's only row block
  if (h->ascending) {
    if (rank1) return SRI_EUNSUPPORTED;
    if (tail && lean) return launch<GS, true, false, true, EPI_NONE, 0, true, true>(args, col_tiles, dev, cus, cap, stream);
    if (tail) return launch<GS, true, false, true, EPI_NONE, 0, true>(args, col_tiles, dev, cus, cap, stream);
    if constexpr (GS::STRIP == 1) return launch<GS, true, false, false, EPI_NONE, 0, true>(args, col_tiles, dev, cus, cap, stream);
    return SRI_EUNSUPPORTED;
  }
  if (tail && lean) return rank1 ? launch<GS, true, true, true, EPI_NONE, 0, false, true>(args, col_tiles, dev, cus, cap, stream)
                                 : launch<GS, true, false, true, EPI_NONE, 0, false, true>(args, col_tiles, dev, cus, cap, stream);
  if (tail) return rank1 ? launch<GS, true, true, true>(args, col_tiles, dev, cus, cap, stream)
                         : launch<GS, true, false, true>(args, col_tiles, dev, cus, cap, stream);
  if constexpr (GS::STRIP == 1) {
    return rank1 ? launch<GS, true, true, false>(args, col_tiles, dev, cus, cap, stream)
                 : launch<GS, true, false, false>(args, col_tiles, dev, cus, cap, stream);
  }
  return SRI_EUNSUPPORTED;
}


int prepare_strip(int dev) { return prepare_strip_geometry<GeoTallStrip>(dev); }
int resident_strip(int dev) { return (dev < 0 || dev >= kMaxDevices) ? 0 : residency<GeoTallStrip, true, false, false>(dev).load(std::memory_order_acquire); }
int dispatch_strip(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream, bool lean) {
  return dispatch_strip_geometry<GeoTallStrip>(h, args, rank1, stream, lean);
}
void pack_strip(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
                std::vector<double>& out) {
  pack_operator<GeoTallStrip>(A, lda, rows, kdim, nk, row_blocks, w, asc, out);
}

// the mini strip (GeoTallMini: at most four rows behind the last 256-row block, on v_mfma_f64_4x4x4_4b_f64)
int prepare_mini(int dev) { return prepare_strip_geometry<GeoTallMini>(dev); }
int resident_mini(int dev) { return (dev < 0 || dev >= kMaxDevices) ? 0 : residency<GeoTallMini, true, false, true>(dev).load(std::memory_order_acquire); }
int dispatch_mini(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream, bool lean) {
  return dispatch_strip_geometry<GeoTallMini>(h, args, rank1, stream, lean);
}
void pack_mini(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
               std::vector<double>& out) {
  pack_operator<GeoTallMini>(A, lda, rows, kdim, nk, row_blocks, w, asc, out);
}

// the 128-row geometry with the mini strip (GeoMidMini: 129 ... 132 rows)
int prepare_midmini(int dev) { return prepare_strip_geometry<GeoMidMini>(dev); }
int resident_midmini(int dev) { return (dev < 0 || dev >= kMaxDevices) ? 0 : residency<GeoMidMini, true, false, true>(dev).load(std::memory_order_acquire); }
int dispatch_midmini(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream, bool lean) {
  return dispatch_strip_geometry<GeoMidMini>(h, args, rank1, stream, lean);
}
void pack_midmini(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
                  std::vector<double>& out) {
  pack_operator<GeoMidMini>(A, lda, rows, kdim, nk, row_blocks, w, asc, out);
}

// the 64-row geometry with the mini strip (GeoFlatMini: 65 ... 68 rows)
int prepare_flatmini(int dev) { return prepare_strip_geometry<GeoFlatMini>(dev); }
int resident_flatmini(int dev) { return (dev < 0 || dev >= kMaxDevices) ? 0 : residency<GeoFlatMini, true, false, true>(dev).load(std::memory_order_acquire); }
int dispatch_flatmini(sri_apply_t* h, const ApplyArgs& args, bool rank1, hipStream_t stream, bool lean) {
  return dispatch_strip_geometry<GeoFlatMini>(h, args, rank1, stream, lean);
}
void pack_flatmini(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc,
                   std::vector<double>& out) {
  pack_operator<GeoFlatMini>(A, lda, rows, kdim, nk, row_blocks, w, asc, out);
}

}  // namespace sri
