// sri_apply_geometry_impl.h — what the three general geometries (GeoFlat, GeoMid, GeoTall) share: which variants of
// sri::apply_kernel a geometry instantiates (prepare_geometry), which one a call takes (dispatch), and the three entry points
// sri_apply_geometries.h declares for it.  Included by sri_apply_{flat,mid,tall}.hip only — one translation unit per geometry, so
// that the variants build in parallel; the geometries of an operator's LAST rows have their own, smaller lists
// (sri_apply_strip.hip, sri_apply_wide.hip).
#ifndef SRI_APPLY_GEOMETRY_IMPL_H
#define SRI_APPLY_GEOMETRY_IMPL_H

#include "sri_apply_kernel.h"
#include "sri_apply_geometries.h"

namespace sri {

template <class G>
static int prepare_geometry(int dev) {
  int rc = prepare<G, true, false, false>(dev);
  if (rc == SRI_OK) rc = prepare<G, true, true, false>(dev);
  if (rc == SRI_OK) rc = prepare<G, true, false, true>(dev);
  if (rc == SRI_OK) rc = prepare<G, true, true, true>(dev);
  if (rc == SRI_OK) rc = prepare<G, false, false, false>(dev);
  if (rc == SRI_OK) rc = prepare<G, false, true, false>(dev);
  if constexpr (G::BM >= 128) {                                 // ascending k order: only operators with rows > 512
    if (rc == SRI_OK) rc = prepare<G, true, false, false, EPI_NONE, 0, true>(dev);
    if (rc == SRI_OK) rc = prepare<G, true, false, true, EPI_NONE, 0, true>(dev);
  }
  if constexpr (G::QUAT_OK) {
    if (rc == SRI_OK) rc = prepare<G, true, true, false, EPI_QUAT, 2>(dev);
    if (rc == SRI_OK) rc = prepare<G, true, true, true, EPI_QUAT, 2>(dev);
  }
  if constexpr (G::BM >= 256 && G::STRIP == 0) {                // the closing product of a rod solve: Q and the position right-hand side
    if (rc == SRI_OK) rc = prepare<G, true, true, true, EPI_QB, 0>(dev);
    if (rc == SRI_OK) rc = prepare<G, true, true, true, EPI_QB, 0, false, true>(dev);
  }
  if constexpr (G::BM >= 128 && G::STRIP == 0 && G::QUAT_OK) {  // ... and the criterion of the fixed point (128-row tiles: only with it)
    if (rc == SRI_OK) rc = prepare<G, true, true, true, EPI_QBC, 2>(dev);
    if (rc == SRI_OK) rc = prepare<G, true, true, true, EPI_QBC, 2, false, true>(dev);
    if (rc == SRI_OK) rc = prepare<G, true, true, false, EPI_QBC, 2>(dev);           // whole k chunks (N = 16 q + 1 grids): the rank-1 term in the store stage
  }
  if constexpr (G::BM >= 128) {                                 // the lean twins of the odd-row-count variants (see run())
    if (rc == SRI_OK) rc = prepare<G, true, false, true, EPI_NONE, 0, false, true>(dev);
    if (rc == SRI_OK) rc = prepare<G, true, true, true, EPI_NONE, 0, false, true>(dev);
    if (rc == SRI_OK) rc = prepare<G, true, false, true, EPI_NONE, 0, true, true>(dev);
    if constexpr (G::QUAT_OK) {
      if (rc == SRI_OK) rc = prepare<G, true, true, true, EPI_QUAT, 2, false, true>(dev);
    }
  }
  if constexpr (G::QUAT_OK && G::lds_bytes_quat(4) + G::XB_BYTES <= 160 * 1024) {          // tiles of one or two k chunks only exist for rows <= 32
    if (rc == SRI_OK) rc = prepare<G, true, true, false, EPI_QUAT, 4>(dev);
    if (rc == SRI_OK) rc = prepare<G, true, true, true, EPI_QUAT, 4>(dev);
    if (rc == SRI_OK) rc = prepare<G, false, true, false, EPI_QUAT, 4>(dev);
  }
  return rc;
}

// `lean`: run() has established that the panel's last column may be read up to its last k chunk's end (LEAN twins exist for the
// 128- and 256-row geometries; the 64-row one is HBM-bound either way).
template <class G>
static int dispatch(sri_apply_t* h, const ApplyArgs& args, bool dma, bool rank1, hipStream_t stream, uint32_t* partials = nullptr, bool lean = false) {
  const size_t col_tiles = (args.m + G::BN - 1) / G::BN;
  const int cus = h->num_cus, dev = h->device;
  const uint32_t cap = h->max_workgroups;
  const bool tail = args.tail_k != 0 || (SRI_TAIL_ABL & 8) != 0;   // (ablation 8: the TAIL instantiation for every operator)
  if (args.y2 != nullptr) {                                       // EPI_QB / EPI_QBC (run() has checked that the handle is of that kind)
    if (!dma || !rank1 || h->ascending || (args.qe != nullptr) != (args.crit != nullptr) || (!tail && args.crit == nullptr)) return SRI_EUNSUPPORTED;
    if (args.crit != nullptr) {                                     // EPI_QBC: + the criterion of the fixed point
      if constexpr (G::BM >= 128 && G::STRIP == 0 && G::QUAT_OK) {
        if (args.nk < 3) return SRI_EUNSUPPORTED;                   // (two side buffers)
        if (!tail) return launch<G, true, true, false, EPI_QBC, 2>(args, col_tiles, dev, cus, cap, stream);
        return lean ? launch<G, true, true, true, EPI_QBC, 2, false, true>(args, col_tiles, dev, cus, cap, stream)
                    : launch<G, true, true, true, EPI_QBC, 2>(args, col_tiles, dev, cus, cap, stream);
      } else {
        return SRI_EUNSUPPORTED;
      }
    }
    if constexpr (G::BM >= 256 && G::STRIP == 0) {
      return lean ? launch<G, true, true, true, EPI_QB, 0, false, true>(args, col_tiles, dev, cus, cap, stream)
                  : launch<G, true, true, true, EPI_QB, 0>(args, col_tiles, dev, cus, cap, stream);
    } else {
      return SRI_EUNSUPPORTED;
    }
  }
  if constexpr (G::BM >= 128) {
    if (h->ascending) {                                           // rows > 512, plain apply (create_from_matrix)
      if (!dma || rank1 || args.qe != nullptr) return SRI_EUNSUPPORTED;
      if (tail && lean) return launch<G, true, false, true, EPI_NONE, 0, true, true>(args, col_tiles, dev, cus, cap, stream);
      return tail ? launch<G, true, false, true, EPI_NONE, 0, true>(args, col_tiles, dev, cus, cap, stream)
                  : launch<G, true, false, false, EPI_NONE, 0, true>(args, col_tiles, dev, cus, cap, stream);
    }
    if (dma && tail && lean) {
      if (args.qe != nullptr) {
        if constexpr (G::QUAT_OK) {
          if (args.nk >= 3) return launch<G, true, true, true, EPI_QUAT, 2, false, true>(args, col_tiles, dev, cus, cap, stream, partials);
        }
      } else {
        return rank1 ? launch<G, true, true, true, EPI_NONE, 0, false, true>(args, col_tiles, dev, cus, cap, stream)
                     : launch<G, true, false, true, EPI_NONE, 0, false, true>(args, col_tiles, dev, cus, cap, stream);
      }
    }
  }
  if (args.qe != nullptr) {                               // fused quaternion iteration (always with the rank-1 term)
    if constexpr (!G::QUAT_OK) {
      return SRI_EUNSUPPORTED;                             // (A/B geometries without room for the epilogue's side buffers)
    } else {
    if (args.nk >= 3) {
      if (dma && tail) return launch<G, true, true, true, EPI_QUAT, 2>(args, col_tiles, dev, cus, cap, stream, partials);
      if (dma) return launch<G, true, true, false, EPI_QUAT, 2>(args, col_tiles, dev, cus, cap, stream, partials);
      return SRI_EUNSUPPORTED;                             // register-staged build: rows < 16, one chunk
    }
    if constexpr (G::lds_bytes_quat(4) + G::XB_BYTES <= 160 * 1024) {
      if (dma && tail) return launch<G, true, true, true, EPI_QUAT, 4>(args, col_tiles, dev, cus, cap, stream, partials);
      if (dma) return launch<G, true, true, false, EPI_QUAT, 4>(args, col_tiles, dev, cus, cap, stream, partials);
      return launch<G, false, true, false, EPI_QUAT, 4>(args, col_tiles, dev, cus, cap, stream, partials);
    }
    return SRI_EUNSUPPORTED;
    }
  }
  if (dma && tail) return rank1 ? launch<G, true, true, true>(args, col_tiles, dev, cus, cap, stream)
                                : launch<G, true, false, true>(args, col_tiles, dev, cus, cap, stream);
  if (dma) return rank1 ? launch<G, true, true, false>(args, col_tiles, dev, cus, cap, stream)
                        : launch<G, true, false, false>(args, col_tiles, dev, cus, cap, stream);
  return rank1 ? launch<G, false, true, false>(args, col_tiles, dev, cus, cap, stream)
               : launch<G, false, false, false>(args, col_tiles, dev, cus, cap, stream);
}

}  // namespace sri

// the three entry points of one general geometry (sri_apply_geometries.h: SRI_GEOMETRY_DECL)
#define SRI_DEFINE_GEOMETRY(name, Geometry)                                                                                       \
  namespace sri {                                                                                                                  \
  int prepare_##name(int dev) { return prepare_geometry<Geometry>(dev); }                                                          \
  int resident_##name(int dev) { return (dev < 0 || dev >= kMaxDevices) ? 0 : residency<Geometry, true, false, false>(dev).load(std::memory_order_acquire); } \
  int dispatch_##name(sri_apply_t* h, const ApplyArgs& args, bool dma, bool rank1, hipStream_t stream, uint32_t* partials, bool lean) { \
    return dispatch<Geometry>(h, args, dma, rank1, stream, partials, lean);                                                        \
  }                                                                                                                                \
  void pack_##name(const double* A, size_t lda, uint32_t rows, uint32_t kdim, uint32_t nk, uint32_t row_blocks, const double* w, bool asc, \
                   std::vector<double>& out) {                                                                                     \
    pack_operator<Geometry>(A, lda, rows, kdim, nk, row_blocks, w, asc, out);                                                      \
  }                                                                                                                                \
  }

#endif  // SRI_APPLY_GEOMETRY_IMPL_H
