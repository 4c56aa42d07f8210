// sri_apply.hip — the apply handle and its C ABI: operator formation and packing (whole and in parts), launch policy, the
// entry points of include/sri_capi.h.  The kernel is csrc/sri_apply_kernel.h; its variants are compiled per geometry in
// sri_apply_{flat,mid,tall,wide,strip}.hip (sri_apply_geometries.h); host panels are sri_host_panels.hip.
//
// Replaces the dense products of the reference:
//   main.cpp:172   r_stack = Dn_NN_inv * (b_NN - ivp)          (A = Dn_NN^-1, integration)
//   main.cpp:97    (I_s (x) Dn) * chi  ==  vec(Dn * X)          (A = D_N, differentiation)
// There is no CPU fallback: without a gfx950 device every entry point fails with an error.
#include "sri_apply_kernel.h"
#include "sri_apply_geometries.h"
#include "sri_column_plan.h"

namespace sri {

static_assert(plan::FLAT == GEO_FLAT && plan::MID == GEO_MID && plan::TALL == GEO_TALL && plan::WIDE == GEO_WIDE &&
              plan::WIDE32 == GEO_WIDE32 && plan::TALL_STRIP == GEO_TALL_STRIP && plan::G96 == GEO_96 && plan::G192 == GEO_192 && plan::G160 == GEO_160 && plan::G224 == GEO_224 &&
              plan::TALL_MINI == GEO_TALL_MINI && plan::MID_MINI == GEO_MID_MINI && plan::FLAT_MINI == GEO_FLAT_MINI, "sri_column_plan.h uses GeoId's values");

static thread_local int g_last_hip_error = 0;
void record_hip_error(int hip_error) { g_last_hip_error = hip_error; }

static const char* kernel_name(int geo, bool dma, bool quat = false, bool lean = false, bool qb = false, bool crit = false) {
  // (the closing product of a rod solve, EPI_QB: Q and the position right-hand side from one launch; EPI_QBC: and the criterion — 256-row geometry)
  static const char* qb_names[4] = {"sri_apply_mfma_f64<256x128,dma,q+b>", "sri_apply_mfma_f64<256x128,dma,q+b,lean>",
                                    "sri_apply_mfma_f64<256x128,dma,q+b+criterion>", "sri_apply_mfma_f64<256x128,dma,q+b+criterion,lean>"};
  static const char* qbc_mid_names[2] = {"sri_apply_mfma_f64<128x128,dma,q+b+criterion>", "sri_apply_mfma_f64<128x128,dma,q+b+criterion,lean>"};
  if (qb && geo == GEO_TALL) return qb_names[(lean ? 1 : 0) + (crit ? 2 : 0)];
  if (qb && crit && geo == GEO_MID) return qbc_mid_names[lean ? 1 : 0];
  static const char* names[3][4] = {
      {"sri_apply_mfma_f64<64x64,reg>", "sri_apply_mfma_f64<64x64,dma>", "sri_apply_mfma_f64<64x64,reg,quat>",
       "sri_apply_mfma_f64<64x64,dma,quat>"},
      {"sri_apply_mfma_f64<128x128,reg>", "sri_apply_mfma_f64<128x128,dma>", "sri_apply_mfma_f64<128x128,reg,quat>",
       "sri_apply_mfma_f64<128x128,dma,quat>"},
      {"sri_apply_mfma_f64<256x128,reg>", "sri_apply_mfma_f64<256x128,dma>", "sri_apply_mfma_f64<256x128,reg,quat>",
       "sri_apply_mfma_f64<256x128,dma,quat>"}};
  // (the lean twins of the odd-row-count variants: 128- and 256-row geometries, LDS-DMA builds)
  static const char* lean_names[3][2] = {{"", ""},
                                         {"sri_apply_mfma_f64<128x128,dma,lean>", "sri_apply_mfma_f64<128x128,dma,quat,lean>"},
                                         {"sri_apply_mfma_f64<256x128,dma,lean>", "sri_apply_mfma_f64<256x128,dma,quat,lean>"}};
  if (lean && dma && geo != GEO_FLAT) return lean_names[geo][quat ? 1 : 0];
  return names[geo][(dma ? 1 : 0) + (quat ? 2 : 0)];
}

struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  int enter(int dev) {
    SRI_HIP(hipGetDevice(&prev));
    if (prev != dev) {
      SRI_HIP(hipSetDevice(dev));
      switched = true;
    }
    return SRI_OK;
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

static int check_device(int device, int* num_cus = nullptr) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    g_last_hip_error = static_cast<int>(e);
    (void)hipGetLastError();
    return SRI_ENODEVICE;
  }
  if (device < 0 || device >= count) return SRI_ENODEVICE;
  hipDeviceProp_t prop;
  SRI_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return SRI_ENODEVICE;   // gfx950-only code object
  if (num_cus) *num_cus = prop.multiProcessorCount;
  return SRI_OK;
}

// `rows`: the operator's column count = the row count of the panels it is applied to (k); `rows_out` (0: square): its row count
// = the row count of the result.  Rectangular operators are internal (the rod pipeline's interpolation, §7): one geometry, no parts.
static int create_from_matrix(sri_apply_t** out, uint32_t rows, const double* A, size_t lda, const double* w_host,
                              int which, int device, uint32_t rows_out = 0) {
  if (rows_out == 0) rows_out = rows;
  const bool square = rows_out == rows;
  if (out == nullptr || A == nullptr || rows < 1 || lda < rows_out) return SRI_EINVAL;
  *out = nullptr;
  int num_cus = 0;
  int rc = check_device(device, &num_cus);
  if (rc != SRI_OK) return rc;
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc != SRI_OK) return rc;

  sri_apply_t* h = new (std::nothrow) sri_apply_handle();
  if (h == nullptr) return SRI_ENOMEM;
  h->device = device;
  h->num_cus = num_cus;
  h->rows = rows;
  h->rows_out = rows_out;
  h->which = which;
  // Rows are tiled in blocks of 64 / 128 / 256 and a partly empty block costs a full one.  Above 256 rows the 128-row geometry
  // (0.77 of the MFMA peak at K = 128, 0.86 from K = 384 on: tile ends amortise over K) is the better choice whenever it
  // saves a half-empty 256-row block, i.e. when the last 256-row block would hold at most 128 rows — measured N = 320:
  // 0.54 -> 0.71, N = 384: 0.66 -> 0.86 of the peak (profiles/r03_sweep_n.jsonl).
  h->geo = plan::own_geometry(rows_out);                          // (sri_column_plan.h)
  // (tiles of one or two k chunks: the fused step needs four side buffers, which the 256-row geometry's ring leaves no room for)
  if (!square && h->geo == GEO_TALL && (rows + BK - 1) / BK < 3) h->geo = GEO_MID;
  // SRI_DEBUG=force_geo=m / f (A/B timing only: profiles/r04_geometry_by_columns.log): the 128 x 128 / 64 x 64 geometry as the
  // operator's own, whatever its rows
  {
    const std::string v = debug_option("force_geo");
    if (v == "m") h->geo = GEO_MID; else if (v == "f") h->geo = GEO_FLAT;
  }
  // The differentiation operators of large grids (and caller-supplied matrices of that size) sum k in the reference's
  // ascending order; the integration operator (||Dn_NN^-1||_inf = 1: no cancellation to protect, and the one the rank-1
  // and fused-step variants are built on) keeps the faster paired order at every size.  The ascending variant exists for
  // the two larger geometries — what is reported is always what pack_operator() and dispatch() use.
  h->ascending = square && h->geo != GEO_FLAT && rows > kAscendingAboveRows && which != SRI_OP_DNN_INV;
  h->bm = h->geo == GEO_FLAT ? GeoFlat::BM : (h->geo == GEO_MID ? GeoMid::BM : GeoTall::BM);
  h->nk = (rows + BK - 1) / BK;
  h->row_blocks = (rows_out + h->bm - 1) / h->bm;
  rc = h->geo == GEO_FLAT ? prepare_flat(device) : (h->geo == GEO_MID ? prepare_mid(device) : prepare_tall(device));
  if (rc != SRI_OK) {
    delete h;
    return rc;
  }

  std::vector<double> packed;
  try {
    const double* wf = (rows % BK != 0) ? w_host : nullptr;
    if (h->geo == GEO_FLAT) pack_flat(A, lda, rows_out, rows, h->nk, h->row_blocks, wf, false, packed);
    else if (h->geo == GEO_MID) pack_mid(A, lda, rows_out, rows, h->nk, h->row_blocks, wf, h->ascending, packed);
    else pack_tall(A, lda, rows_out, rows, h->nk, h->row_blocks, wf, h->ascending, packed);
  } catch (const std::bad_alloc&) {
    delete h;
    return SRI_ENOMEM;
  }
  auto fail = [&](int code) {
    if (h->d_apack) (void)hipFree(h->d_apack);
    if (h->d_apack_small) (void)hipFree(h->d_apack_small);
    if (h->d_w) (void)hipFree(h->d_w);
    for (sri_apply_handle::Part& part : h->parts)
      if (part.d_apack) (void)hipFree(part.d_apack);
    for (sri_apply_handle::Alternate& alt : h->alternates)
      if (alt.d_apack) (void)hipFree(alt.d_apack);
    delete h;
    return code;
  };
  // (SRI_OPT_SMALL_KERNEL = 0 keeps a 64-row operator on the general kernel per handle: run())
  if (square && small_supported(rows)) {
    rc = small_prepare(device);
    if (rc != SRI_OK) return fail(rc);
    std::vector<double> sp;
    try {
      small_pack(A, lda, sp);
    } catch (const std::bad_alloc&) {
      return fail(SRI_ENOMEM);
    }
    if (hipMalloc(&h->d_apack_small, sp.size() * sizeof(double)) != hipSuccess) return fail(SRI_ENOMEM);
    if (hipMemcpy(h->d_apack_small, sp.data(), sp.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(SRI_EHIP);
  }
  if (hipMalloc(&h->d_apack, packed.size() * sizeof(double)) != hipSuccess) return fail(SRI_ENOMEM);
  if (hipMemcpy(h->d_apack, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
    return fail(SRI_EHIP);
  // Parts (round 4).  Rows are tiled in blocks of 256 / 128 / 64 and a partly empty block costs a full one; with one geometry
  // per operator a row count of 64 U - r (U 64-row units) is padded to a whole number of 128- or 256-row blocks.  When
  // U = 4a + 1 (rows 257-320, 513-576, 1025-1088, ...) the first 256a rows go to the 256-row geometry and the last 1-64 rows
  // to a 64 x 256 part of their own (GeoWide); when U = 3 (129-192 rows) the first 128 rows go to the 128-row geometry.  Each
  // part is a launch of the same kernel on the rows it owns (ApplyArgs::rows_out; all k): same k order, same chain of fused
  // multiply-adds per output element — bit-identical to the single-geometry result.  The second launch re-reads the panel
  // (from the Infinity Cache as far as it reaches), which is why thinner last parts do not pay (DESIGN.md §4 "Other grid sizes").
  // SRI_OPT_ROW_PARTS = 0 launches an operator whole although it is kept in parts (A/B, tests: run()).
  {
    int np = 0;
    if (square) {
      plan::RowPart rp[2];
      // A/B only (SRI_DEBUG): strip=0 — thin last rows as a part of their own instead of a strip; parts_ext=0 — 256 a + 65 ... 96 /
      // 129 ... 192 rows stay whole
      np = plan::row_parts(rows, debug_option("strip") != "0", rp, debug_option("parts_ext") != "0", debug_option("mini") != "0");   // (sri_column_plan.h; tests/cpp/column_plan_check.cpp)
      for (int i = 0; i < np; ++i) {
        h->parts[i].geo = rp[i].geo;
        h->parts[i].row0 = rp[i].row0;
        h->parts[i].rows = rp[i].rows;
        h->parts[i].row_blocks = rp[i].row_blocks;
      }
    }
    if (np != 0) {
      static const char* const geo_names[13] = {"<64x64,dma>", "<128x128,dma>", "<256x128,dma>", "<64x256,dma>", "<32x256,dma>", "<256+16x128,dma>", "<96x128,dma>", "<192x64,dma>", "<160x64,dma>", "<224x64,dma>", "<256+4x128,dma>", "<128+4x128,dma>", "<64+4x64,dma>"};
      const double* wf = (rows % BK != 0) ? w_host : nullptr;
      h->parts_label = "sri_apply_mfma_f64";
      for (int i = 0; i < np && rc == SRI_OK; ++i) {
        sri_apply_handle::Part& part = h->parts[i];
        switch (part.geo) {
          case GEO_TALL: rc = prepare_tall(device); break;
          case GEO_MID: rc = prepare_mid(device); break;
          case GEO_WIDE: rc = prepare_wide64(device); break;
          case GEO_WIDE32: rc = prepare_wide32(device); break;
          case GEO_96: rc = prepare_g96(device); break;
          case GEO_192: rc = prepare_g192(device); break;
          case GEO_160: rc = prepare_g160(device); break;
          case GEO_224: rc = prepare_g224(device); break;
          case GEO_TALL_MINI: rc = prepare_mini(device); break;
          case GEO_MID_MINI: rc = prepare_midmini(device); break;
          case GEO_FLAT_MINI: rc = prepare_flatmini(device); break;
          default: rc = prepare_strip(device); break;
        }
        if (rc != SRI_OK) break;
        std::vector<double> pk;
        try {
          const double* Ap = A + part.row0;
          const double* wp = wf != nullptr ? wf + part.row0 : nullptr;
          switch (part.geo) {
            case GEO_TALL: pack_tall(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
            case GEO_MID: pack_mid(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
            case GEO_WIDE: pack_wide64(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
            case GEO_WIDE32: pack_wide32(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
            case GEO_96: pack_g96(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
            case GEO_192: pack_g192(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
            case GEO_160: pack_g160(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
            case GEO_224: pack_g224(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
            case GEO_TALL_MINI: pack_mini(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
            case GEO_MID_MINI: pack_midmini(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
            case GEO_FLAT_MINI: pack_flatmini(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
            default: pack_strip(Ap, lda, part.rows, rows, h->nk, part.row_blocks, wp, h->ascending, pk); break;
          }
          h->parts_label += (i == 0 ? "" : "+");
          h->parts_label += geo_names[part.geo];
        } catch (const std::bad_alloc&) {
          return fail(SRI_ENOMEM);
        }
        if (hipMalloc(&part.d_apack, pk.size() * sizeof(double)) != hipSuccess) return fail(SRI_ENOMEM);
        if (hipMemcpy(part.d_apack, pk.data(), pk.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(SRI_EHIP);
        if (part.geo == GEO_WIDE || part.geo == GEO_WIDE32 || part.geo == GEO_96 || part.geo == GEO_192 || part.geo == GEO_160 || part.geo == GEO_224) h->parts_have_wide = true;
      }
      if (rc != SRI_OK) return fail(rc);
      h->num_parts = np;
      h->parts_name = h->parts_label.c_str();
      // (",lean": a part in a geometry that has lean twins took one — the 96 / 160 / 192-row blocks and the wide parts have none)
      bool any_lean_twin = false;
      for (int i = 0; i < np; ++i) any_lean_twin = any_lean_twin || h->parts[i].geo == GEO_TALL || h->parts[i].geo == GEO_MID || h->parts[i].geo == GEO_TALL_STRIP || h->parts[i].geo == GEO_TALL_MINI || h->parts[i].geo == GEO_MID_MINI || h->parts[i].geo == GEO_FLAT_MINI;
      h->parts_label_lean = h->parts_label + (any_lean_twin ? ",lean" : "");
      h->parts_name_lean = h->parts_label_lean.c_str();
    }
  }
  h->crit_on_mid = debug_option("criterion_geometry") != "own";
  // Alternates (sri_column_plan.h): the whole operator once more in each finer geometry — 128 x 128 and 64 x 64 tiles for a
  // 256-row operator, 64 x 64 for a 128-row one (the 64 x 64 geometry has no ascending-k variant: not for operators that use it).
  if (square && rows >= BK && h->geo != GEO_FLAT) {
    const double* wf = (rows % BK != 0) ? w_host : nullptr;
    for (int geo = h->geo - 1; geo >= GEO_FLAT; --geo) {
      if (geo == GEO_FLAT && h->ascending) break;
      sri_apply_handle::Alternate& alt = h->alternates[h->num_alternates];
      rc = geo == GEO_MID ? prepare_mid(device) : prepare_flat(device);
      if (rc != SRI_OK) return fail(rc);
      const uint32_t bm = geo == GEO_MID ? GeoMid::BM : GeoFlat::BM;
      alt.geo = geo;
      alt.row_blocks = (rows + bm - 1) / bm;
      std::vector<double> pk;
      try {
        if (geo == GEO_MID) pack_mid(A, lda, rows, rows, h->nk, alt.row_blocks, wf, h->ascending, pk);
        else pack_flat(A, lda, rows, rows, h->nk, alt.row_blocks, wf, false, pk);
      } catch (const std::bad_alloc&) {
        return fail(SRI_ENOMEM);
      }
      if (hipMalloc(&alt.d_apack, pk.size() * sizeof(double)) != hipSuccess) return fail(SRI_ENOMEM);
      ++h->num_alternates;                                         // (counted once it owns memory: fail() frees by pointer anyway)
      if (hipMemcpy(alt.d_apack, pk.data(), pk.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(SRI_EHIP);
    }
    for (int l = 0; l < 2; ++l)
      for (int a = 0; a < h->num_alternates; ++a) {
        const std::string primary = h->num_parts != 0 ? (l ? h->parts_label_lean : h->parts_label)
                                                      : std::string(kernel_name(h->geo, true, false, l != 0));
        h->mixed_labels[l][a] = primary + (h->alternates[a].geo == GEO_MID ? "|<128x128,dma>" : "|<64x64,dma>");
      }
  }
  if (w_host != nullptr) {
    size_t padded = static_cast<size_t>(h->row_blocks) * h->bm;
    if (h->num_parts != 0) padded = std::max<size_t>(padded, static_cast<size_t>(rows) + 256);
    if (hipMalloc(&h->d_w, padded * sizeof(double)) != hipSuccess) return fail(SRI_ENOMEM);
    if (hipMemset(h->d_w, 0, padded * sizeof(double)) != hipSuccess) return fail(SRI_EHIP);                  // (zero padding)
    if (hipMemcpy(h->d_w, w_host, rows_out * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(SRI_EHIP);
  }
  *out = h;
  return SRI_OK;
}

constexpr int kWhichInterpolation = -3;   // sri_apply_handle::which of create_interpolation_operator()

static int measured_residency(int geo, int dev);

struct ManyPanels {
  const double* const* xs;   // device array of `count` device pointers
  double* const* ys;
  size_t count;
};

// index of the handle's 128-row packing if its own geometry is the 256-row one and EPI_QBC should run there instead (-1: no;
// SRI_DEBUG=criterion_geometry=own at handle creation keeps the own geometry: A/B)
static int criterion_alternate(const sri_apply_t* h) {
  if (h->geo != GEO_TALL || !h->crit_on_mid) return -1;
  for (int a = 0; a < h->num_alternates; ++a)
    if (h->alternates[a].geo == GEO_MID) return a;
  return -1;
}

struct PositionOut {         // integrate_with_position_rhs: the second panel of an EPI_QB launch
  double* b;
  size_t ld;
  const FixedPointCriterion* crit = nullptr;   // ... and, with it, EPI_QBC
};

static int run(sri_apply_t* h, const double* dX, size_t ldx, double* dY, size_t ldy, size_t M, const double* d_xb,
               hipStream_t stream, double* d_tips = nullptr, const QuatEpilogue* quat = nullptr, uint32_t* partials = nullptr,
               const ManyPanels* many = nullptr, const PositionOut* pos = nullptr) {
  if (h == nullptr || ldx < h->rows || ldy < h->rows_out) return SRI_EINVAL;
  if (pos != nullptr) {
    if (pos->b == nullptr || quat != nullptr || many != nullptr || d_tips != nullptr || d_xb == nullptr || M % 4 != 0 ||
        pos->ld < h->rows_out + (h->rows_out & 1u))
      return SRI_EINVAL;
    // (the variant exists for odd-row-count operators in the 256-row geometry, paired k order, LDS-DMA path: in 128 x 128 tiles —
    //  N = 128, and 257 ... 384 rows — three more stores per rod and row pair at every tile end cost what the lighter pass saves:
    //  profiles/r05_rod_ab_closing_product.log)
    // (with the criterion — EPI_QBC — the whole confirmation pass goes, and 128-row tiles take part: their two workgroups per CU cover
    //  each other's tile ends)
    const bool geo_ok = h->geo == GEO_TALL || (h->geo == GEO_MID && pos->crit != nullptr);   // (own geometry; the launch itself: primary())
    // (whole k chunks — rows % 16 == 0, the N = 16 q + 1 grids — only with the criterion: that variant applies the rank-1 term itself)
    if ((h->rows % 16 == 0 && pos->crit == nullptr) || h->rows < 16 || !geo_ok || h->ascending || h->rows != h->rows_out) return SRI_EUNSUPPORTED;
    if (pos->crit != nullptr) {
      const FixedPointCriterion& cr = *pos->crit;
      if (cr.qe == nullptr || cr.leg1 == nullptr || cr.words == nullptr) return SRI_EINVAL;
      // (the criterion re-reads X as the accumulators sit: 16-byte pairs of rows, the last of which may reach one row past an odd row count)
      if (ldx % 2 != 0 || reinterpret_cast<uintptr_t>(dX) % 16 != 0 || ldx < h->rows + (h->rows & 1u) || h->nk < 3) return SRI_EUNSUPPORTED;
    }
  }
  if (many == nullptr && M > 0 && (dX == nullptr || dY == nullptr)) return SRI_EINVAL;
  if (many != nullptr && (many->xs == nullptr || many->ys == nullptr || d_xb != nullptr || d_tips != nullptr || quat != nullptr)) return SRI_EINVAL;
  if (M == 0 || (many != nullptr && many->count == 0)) return SRI_OK;
  if (many != nullptr && (many->count > 0xffffffffull || h->rows < 16)) return SRI_EUNSUPPORTED;
  const bool rank1 = d_xb != nullptr;
  if (rank1 && h->d_w == nullptr) return SRI_EINVAL;
  if (quat != nullptr && (!rank1 || M % 4 != 0 || quat->qe == nullptr || quat->leg1 == nullptr))
    return SRI_EINVAL;
  DeviceGuard guard;
  int rc = guard.enter(h->device);
  if (rc != SRI_OK) return rc;

  if (h->d_apack_small != nullptr && h->use_small && !rank1 && quat == nullptr && many == nullptr && small_batch(M, h->num_cus, h->max_workgroups)) {
    h->last_kernel.store("sri_apply_mfma_f64<64x64,dma,one-tile-per-workgroup>", std::memory_order_relaxed);
    return small_launch(h->d_apack_small, dX, ldx, dY, ldy, M, d_tips, h->device, h->num_cus, h->max_workgroups, stream);
  }

  ApplyArgs args;
  args.apack = h->d_apack;
  args.x = dX;
  args.y = dY;
  args.w = h->d_w;
  args.xb = d_xb;
  args.tips = d_tips;
  const FixedPointCriterion* const crit = pos ? pos->crit : nullptr;
  args.qe = quat ? quat->qe : (crit ? crit->qe : nullptr);
  args.leg1 = quat ? quat->leg1 : (crit ? crit->leg1 : nullptr);
  args.crit = crit ? crit->words : nullptr;
  args.crit_clear = crit ? crit->clear : nullptr;
  args.crit_words = crit ? kCriterionWords : 0u;
  args.side_bufs = 0;   // set by launch()
  args.tips_prev = quat ? quat->tips_prev : nullptr;
  args.partial = quat ? quat->partial : nullptr;
  args.stop = quat ? quat->stop : nullptr;
  args.fold = quat ? quat->fold : nullptr;
  args.fold_n = quat ? quat->fold_n : 0u;
  args.iteration = quat ? quat->iteration : 0u;
  args.fold_tol = quat ? quat->fold_tol : 0.0;
  args.fold_max_iter = quat ? quat->fold_max_iter : 0ull;
  args.state = quat ? quat->state : nullptr;
  if (quat != nullptr && quat->state != nullptr && quat->fold_n != 0u && quat->fold == nullptr) return SRI_EINVAL;
  if (quat != nullptr && (quat->partial != nullptr) != (quat->tips_prev != nullptr)) return SRI_EINVAL;
  if (quat != nullptr && quat->partial != nullptr && d_tips == nullptr) return SRI_EINVAL;
  args.ldx = ldx;
  args.ldy = ldy;
  args.m = M;
  args.rows = h->rows;
  args.rows_out = h->rows_out;
  args.nk = h->nk;
  args.row_blocks = h->row_blocks;
  args.col_tiles = 0;   // set by launch()
  args.tile_order = h->xcd_walk ? 1u : 0u;   // in: the XCD-grouped walk is allowed; launch() decides whether this grid takes it
  args.xs = many ? many->xs : nullptr;
  args.ys = many ? many->ys : nullptr;
  args.panels = many ? static_cast<uint32_t>(many->count) : 0u;
  args.stamps = nullptr;
  args.y2 = pos ? pos->b : nullptr;
  args.ldy2 = pos ? pos->ld : 0;

  // LDS-DMA (16 bytes per lane) and 16-byte stores work at 8-byte alignment on gfx950 at ~96 % of the aligned
  // rate (tools/unaligned_probe.hip), so any fp64 panel takes the DMA path; a row count that is not a multiple
  // of 16 is handled by the tail fix-up.  Only tiny operators (< 16 rows: no full chunk to fall back on) use
  // the register-staged build.
  const bool dma = h->rows >= 16;
  // the LDS-DMA pieces address a tile as a uniform 64-bit base + a 32-bit lane offset: 128 columns must span < 4 GiB
  if (dma && (static_cast<unsigned long long>(std::max(ldx, ldy)) * 8ull * 128ull + static_cast<unsigned long long>(std::max(h->rows, h->rows_out)) * 8ull) >= (1ull << 32))
    return SRI_EUNSUPPORTED;
  args.tail_k = h->rows % 16;
  // (many panels: their bases are only known on the device — plain stores)
  args.nt_stores = (many == nullptr && ldy % 16 == 0 && reinterpret_cast<uintptr_t>(dY) % 128 == 0) ? 1u : 0u;
  // The odd-row-count kernels fetch the last k chunk of every column where it lies: for the panel's LAST column that reaches up
  // to 16 - rows % 16 elements past the column's end.  The edge-safe variants step those pieces back and rebuild the image from
  // bounds-checked loads — logic that is dormant in every tile but one and still costs the instantiation 12-30 spilled scalar
  // registers and 1.6 % on identical work (round 4: profiles/r04_ab_tail_lean.log).  Memory protection is page-granular (4 KiB at
  // least): if the bytes past the end lie in the SAME page as the panel's last valid byte they can be read, whatever they hold
  // (tail_fix_fast overwrites their LDS image before any MFMA sees it) — then, and only then, the LEAN twin runs (same bits).
  const bool may_lean = dma && args.tail_k != 0 && many == nullptr && h->lean_tail;
  // columns [col0, col0 + cols) of the panel: may their launch take the lean twin?
  auto lean_for = [&](size_t col0, size_t cols) {
    if (!may_lean) return false;
    if (col0 + cols < M) return true;                               // (the bytes behind an interior column are the panel's own)
    const uintptr_t end = reinterpret_cast<uintptr_t>(dX + (M - 1) * ldx + h->rows);      // one past the last valid byte
    const uintptr_t over = (16u - args.tail_k) * sizeof(double);
    return ((end - 1) >> 12) == ((end + over - 1) >> 12);
  };
  auto columns = [&](size_t col0, size_t cols) {                      // the launch arguments for that range of columns
    ApplyArgs a = args;
    if (many == nullptr) {
      a.x = dX + col0 * ldx;
      a.y = dY + col0 * ldy;
      if (pos != nullptr) a.y2 = pos->b + col0 / 4 * 3 * pos->ld;
      a.nt_stores = (ldy % 16 == 0 && reinterpret_cast<uintptr_t>(a.y) % 128 == 0) ? 1u : 0u;
    }
    if (d_xb != nullptr) a.xb = d_xb + col0;
    if (d_tips != nullptr) a.tips = d_tips + col0;
    a.m = cols;
    return a;
  };
  // Parts: plain and rank-1-EPILOGUE launches only (the rank-1 term folded into the product needs the x_base side buffer, which
  // the wide geometry has no room for beside two workgroups per CU; the fused step and many-panel launches keep one geometry).
  const bool in_parts = h->num_parts != 0 && h->use_parts && dma && pos == nullptr && quat == nullptr && many == nullptr && !(h->parts_have_wide && rank1 && args.tail_k != 0) &&
      static_cast<unsigned long long>(std::max(ldx, ldy)) * 8ull * GeoWide::BN + static_cast<unsigned long long>(h->rows) * 8ull < (1ull << 32);
  // EPI_QBC's tile end is three times the plain one.  In the 256-row geometry (one workgroup per CU, two wave groups coupled by a
  // barrier per stage) the partner group waits it out; in 128 x 128 tiles two INDEPENDENT workgroups share a CU and cover each other's
  // tile ends — so the closing product with the criterion runs on the handle's 128-row packing of the same operator where it has one
  // (the column plan's alternate; same bits): N = 256 solve 1.50 -> 1.42 ms, N = 512 1.17 -> 1.11 (profiles/r05_rod_ab_criterion_geometry.log).
  const int crit_alt = criterion_alternate(h);
  // the operator's own path over `a.m` columns
  auto primary = [&](const ApplyArgs& cols_args, bool lean) -> int {
    if (crit != nullptr && crit_alt >= 0) {
      ApplyArgs a = cols_args;
      a.apack = h->alternates[crit_alt].d_apack;
      a.row_blocks = h->alternates[crit_alt].row_blocks;
      h->last_kernel.store(kernel_name(GEO_MID, dma, false, lean, true, true), std::memory_order_relaxed);
      return dispatch_mid(h, a, dma, rank1, stream, nullptr, lean);
    }
    if (in_parts) {
      for (int i = 0; i < h->num_parts; ++i) {
        const sri_apply_handle::Part& part = h->parts[i];
        ApplyArgs a = cols_args;
        a.apack = part.d_apack;
        a.y = cols_args.y + part.row0;
        a.w = h->d_w != nullptr ? h->d_w + part.row0 : nullptr;
        a.rows_out = part.rows;
        a.row_blocks = part.row_blocks;
        a.tips = part.row0 == 0 ? cols_args.tips : nullptr;         // the tip node is row 0
        a.nt_stores = (ldy % 16 == 0 && reinterpret_cast<uintptr_t>(a.y) % 128 == 0) ? 1u : 0u;
        const int prc = part.geo == GEO_224 ? dispatch_g224(h, a, rank1, stream)
                        : part.geo == GEO_160 ? dispatch_g160(h, a, rank1, stream)
                        : part.geo == GEO_192 ? dispatch_g192(h, a, rank1, stream)
                        : part.geo == GEO_96 ? dispatch_g96(h, a, rank1, stream)
                        : part.geo == GEO_WIDE ? dispatch_wide64(h, a, rank1, stream)
                        : part.geo == GEO_WIDE32 ? dispatch_wide32(h, a, rank1, stream)
                        : part.geo == GEO_TALL_STRIP ? dispatch_strip(h, a, rank1, stream, lean)
                        : part.geo == GEO_TALL_MINI ? dispatch_mini(h, a, rank1, stream, lean)
                        : part.geo == GEO_MID_MINI ? dispatch_midmini(h, a, rank1, stream, lean)
                        : part.geo == GEO_FLAT_MINI ? dispatch_flatmini(h, a, rank1, stream, lean)
                                             : (part.geo == GEO_MID ? dispatch_mid(h, a, dma, rank1, stream, nullptr, lean)
                                                                    : dispatch_tall(h, a, dma, rank1, stream, nullptr, lean));
        if (prc != SRI_OK) return prc;
      }
      return SRI_OK;
    }
    if (h->geo == GEO_FLAT) return dispatch_flat(h, cols_args, dma, rank1, stream, partials, false);
    if (h->geo == GEO_MID) return dispatch_mid(h, cols_args, dma, rank1, stream, partials, lean);
    return dispatch_tall(h, cols_args, dma, rank1, stream, partials, lean);
  };

  // Which columns go where (sri_column_plan.h): launches of a full grid on one panel.  The fused step keeps ONE geometry per launch
  // (its convergence monitors belong to the workgroups of one grid) — its own, or a finer one for all columns of a small batch.
  plan::Choice choice;
  choice.primary_cols = M;
  if (h->num_alternates != 0 && h->column_plan && dma && many == nullptr && h->max_workgroups == 0 && pos == nullptr) {
    // (the residency of every geometry is what prepare() measured on THIS device, not the table's figure)
    plan::Launch own[2], finer[2];
    int num_own = 0;
    if (in_parts) {
      for (int i = 0; i < h->num_parts; ++i) own[num_own++] = plan::Launch{h->parts[i].geo, h->parts[i].row_blocks, measured_residency(h->parts[i].geo, h->device)};
    } else {
      own[num_own++] = plan::Launch{h->geo, h->row_blocks, measured_residency(h->geo, h->device)};
    }
    for (int a = 0; a < h->num_alternates; ++a)
      finer[a] = plan::Launch{h->alternates[a].geo, h->alternates[a].row_blocks, measured_residency(h->alternates[a].geo, h->device)};
    choice = plan::choose(own, num_own, finer, h->num_alternates, M, h->nk, h->num_cus, quat == nullptr);
  }
  // the launches of a choice, on `stream`
  auto execute = [&](const plan::Choice& ch) -> int {
    if (ch.alt < 0) {
      const bool lean = lean_for(0, M);
      h->last_kernel.store(in_parts ? (lean ? h->parts_name_lean : h->parts_name)
                                    : kernel_name(h->geo, dma, quat != nullptr, lean && (quat == nullptr || args.nk >= 3), pos != nullptr, crit != nullptr), std::memory_order_relaxed);
      return primary(many == nullptr ? columns(0, M) : args, lean);
    }
    const sri_apply_handle::Alternate& alt = h->alternates[ch.alt];
    const size_t kept = ch.primary_cols;
    if (kept != 0) {
      const bool lean = lean_for(0, kept);
      const int prc = primary(columns(0, kept), lean);
      if (prc != SRI_OK) return prc;
      h->last_kernel.store(h->mixed_labels[lean ? 1 : 0][ch.alt].c_str(), std::memory_order_relaxed);
    }
    ApplyArgs rest = columns(kept, M - kept);
    rest.apack = alt.d_apack;
    rest.row_blocks = alt.row_blocks;
    const bool rest_lean = lean_for(kept, M - kept) && alt.geo != GEO_FLAT;
    if (kept == 0) h->last_kernel.store(kernel_name(alt.geo, true, quat != nullptr, rest_lean && (quat == nullptr || args.nk >= 3)), std::memory_order_relaxed);
    return alt.geo == GEO_MID ? dispatch_mid(h, rest, dma, rank1, stream, partials, rest_lean)
                              : dispatch_flat(h, rest, dma, rank1, stream, partials, false);
  };
  // The cost model ranks candidates with ONE device's tile times (sri_column_plan.h).  Where it wants to deviate from the
  // operator's own geometry, the first call at this column count lets THIS device decide: both schedules run on the call's own
  // panel (they write the same bits) — once each untimed (code loads, first touches, the clock leaving its idle state), then
  // own - planned - planned - own between five events, the two of each kind ADDED (a clock that is still ramping drifts through
  // the four launches: in the sums a linear drift cancels; round 5's first cut compared the minima, and a first call on a cold
  // device — own first and LAST — gave the verdict to the own geometry where the plan is 7 % faster: N = 320, 385 in
  // profiles/r05_sweep_n.jsonl of two boxes) — and the handle remembers which one was faster (a tie within 3 % goes to the own geometry).  Plain applies and sri_integrate only (a fused step must run once);
  // never inside a stream capture (the model decides there); SRI_OPT_PLAN_CALIBRATION = 0 trusts the model everywhere.
  if (choice.alt >= 0 && quat == nullptr && h->plan_calibrate) {
    plan::Choice own_only;
    own_only.primary_cols = M;
    sri_apply_handle::PlanVerdict* verdict = nullptr;
    for (int i = 0; i < h->plan_verdict_count; ++i)
      if (h->plan_verdicts[i].cols == M && h->plan_verdicts[i].rank1 == rank1) verdict = &h->plan_verdicts[i];
    // (the first call of a process finds the device in its idle state, and four launches of half a millisecond on a clock that is
    //  still settling have decided wrongly — N = 320 in two of seven sweeps on different boxes, 0.77 instead of 0.83 of the peak for
    //  the rest of the process: the sixteenth call at the column count measures once more, and THAT verdict stays)
    const bool recheck = verdict != nullptr && verdict->calls + 1 == sri_apply_handle::kPlanRecheckCall;
    if (verdict != nullptr && !recheck) {
      if (verdict->calls < sri_apply_handle::kPlanRecheckCall) ++verdict->calls;
      return execute(verdict->use_plan ? choice : own_only);
    }
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capturing) != hipSuccess) {
      (void)hipGetLastError();
      capturing = hipStreamCaptureStatusActive;                        // (cannot tell: do not synchronise)
    }
    if (capturing == hipStreamCaptureStatusNone) {
      for (hipEvent_t& e : h->plan_events)
        if (e == nullptr) SRI_HIP(hipEventCreate(&e));
      const plan::Choice* order[4] = {&own_only, &choice, &choice, &own_only};
      rc = execute(own_only);
      if (rc == SRI_OK) rc = execute(choice);
      if (rc != SRI_OK) return rc;
      SRI_HIP(hipEventRecord(h->plan_events[0], stream));
      for (int i = 0; i < 4; ++i) {
        rc = execute(*order[i]);
        if (rc != SRI_OK) return rc;
        SRI_HIP(hipEventRecord(h->plan_events[i + 1], stream));
      }
      SRI_HIP(hipEventSynchronize(h->plan_events[4]));
      float t[4] = {0, 0, 0, 0};
      for (int i = 0; i < 4; ++i) SRI_HIP(hipEventElapsedTime(&t[i], h->plan_events[i], h->plan_events[i + 1]));
      sri_apply_handle::PlanVerdict& v = recheck ? *verdict : h->plan_verdicts[h->plan_verdict_next];
      if (!recheck) {
        h->plan_verdict_next = (h->plan_verdict_next + 1) % sri_apply_handle::kPlanVerdicts;
        if (h->plan_verdict_count < sri_apply_handle::kPlanVerdicts) ++h->plan_verdict_count;
      }
      v.calls = recheck ? sri_apply_handle::kPlanRecheckCall : 1u;
      v.cols = M;
      v.rank1 = rank1;
      v.own_us = 1e3f * 0.5f * (t[0] + t[3]);
      v.plan_us = 1e3f * 0.5f * (t[1] + t[2]);
      v.use_plan = v.plan_us < 0.97f * v.own_us;
      if (!v.use_plan) h->last_kernel.store(in_parts ? (lean_for(0, M) ? h->parts_name_lean : h->parts_name)
                                                     : kernel_name(h->geo, dma, false, lean_for(0, M)), std::memory_order_relaxed);
      // (the last schedule that ran was the own geometry's: Y is complete either way; last_kernel names what later calls will take)
      if (v.use_plan) {
        // report the planned schedule's name without running it a third time: the labels execute() would store
        const size_t kept = choice.primary_cols;
        const sri_apply_handle::Alternate& alt = h->alternates[choice.alt];
        if (kept != 0) h->last_kernel.store(h->mixed_labels[lean_for(0, kept) ? 1 : 0][choice.alt].c_str(), std::memory_order_relaxed);
        else h->last_kernel.store(kernel_name(alt.geo, true, false, lean_for(0, M) && alt.geo != GEO_FLAT), std::memory_order_relaxed);
      }
      return SRI_OK;
    }
    if (verdict != nullptr) return execute(verdict->use_plan ? choice : own_only);   // (a capture at the re-check call: next time)
  }
  return execute(choice);
}

int integrate_with_position_rhs(sri_apply_t* h, const double* dG, size_t ldg, const double* d_x_base, double* dQ, size_t ldq, double* dB,
                                size_t ldb, size_t M, void* stream, const FixedPointCriterion* crit) {
  if (h == nullptr || h->which != SRI_OP_DNN_INV || d_x_base == nullptr) return SRI_EINVAL;
  const PositionOut pos{dB, ldb, crit};
  return run(h, dG, ldg, dQ, ldq, M, d_x_base, static_cast<hipStream_t>(stream), nullptr, nullptr, nullptr, nullptr, &pos);
}

uint32_t criterion_wave_rows(const sri_apply_t* h) {
  if (h == nullptr) return 0u;
  if (criterion_alternate(h) >= 0 || h->geo == GEO_MID) return static_cast<uint32_t>(GeoMid::WM);
  return h->geo == GEO_TALL ? static_cast<uint32_t>(GeoTall::WM) : 0u;
}

int integrate_quat_step(sri_apply_t* h, const double* dG, size_t ldg, const double* d_x_base, const QuatEpilogue& quat,
                        double* dG_next, size_t ldn, size_t M, double* d_tips, void* stream, uint32_t* partials) {
  if (h == nullptr || (h->which != SRI_OP_DNN_INV && h->which != kWhichInterpolation) || d_x_base == nullptr) return SRI_EINVAL;
  if (quat.partial != nullptr && partials == nullptr) return SRI_EINVAL;
  return run(h, dG, ldg, dG_next, ldn, M, d_x_base, static_cast<hipStream_t>(stream), d_tips, &quat, partials);
}

// what sri_host_panels.hip enqueues for a block that sits in one of its device slots (plain apply, or sri_integrate's form)
int apply_device_panel(sri_apply_t* h, const double* dX, size_t ldx, double* dY, size_t ldy, size_t M, const double* d_x_base, void* stream) {
  return run(h, dX, ldx, dY, ldy, M, d_x_base, static_cast<hipStream_t>(stream));
}

// rows_out x kdim operator with rank-1 weights w (rows_out; result = A X - w x_base^T), for integrate_quat_step only:
// G = 1/2 A(K) (A X - w x_base^T) with A a coarse-to-fine interpolation matrix is the rod pipeline's prolongation (sri_rod.hip)
int create_interpolation_operator(sri_apply_t** out, uint32_t rows_out, uint32_t kdim, const double* A, size_t lda, const double* w, int device) {
  if (w == nullptr || kdim < 16 || kdim % BK == 0) return SRI_EUNSUPPORTED;      // (the rank-1 term rides in the free slot of the last k chunk)
  return create_from_matrix(out, kdim, A, lda, w, kWhichInterpolation, device, rows_out);
}

// resident workgroups per CU of geometry `geo` on `dev`, as its prepare() measured them (0: never prepared — the plan's table then)
static int measured_residency(int geo, int dev) {
  switch (geo) {
    case GEO_FLAT: return resident_flat(dev);
    case GEO_MID: return resident_mid(dev);
    case GEO_TALL: return resident_tall(dev);
    case GEO_WIDE: return resident_wide64(dev);
    case GEO_WIDE32: return resident_wide32(dev);
    case GEO_TALL_STRIP: return resident_strip(dev);
    case GEO_TALL_MINI: return resident_mini(dev);
    case GEO_MID_MINI: return resident_midmini(dev);
    case GEO_FLAT_MINI: return resident_flatmini(dev);
    case GEO_96: return resident_g96(dev);
    case GEO_192: return resident_g192(dev);
    case GEO_160: return resident_g160(dev);
    case GEO_224: return resident_g224(dev);
    default: return 0;
  }
}

uint32_t max_partials(const sri_apply_t* h) { return h ? static_cast<uint32_t>(h->num_cus) * 8u * 4u : 0; }

uint32_t padded_rows(const sri_apply_t* h) { return h ? h->row_blocks * h->bm : 0; }

// w = Dn_NN^-1 Dn_IN of an integration handle exactly as the kernel applies it (rows doubles, host).
int ivp_weights(const sri_apply_t* h, double* w_host) {
  if (h == nullptr || w_host == nullptr || h->which != SRI_OP_DNN_INV || h->d_w == nullptr) return SRI_EINVAL;
  return hipMemcpy(w_host, h->d_w, sizeof(double) * h->rows_out, hipMemcpyDeviceToHost) == hipSuccess ? SRI_OK : SRI_EHIP;
}

}  // namespace sri

// ------------------------------------------------------------------------------------------
// C ABI (include/sri_capi.h)
// ------------------------------------------------------------------------------------------
extern "C" {

int sri_apply_create(sri_apply_t** out, uint32_t n, int which, int device) {
  if (out == nullptr || n < 2) return SRI_EINVAL;
  *out = nullptr;
  if (which != SRI_OP_DN && which != SRI_OP_DNN_INV && which != SRI_OP_DNN) return SRI_EINVAL;
  if (which != SRI_OP_DN && n < 3) return SRI_EINVAL;
  std::vector<double> A, w;
  uint32_t rows = 0;
  try {
    if (which == SRI_OP_DN) {
      rows = n;
      A.resize(static_cast<size_t>(n) * n);
      int rc = sri::cheb_dn(n, A.data(), n);
      if (rc != SRI_OK) return rc;
    } else {
      rows = n - 1;
      A.resize(static_cast<size_t>(rows) * rows);
      std::vector<double> Din(rows);
      if (which == SRI_OP_DNN_INV) {
        int rc = sri::cheb_integration(n, A.data(), rows, Din.data());
        if (rc != SRI_OK) return rc;
        // w = Dnn_inv * Din, ascending k: the rank-1 factor of the initial-value term
        w.assign(rows, 0.0);
        for (uint32_t k = 0; k < rows; ++k)
          for (uint32_t i = 0; i < rows; ++i) w[i] = w[i] + A[i + static_cast<size_t>(k) * rows] * Din[k];
      } else {
        std::vector<double> D(static_cast<size_t>(n) * n);
        int rc = sri::cheb_dn(n, D.data(), n);
        if (rc != SRI_OK) return rc;
        rc = sri::cheb_blocks(n, D.data(), n, A.data(), rows, nullptr);
        if (rc != SRI_OK) return rc;
      }
    }
  } catch (const std::bad_alloc&) {
    return SRI_ENOMEM;
  }
  return sri::create_from_matrix(out, rows, A.data(), rows, w.empty() ? nullptr : w.data(), which, device);
}

int sri_apply_create_matrix(sri_apply_t** out, uint32_t rows, const double* A, size_t lda, int device) {
  return sri::create_from_matrix(out, rows, A, lda, nullptr, -1, device);
}

int sri_apply_set_max_workgroups(sri_apply_t* h, uint32_t max_workgroups) {
  if (h == nullptr) return SRI_EINVAL;
  h->max_workgroups = max_workgroups;
  return SRI_OK;
}

int sri_apply_set_lean_tail(sri_apply_t* h, int enable) {
  if (h == nullptr) return SRI_EINVAL;
  h->lean_tail = enable != 0;
  return SRI_OK;
}

int sri_apply_set_column_plan(sri_apply_t* h, int enable) {
  if (h == nullptr) return SRI_EINVAL;
  h->column_plan = enable != 0;
  return SRI_OK;
}

int sri_apply_set_option(sri_apply_t* h, int option, long value) {
  if (h == nullptr) return SRI_EINVAL;
  switch (option) {
    case SRI_OPT_COLUMN_PLAN: h->column_plan = value != 0; return SRI_OK;
    case SRI_OPT_LEAN_TAIL: h->lean_tail = value != 0; return SRI_OK;
    case SRI_OPT_ROW_PARTS: h->use_parts = value != 0; return SRI_OK;
    case SRI_OPT_SMALL_KERNEL: h->use_small = value != 0; return SRI_OK;
    case SRI_OPT_XCD_TILE_WALK: h->xcd_walk = value != 0; return SRI_OK;
    case SRI_OPT_PLAN_CALIBRATION:
      h->plan_calibrate = value != 0;
      h->plan_verdict_count = h->plan_verdict_next = 0;                 // (what was measured is forgotten either way)
      return SRI_OK;
    case SRI_OPT_HOST_COPY_THREADS: return sri_apply_set_host_copy_threads(h, static_cast<int>(value));
    case SRI_OPT_HOST_BLOCK_COLUMNS:
      if (value < 0 || (value != 0 && value < 128) || value > (1L << 24)) return SRI_EINVAL;
      h->host_block_cols = static_cast<size_t>(value) & ~static_cast<size_t>(127);      // whole 128-column tiles
      return SRI_OK;
    case SRI_OPT_MAX_WORKGROUPS:
      if (value < 0 || value > 0x7fffffffL) return SRI_EINVAL;
      h->max_workgroups = static_cast<uint32_t>(value);
      return SRI_OK;
    default: return SRI_EINVAL;
  }
}

int sri_apply_get_option(const sri_apply_t* h, int option, long* value) {
  if (h == nullptr || value == nullptr) return SRI_EINVAL;
  switch (option) {
    case SRI_OPT_COLUMN_PLAN: *value = h->column_plan ? 1 : 0; return SRI_OK;
    case SRI_OPT_LEAN_TAIL: *value = h->lean_tail ? 1 : 0; return SRI_OK;
    case SRI_OPT_ROW_PARTS: *value = h->use_parts ? 1 : 0; return SRI_OK;
    case SRI_OPT_SMALL_KERNEL: *value = h->use_small ? 1 : 0; return SRI_OK;
    case SRI_OPT_XCD_TILE_WALK: *value = h->xcd_walk ? 1 : 0; return SRI_OK;
    case SRI_OPT_PLAN_CALIBRATION: *value = h->plan_calibrate ? 1 : 0; return SRI_OK;
    case SRI_OPT_HOST_COPY_THREADS: *value = h->host_copy_threads; return SRI_OK;
    case SRI_OPT_HOST_BLOCK_COLUMNS: *value = static_cast<long>(h->host_block_cols); return SRI_OK;
    case SRI_OPT_MAX_WORKGROUPS: *value = static_cast<long>(h->max_workgroups); return SRI_OK;
    default: return SRI_EINVAL;
  }
}

int sri_apply_dim(const sri_apply_t* h, uint32_t* rows) {
  if (h == nullptr || rows == nullptr) return SRI_EINVAL;
  *rows = h->rows;
  return SRI_OK;
}

int sri_apply(sri_apply_t* h, const double* dX, size_t ldx, double* dY, size_t ldy, size_t M, void* hip_stream) {
  return sri::run(h, dX, ldx, dY, ldy, M, nullptr, static_cast<hipStream_t>(hip_stream));
}

int sri_apply_many(sri_apply_t* h, const double* const* dXs, size_t ldx, double* const* dYs, size_t ldy, size_t M, size_t count,
                   void* hip_stream) {
  if (h == nullptr) return SRI_EINVAL;
  const sri::ManyPanels many{dXs, dYs, count};
  return sri::run(h, nullptr, ldx, nullptr, ldy, M, nullptr, static_cast<hipStream_t>(hip_stream), nullptr, nullptr, nullptr, &many);
}

int sri_apply_tips(sri_apply_t* h, const double* dX, size_t ldx, double* dY, size_t ldy, size_t M, double* d_tips,
                   void* hip_stream) {
  if (d_tips == nullptr) return SRI_EINVAL;
  return sri::run(h, dX, ldx, dY, ldy, M, nullptr, static_cast<hipStream_t>(hip_stream), d_tips);
}

int sri_integrate_tips(sri_apply_t* h, const double* dB, size_t ldb, const double* d_x_base, double* dY, size_t ldy, size_t M,
                       double* d_tips, void* hip_stream) {
  if (h == nullptr || h->which != SRI_OP_DNN_INV || d_x_base == nullptr || d_tips == nullptr) return SRI_EINVAL;
  return sri::run(h, dB, ldb, dY, ldy, M, d_x_base, static_cast<hipStream_t>(hip_stream), d_tips);
}

int sri_integrate(sri_apply_t* h, const double* dB, size_t ldb, const double* d_x_base, double* dY, size_t ldy,
                  size_t M, void* hip_stream) {
  if (h == nullptr || h->which != SRI_OP_DNN_INV || d_x_base == nullptr) return SRI_EINVAL;
  return sri::run(h, dB, ldb, dY, ldy, M, d_x_base, static_cast<hipStream_t>(hip_stream));
}

int sri_apply_destroy(sri_apply_t* h) {
  if (h == nullptr) return SRI_OK;
  sri::DeviceGuard guard;
  (void)guard.enter(h->device);
  if (h->d_apack) (void)hipFree(h->d_apack);
  if (h->d_apack_small) (void)hipFree(h->d_apack_small);
  if (h->d_w) (void)hipFree(h->d_w);
  for (sri_apply_handle::Part& part : h->parts)
    if (part.d_apack) (void)hipFree(part.d_apack);
  for (sri_apply_handle::Alternate& alt : h->alternates)
    if (alt.d_apack) (void)hipFree(alt.d_apack);
  sri::host_pipe_destroy(h->pipe);   // sri_host_panels.hip
  for (hipEvent_t e : h->plan_events)
    if (e != nullptr) (void)hipEventDestroy(e);
  delete h;
  return SRI_OK;
}

const char* sri_apply_last_kernel(const sri_apply_t* h) { return h ? h->last_kernel.load(std::memory_order_relaxed) : "none"; }

int sri_apply_summation_order(const sri_apply_t* h, int* ascending) {
  if (h == nullptr || ascending == nullptr) return SRI_EINVAL;
  *ascending = h->ascending ? 1 : 0;
  return SRI_OK;
}

int sri_last_hip_error(void) { return sri::g_last_hip_error; }

int sri_device_count(int* count) {
  if (count == nullptr) return SRI_EINVAL;
  *count = 0;
  hipError_t e = hipGetDeviceCount(count);
  if (e != hipSuccess) {
    sri::g_last_hip_error = static_cast<int>(e);
    (void)hipGetLastError();
    *count = 0;
    return SRI_ENODEVICE;
  }
  return SRI_OK;
}

}  // extern "C"
