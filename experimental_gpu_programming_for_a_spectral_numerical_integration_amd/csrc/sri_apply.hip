// sri_apply.hip — the hot path: Y[R x M] = A[R x R] * X[R x M] in fp64 on gfx950 (MI355X).
//
// Replaces the dense products of the reference:
//   main.cpp:172   r_stack = Dn_NN_inv * (b_NN - ivp)          (A = Dn_NN^-1, integration)
//   main.cpp:97    (I_s (x) Dn) * chi  ==  vec(Dn * X)          (A = D_N, differentiation)
// batched over rods: X holds one column per (rod, state component), column-major, so a column
// is R contiguous doubles (main.cpp:113,130-133).
//
// Design (DESIGN.md has the numbers):
//   * v_mfma_f64_16x16x4_f64 does all the arithmetic.  Measured on MI355X (profiles/
//     r01_mfma_f64_probe.log): one issue per 64 cycles per SIMD, same dependent latency,
//     77.5 TF chip-wide; f64 VALU work shares that pipe, so the inner loop holds no f64 VALU.
//   * A (up to 8 MiB at N=1024) does not fit a CU.  It is packed on the host ONCE, at handle
//     creation, into the exact byte order the MFMA A-fragments are read in
//     ([row block][k chunk][16-row tile][k pair][lane][2]), so that a K-slab is one linear
//     LDS-DMA copy (global_load_lds_dwordx4, no VGPRs) and every fragment read is a
//     conflict-free 1-KiB ds_read_b128.  A stays resident in each XCD's 4 MiB L2; HBM sees it once.
//   * The X panel is streamed from HBM exactly once, also by LDS-DMA: every wave instruction
//     moves 8 columns x 128 contiguous bytes.  The LDS image keeps one 128-byte row per column
//     with its eight 16-byte pieces XOR-swizzled by (column>>1)&7 — applied on the per-lane
//     SOURCE address, because the LDS side of an LDS-DMA is lane-linear — which makes the
//     B-fragment ds_read_b128 conflict-free (tools/lds_bank_model.py).
//   * k is consumed in the order (k pair j, lane k-slot g, element e) -> 8j + 2g + e and the 16
//     rows of every output tile are permuted (row_of_slot) so that each lane ends up with
//     adjacent rows: Y leaves in 16-byte stores, four lanes covering one 64-byte segment.
//     Both permutations are folded into the host-side packing of A; X and Y keep the caller's
//     plain column-major layout.
//   * 256-thread workgroups (one wave per SIMD), two workgroups per CU (<= 256 VGPRs,
//     80 KiB LDS each) so that one workgroup's barrier / epilogue is covered by the other's
//     MFMAs.  Two-stage LDS ring, one barrier per 16-deep k chunk.
//
// There is no CPU fallback: without a gfx950 device every entry point fails with an error.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "sri_internal.h"

namespace sri {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

static thread_local int g_last_hip_error = 0;

#define SRI_HIP(call)                                            \
  do {                                                           \
    hipError_t e__ = (call);                                     \
    if (e__ != hipSuccess) {                                     \
      g_last_hip_error = static_cast<int>(e__);                  \
      (void)hipGetLastError();                                   \
      return SRI_EHIP;                                           \
    }                                                            \
  } while (0)

// ---------------------------------------------------------------------------------------------
// Tile geometry
// ---------------------------------------------------------------------------------------------
constexpr int BK = kChunkK;          // 16 k per stage
constexpr int KPAIRS = BK / 8;       // 16-byte fragment reads per stage and tile
constexpr int NSTAGE = 2;
constexpr int X_ROW_BYTES = BK * 8;  // one column of the X stage: 128 B = 8 pieces of 16 B

template <int WM_, int WN_, int TM_, int TN_>
struct Geo {
  static constexpr int WM = WM_, WN = WN_, TM = TM_, TN = TN_;
  static constexpr int NW = WM * WN;
  static constexpr int THREADS = 64 * NW;
  static constexpr int BM = WM * TM * 16;
  static constexpr int BN = WN * TN * 16;
  static constexpr int A_BYTES = BM * BK * 8;
  static constexpr int X_BYTES = BN * BK * 8;
  static constexpr int STAGE_BYTES = A_BYTES + X_BYTES;
  static constexpr int LDS_BYTES = NSTAGE * STAGE_BYTES;
  static constexpr int A_DMA_PER_WAVE = A_BYTES / 1024 / NW;
  static constexpr int X_DMA_PER_WAVE = X_BYTES / 1024 / NW;
  static_assert(A_BYTES % (1024 * NW) == 0 && X_BYTES % (1024 * NW) == 0, "stage must split into whole 1-KiB DMA pieces per wave");
  static_assert(THREADS == 256, "two 256-thread workgroups per CU is what the LDS/VGPR budget is sized for");
};

struct ApplyArgs {
  const double* apack;   // [row_blocks][nk][BM*BK] packed fragments
  const double* x;
  double* y;
  const double* w;       // optional rank-1 epilogue: y -= w[row] * xb[col]   (w padded to row_blocks*BM)
  const double* xb;
  size_t ldx, ldy, m;    // m = number of columns
  uint32_t rows;         // valid rows of A (= valid k)
  uint32_t nk;           // k chunks
  uint32_t row_blocks;
};

__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// ---------------------------------------------------------------------------------------------
// The kernel.
//   DMA   : X by LDS-DMA (needs 16-byte aligned columns and rows % 16 == 0), Y by 16-byte stores.
//           Otherwise X goes through registers with bounds checks and zero fill, Y by guarded
//           8-byte stores: any rows / ld / M / alignment.
//   RANK1 : epilogue subtracts w[row]*xb[col] (the initial-value term of main.cpp:162-164,172).
// ---------------------------------------------------------------------------------------------
template <class G, bool DMA, bool RANK1>
__global__ __launch_bounds__(G::THREADS, 2) void apply_kernel(const ApplyArgs p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int WN = G::WN, TM = G::TM, TN = G::TN, NW = G::NW;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wid / WN, wn = wid % WN;
  const int c = lane & 15, g = lane >> 4;

  const uint32_t rb = blockIdx.x % p.row_blocks;
  const size_t ct = blockIdx.x / p.row_blocks;
  const size_t col0 = ct * G::BN;
  const size_t last_col = p.m - 1;

  // ---- per-lane constants of the X staging --------------------------------------------------
  // DMA: instruction i (= wid + it*NW) covers columns 8i..8i+7 of the tile; this lane writes
  // piece position (lane & 7) of column 8i + (lane >> 3), which must hold source piece
  // (lane & 7) ^ sigma(column).
  const char* xsrc[DMA ? G::X_DMA_PER_WAVE : 1];
  if constexpr (DMA) {
#pragma unroll
    for (int it = 0; it < G::X_DMA_PER_WAVE; ++it) {
      const int cc = (wid + it * NW) * 8 + (lane >> 3);
      const int piece = (lane & 7) ^ ((cc >> 1) & 7);
      size_t col = col0 + cc;
      col = col < last_col ? col : last_col;          // clamp: tail columns re-read a valid one
      xsrc[it] = reinterpret_cast<const char*>(p.x + col * p.ldx) + piece * 16;
    }
  }
  const char* asrc = reinterpret_cast<const char*>(p.apack) +
                     static_cast<size_t>(rb) * p.nk * G::A_BYTES + lane * 16;

  // Register staging for the generic path: the X stage is BN*BK doubles, THREADS*XR of them.
  constexpr int XR = G::BN * BK / G::THREADS;
  double xreg[DMA ? 1 : XR];

  auto stage_issue = [&](uint32_t kc, int buf) {
    char* sA = smem + buf * G::STAGE_BYTES;
    const char* gA = asrc + static_cast<size_t>(kc) * G::A_BYTES;
#pragma unroll
    for (int it = 0; it < G::A_DMA_PER_WAVE; ++it) {
      const int i = wid + it * NW;
      lds_dma16(gA + i * 1024, sA + i * 1024);
    }
    if constexpr (DMA) {
      char* sX = sA + G::A_BYTES;
#pragma unroll
      for (int it = 0; it < G::X_DMA_PER_WAVE; ++it) {
        const int i = wid + it * NW;
        lds_dma16(xsrc[it] + static_cast<size_t>(kc) * X_ROW_BYTES, sX + i * 1024);
      }
    } else {
#pragma unroll
      for (int r = 0; r < XR; ++r) {
        const int idx = tid + r * G::THREADS;      // element of the stage, k fastest
        const int cc = idx / BK, k = idx % BK;
        const size_t col = col0 + cc;
        const uint32_t kk = kc * BK + k;
        xreg[r] = (col < p.m && kk < p.rows) ? p.x[col * p.ldx + kk] : 0.0;
      }
    }
  };
  auto stage_commit = [&](int buf) {   // generic path only: registers -> swizzled LDS image
    if constexpr (!DMA) {
      char* sX = smem + buf * G::STAGE_BYTES + G::A_BYTES;
#pragma unroll
      for (int r = 0; r < XR; ++r) {
        const int idx = tid + r * G::THREADS;
        const int cc = idx / BK, k = idx % BK;
        const int pos = (k >> 1) ^ ((cc >> 1) & 7);
        *reinterpret_cast<double*>(sX + cc * X_ROW_BYTES + pos * 16 + (k & 1) * 8) = xreg[r];
      }
    }
  };

  // ---- accumulators --------------------------------------------------------------------------
  d4 acc[TM][TN];
#pragma unroll
  for (int tm = 0; tm < TM; ++tm)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) acc[tm][tn] = d4{0.0, 0.0, 0.0, 0.0};

  // fragment read addresses (bytes, relative to the stage base)
  const int a_off = wm * (TM * KPAIRS * 1024) + lane * 16;
  const int sig = (c >> 1) & 7;
  const int x_off = G::A_BYTES + (wn * TN * 16 + c) * X_ROW_BYTES;

  auto compute = [&](int buf) {
    const char* base = smem + buf * G::STAGE_BYTES;
#pragma unroll
    for (int j = 0; j < KPAIRS; ++j) {
      d2 a[TM], b[TN];
#pragma unroll
      for (int tm = 0; tm < TM; ++tm)
        a[tm] = *reinterpret_cast<const d2*>(base + a_off + (tm * KPAIRS + j) * 1024);
      const int xpiece = ((4 * j + g) ^ sig) * 16;
#pragma unroll
      for (int tn = 0; tn < TN; ++tn)
        b[tn] = *reinterpret_cast<const d2*>(base + x_off + tn * 16 * X_ROW_BYTES + xpiece);
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
          for (int tn = 0; tn < TN; ++tn)
            acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[tm][e], b[tn][e], acc[tm][tn], 0, 0, 0);
    }
  };

  // ---- main loop: 2-stage ring, one barrier per chunk ------------------------------------------
  stage_issue(0, 0);
  stage_commit(0);
  for (uint32_t kc = 0; kc < p.nk; ++kc) {
    const int buf = kc & 1;
    __syncthreads();                       // chunk kc landed (drains the LDS-DMA), chunk kc-1 fully consumed
    const bool more = kc + 1 < p.nk;
    if (more) stage_issue(kc + 1, buf ^ 1);
    compute(buf);
    if (more) stage_commit(buf ^ 1);
  }

  // ---- epilogue -------------------------------------------------------------------------------
  const uint32_t row_base = rb * G::BM + wm * (TM * 16) + 2 * g;
#pragma unroll
  for (int tn = 0; tn < TN; ++tn) {
    const size_t col = col0 + (wn * TN + tn) * 16 + c;
    if (col >= p.m) continue;
    double* ycol = p.y + col * p.ldy;
    double xbv = 0.0;
    if constexpr (RANK1) xbv = p.xb[col];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const uint32_t row = row_base + tm * 16 + q * 8;
        double v0 = acc[tm][tn][2 * q], v1 = acc[tm][tn][2 * q + 1];
        if constexpr (RANK1) {
          if (row < p.rows) v0 -= p.w[row] * xbv;
          if (row + 1 < p.rows) v1 -= p.w[row + 1] * xbv;
        }
        if constexpr (DMA) {
          if (row < p.rows) *reinterpret_cast<d2*>(ycol + row) = d2{v0, v1};
        } else {
          if (row < p.rows) ycol[row] = v0;
          if (row + 1 < p.rows) ycol[row + 1] = v1;
        }
      }
  }
}

// ---------------------------------------------------------------------------------------------
// Handle
// ---------------------------------------------------------------------------------------------
using GeoTall = Geo<4, 1, 4, 4>;   // BM 256 x BN 64 : rows > 128   (N = 256, 1024, ...)
using GeoMid = Geo<2, 2, 4, 2>;    // BM 128 x BN 64 : 64 < rows <= 128
using GeoFlat = Geo<1, 4, 4, 1>;   // BM  64 x BN 64 : rows <= 64   (N = 16, 64)

enum GeoId { GEO_FLAT = 0, GEO_MID = 1, GEO_TALL = 2 };

}  // namespace sri

struct sri_apply_handle {
  int device = 0;
  uint32_t rows = 0;
  int which = -1;
  int geo = 0;
  uint32_t bm = 0, nk = 0, row_blocks = 0;
  double* d_apack = nullptr;
  double* d_w = nullptr;        // Dnn_inv * Din (padded), for sri_integrate
  const char* last_kernel = "none";
};

namespace sri {

template <class G>
static void pack_operator(const double* A, size_t lda, uint32_t rows, uint32_t nk, uint32_t row_blocks,
                          std::vector<double>& out) {
  out.assign(static_cast<size_t>(row_blocks) * nk * G::BM * BK, 0.0);
  size_t o = 0;
  for (uint32_t rb = 0; rb < row_blocks; ++rb)
    for (uint32_t kc = 0; kc < nk; ++kc)
      for (int tile = 0; tile < G::BM / 16; ++tile)
        for (int j = 0; j < KPAIRS; ++j)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 2; ++e, ++o) {
              const uint32_t row = rb * G::BM + tile * 16 + row_of_slot(l & 15);
              const uint32_t k = kc * BK + k_of(j, l >> 4, e);
              if (row < rows && k < rows) out[o] = A[row + static_cast<size_t>(k) * lda];
            }
}

template <class G, bool DMA, bool RANK1>
static int launch(const ApplyArgs& args, size_t col_tiles, hipStream_t stream) {
  static bool attr_set[16] = {};   // per device
  int dev = 0;
  SRI_HIP(hipGetDevice(&dev));
  if (dev >= 0 && dev < 16 && !attr_set[dev]) {
    SRI_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&apply_kernel<G, DMA, RANK1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
    attr_set[dev] = true;
  }
  const size_t blocks = col_tiles * args.row_blocks;
  if (blocks == 0) return SRI_OK;
  if (blocks > 0x7fffffffull) return SRI_EUNSUPPORTED;
  hipLaunchKernelGGL((apply_kernel<G, DMA, RANK1>), dim3(static_cast<unsigned>(blocks)), dim3(G::THREADS),
                     G::LDS_BYTES, stream, args);
  SRI_HIP(hipGetLastError());
  return SRI_OK;
}

template <class G>
static int dispatch(sri_apply_t* h, const ApplyArgs& args, bool dma, bool rank1, hipStream_t stream) {
  const size_t col_tiles = (args.m + G::BN - 1) / G::BN;
  if (dma && rank1) return launch<G, true, true>(args, col_tiles, stream);
  if (dma) return launch<G, true, false>(args, col_tiles, stream);
  if (rank1) return launch<G, false, true>(args, col_tiles, stream);
  return launch<G, false, false>(args, col_tiles, stream);
}

static const char* kernel_name(int geo, bool dma) {
  static const char* names[3][2] = {
      {"sri_apply_mfma_f64<64x64,reg>", "sri_apply_mfma_f64<64x64,dma>"},
      {"sri_apply_mfma_f64<128x64,reg>", "sri_apply_mfma_f64<128x64,dma>"},
      {"sri_apply_mfma_f64<256x64,reg>", "sri_apply_mfma_f64<256x64,dma>"}};
  return names[geo][dma ? 1 : 0];
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

static int check_device(int device) {
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
  return SRI_OK;
}

static int create_from_matrix(sri_apply_t** out, uint32_t rows, const double* A, size_t lda, const double* w_host,
                              int which, int device) {
  if (out == nullptr || A == nullptr || rows < 1 || lda < rows) return SRI_EINVAL;
  *out = nullptr;
  int rc = check_device(device);
  if (rc != SRI_OK) return rc;
  DeviceGuard guard;
  rc = guard.enter(device);
  if (rc != SRI_OK) return rc;

  sri_apply_t* h = new (std::nothrow) sri_apply_handle();
  if (h == nullptr) return SRI_ENOMEM;
  h->device = device;
  h->rows = rows;
  h->which = which;
  h->geo = rows <= 64 ? GEO_FLAT : (rows <= 128 ? GEO_MID : GEO_TALL);
  h->bm = h->geo == GEO_FLAT ? GeoFlat::BM : (h->geo == GEO_MID ? GeoMid::BM : GeoTall::BM);
  h->nk = (rows + BK - 1) / BK;
  h->row_blocks = (rows + h->bm - 1) / h->bm;

  std::vector<double> packed;
  try {
    if (h->geo == GEO_FLAT) pack_operator<GeoFlat>(A, lda, rows, h->nk, h->row_blocks, packed);
    else if (h->geo == GEO_MID) pack_operator<GeoMid>(A, lda, rows, h->nk, h->row_blocks, packed);
    else pack_operator<GeoTall>(A, lda, rows, h->nk, h->row_blocks, packed);
  } catch (const std::bad_alloc&) {
    delete h;
    return SRI_ENOMEM;
  }
  auto fail = [&](int code) {
    if (h->d_apack) (void)hipFree(h->d_apack);
    if (h->d_w) (void)hipFree(h->d_w);
    delete h;
    return code;
  };
  if (hipMalloc(&h->d_apack, packed.size() * sizeof(double)) != hipSuccess) return fail(SRI_ENOMEM);
  if (hipMemcpy(h->d_apack, packed.data(), packed.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
    return fail(SRI_EHIP);
  if (w_host != nullptr) {
    const size_t padded = static_cast<size_t>(h->row_blocks) * h->bm;
    std::vector<double> w(padded, 0.0);
    std::memcpy(w.data(), w_host, sizeof(double) * rows);
    if (hipMalloc(&h->d_w, padded * sizeof(double)) != hipSuccess) return fail(SRI_ENOMEM);
    if (hipMemcpy(h->d_w, w.data(), padded * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return fail(SRI_EHIP);
  }
  *out = h;
  return SRI_OK;
}

static int run(sri_apply_t* h, const double* dX, size_t ldx, double* dY, size_t ldy, size_t M, const double* d_xb,
               hipStream_t stream) {
  if (h == nullptr || (M > 0 && (dX == nullptr || dY == nullptr)) || ldx < h->rows || ldy < h->rows) return SRI_EINVAL;
  if (M == 0) return SRI_OK;
  const bool rank1 = d_xb != nullptr;
  if (rank1 && h->d_w == nullptr) return SRI_EINVAL;
  DeviceGuard guard;
  int rc = guard.enter(h->device);
  if (rc != SRI_OK) return rc;

  ApplyArgs args;
  args.apack = h->d_apack;
  args.x = dX;
  args.y = dY;
  args.w = h->d_w;
  args.xb = d_xb;
  args.ldx = ldx;
  args.ldy = ldy;
  args.m = M;
  args.rows = h->rows;
  args.nk = h->nk;
  args.row_blocks = h->row_blocks;

  const bool dma = (h->rows % 16 == 0) && (ldx % 2 == 0) && (ldy % 2 == 0) &&
                   (reinterpret_cast<uintptr_t>(dX) % 16 == 0) && (reinterpret_cast<uintptr_t>(dY) % 16 == 0);
  h->last_kernel = kernel_name(h->geo, dma);
  if (h->geo == GEO_FLAT) return dispatch<GeoFlat>(h, args, dma, rank1, stream);
  if (h->geo == GEO_MID) return dispatch<GeoMid>(h, args, dma, rank1, stream);
  return dispatch<GeoTall>(h, args, dma, rank1, stream);
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

int sri_apply_dim(const sri_apply_t* h, uint32_t* rows) {
  if (h == nullptr || rows == nullptr) return SRI_EINVAL;
  *rows = h->rows;
  return SRI_OK;
}

int sri_apply(sri_apply_t* h, const double* dX, size_t ldx, double* dY, size_t ldy, size_t M, void* hip_stream) {
  return sri::run(h, dX, ldx, dY, ldy, M, nullptr, static_cast<hipStream_t>(hip_stream));
}

int sri_integrate(sri_apply_t* h, const double* dB, size_t ldb, const double* d_x_base, double* dY, size_t ldy,
                  size_t M, void* hip_stream) {
  if (h == nullptr || h->which != SRI_OP_DNN_INV || d_x_base == nullptr) return SRI_EINVAL;
  return sri::run(h, dB, ldb, dY, ldy, M, d_x_base, static_cast<hipStream_t>(hip_stream));
}

int sri_apply_host(sri_apply_t* h, const double* X, size_t ldx, double* Y, size_t ldy, size_t M) {
  if (h == nullptr || (M > 0 && (X == nullptr || Y == nullptr)) || ldx < h->rows || ldy < h->rows) return SRI_EINVAL;
  if (M == 0) return SRI_OK;
  sri::DeviceGuard guard;
  int rc = guard.enter(h->device);
  if (rc != SRI_OK) return rc;
  using sri::g_last_hip_error;
  const size_t rows = h->rows;
  const size_t ldd = (rows + 1) & ~static_cast<size_t>(1);   // even leading dimension: DMA path when rows % 16 == 0
  double *dX = nullptr, *dY = nullptr;
  if (hipMalloc(&dX, ldd * M * sizeof(double)) != hipSuccess) return SRI_ENOMEM;
  if (hipMalloc(&dY, ldd * M * sizeof(double)) != hipSuccess) { (void)hipFree(dX); return SRI_ENOMEM; }
  auto done = [&](int code) { (void)hipFree(dX); (void)hipFree(dY); return code; };
  if (hipMemcpy2D(dX, ldd * sizeof(double), X, ldx * sizeof(double), rows * sizeof(double), M, hipMemcpyHostToDevice) != hipSuccess)
    return done(SRI_EHIP);
  rc = sri::run(h, dX, ldd, dY, ldd, M, nullptr, nullptr);
  if (rc != SRI_OK) return done(rc);
  if (hipMemcpy2D(Y, ldy * sizeof(double), dY, ldd * sizeof(double), rows * sizeof(double), M, hipMemcpyDeviceToHost) != hipSuccess)
    return done(SRI_EHIP);
  return done(SRI_OK);
}

int sri_apply_destroy(sri_apply_t* h) {
  if (h == nullptr) return SRI_OK;
  sri::DeviceGuard guard;
  (void)guard.enter(h->device);
  if (h->d_apack) (void)hipFree(h->d_apack);
  if (h->d_w) (void)hipFree(h->d_w);
  delete h;
  return SRI_OK;
}

const char* sri_apply_last_kernel(const sri_apply_t* h) { return h ? h->last_kernel : "none"; }

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
