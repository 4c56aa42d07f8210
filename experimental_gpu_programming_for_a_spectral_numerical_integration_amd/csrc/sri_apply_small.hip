// sri_apply_small.hip — Y[64 x M] = A[64 x 64] * X[64 x M] in fp64 on gfx950 for SMALL batches: the latency kernel behind
// BASELINE config 2 (D_N at N = 64, 4 096 rods; main.cpp:97 batched).
//
// At 4 096 rods the whole job is 256 tiles of 64 columns — one per CU — and 16.8 MB: there is nothing to stream, every
// microsecond is latency.  Measured on the general kernel (tools/small_diag.hip, profiles/r02_small_diag.log): of 6.4 us per
// back-to-back launch, 2.9 us is the launch itself (an empty kernel of the same shape), ~0.6 us the first loads, 1.7 us the
// MFMAs (the chip-wide floor for 0.134 GFLOP) and 2.1 us the stores, which all sat behind the last MFMA and, written
// through L2, were flushed again at the end of the kernel.  This kernel is the general one (sri_apply.hip) re-cut for
// exactly one tile per workgroup:
//   * everything is requested at once: the operator (host-packed in MFMA fragment order, 32 KiB) and the tile's panel
//     (64 columns x 512 B = 32 KiB), both by LDS-DMA (global_load_lds_dwordx4; the panel as 8 columns x 128 contiguous
//     bytes per wave instruction, swizzled on the source side exactly like the general kernel), interleaved k chunk by k
//     chunk, and awaited chunk by chunk with a counted s_waitcnt vmcnt + raw s_barrier: the first MFMA waits for 8 KiB of
//     operator + 8 KiB of panel, not for 64 KiB;
//   * the operator goes LDS -> registers as it lands (each of the 4 waves owns 32 rows x 64 k = 64 VGPRs of fragments),
//     which the second pass reuses.  (Plain global loads into registers would be shorter still, but hipcc answers a
//     register load beside an LDS-DMA in flight with s_waitcnt vmcnt(0).)
//   * each wave takes its 32 columns in TWO passes of 16: the first while the chunks are still arriving, and its stores
//     then drain under the second pass's MFMAs;
//   * Y leaves in NON-TEMPORAL 16-byte stores (when its columns start on 128-byte lines): written once, never read
//     here, and lines that linger in L2 are flushed at the end of the kernel (6.1 -> 5.4 us per launch);
//   * arithmetic: v_mfma_f64_16x16x4_f64 only, k consumed in the paired order of the general kernel for this size
//     (sri_internal.h: k_of_paired), tile accumulated transposed so that Y leaves in 16-byte stores — the same bits as
//     the general kernel produces, whichever of the two a batch size selects.
// Result: 5.4 us per back-to-back launch instead of 6.3 (2.9 of them the launch floor).  Batches with more tiles than
// resident workgroups (2 per CU) stay on the general kernel, whose k-chunk ring streams at the device's copy rate.
//
// There is no CPU fallback: without a gfx950 device every entry point fails with an error.
#include <hip/hip_runtime.h>

#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#define SRI_DEVICE_HELPERS 1
#include "sri_internal.h"

namespace sri {
namespace small {

typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int ROWS = kSmallRows;          // 64
constexpr int BN = 64;                    // columns per tile
constexpr int CHUNKS = ROWS / 16;         // 16-k chunks per tile
constexpr int CHUNK_BYTES = BN * 16 * 8;  // 8 KiB: 64 columns x 128 B
constexpr int X_BYTES = CHUNKS * CHUNK_BYTES;          // the tile's panel: 32 KiB
constexpr int A_BYTES = ROWS * ROWS * 8;              // operator staging: 32 KiB
constexpr int LDS_BYTES = X_BYTES + A_BYTES;          // 64 KiB: two workgroups per CU
constexpr int THREADS = 256;
constexpr int KPAIRS = ROWS / 8;                      // 8 fragment pairs along k

struct Args {
  const double* apack;   // [k chunk 4][wave row half 2][row tile 2][k pair 2][lane 64][2]: 32 pieces of 1 KiB
  const double* x;
  double* y;
  double* tips;          // optional: tips[col] = Y[0][col]
  size_t ldx, ldy, m;
  uint32_t tiles;        // == gridDim.x
  uint32_t nt_stores;    // 1: every column of Y starts on a 128-byte line -> non-temporal stores
};

__device__ __forceinline__ void lds_dma16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void barrier() {
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <class T>
__device__ __forceinline__ void opaque(T& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(v));
#endif
}

// ABL (diagnostic builds only, tools/small_diag.hip; results are WRONG for 1..8): 1 = no MFMAs (fragment reads kept),
// 2 = no Y stores, 8 = no fragment reads after the first k-step of a chunk, 16 = plain instead of non-temporal Y stores,
// 32 = the columns in one pass instead of two.
template <int ABL = 0>
__global__ __launch_bounds__(THREADS, 2) void apply_small_kernel(const Args p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wid >> 1, wc = wid & 1;          // wave = rows 32 wr .. +31  x  columns 32 wc .. +31 of the tile
  const int c = lane & 15, g = lane >> 4;
  const uint32_t tile = blockIdx.x;               // one tile per workgroup: gridDim.x == tiles
  const size_t last_col = p.m - 1;
  char* const a_stage = smem + X_BYTES;

  // ---- k chunk kc of the tile's panel -> LDS.  A chunk is 8 one-KiB pieces, piece q = columns 8q .. 8q+7 x 128 B; wave w
  // moves pieces w and w + 4.  Lane: column 8q + (lane >> 3), image piece lane & 7 <- source piece (lane & 7) ^
  // sigma(column), sigma = (column >> 1) & 7 (conflict-free 16-byte fragment reads, as in sri_apply.hip).
  auto issue_x = [&](int kc) __attribute__((always_inline)) {
    const size_t col0 = static_cast<size_t>(tile) * BN;
    char* dst = smem + kc * CHUNK_BYTES;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int q = wid + 4 * h;
      const int cc = q * 8 + (lane >> 3);
      const int piece = (lane & 7) ^ ((cc >> 1) & 7);
      size_t col = col0 + cc;
      col = col < last_col ? col : last_col;                         // ragged last tile: re-read a valid column
      lds_dma16(reinterpret_cast<const char*>(p.x + col * p.ldx) + kc * 128 + piece * 16, dst + q * 1024);
    }
  };
  // k chunk kc of the operator: 8 one-KiB pieces, already in fragment order
  auto issue_a = [&](int kc) __attribute__((always_inline)) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = kc * 8 + wid + 4 * h;
      lds_dma16(reinterpret_cast<const char*>(p.apack) + i * 1024 + lane * 16, a_stage + i * 1024);
    }
  };

  // ---- fragment reads, the paired k order of the general kernel for operators of this size (sri_internal.h:
  // k_of_paired): lane (c, g) reads the 16 bytes X[column 16 tn + c of the wave][k = 8j + 2g, 8j + 2g + 1] of k pair j and
  // feeds element e to MFMA (j, e).  The results are the same bits whichever kernel a batch size selects.
  const int sig = (c >> 1) & 7;
  const int xrd = (wc * 32 + c) * 128;
  d2 a[2][KPAIRS];                                  // the operator: 16 fragment pairs per lane
  d4 acc[2][2];
  auto read_a = [&](int kc) __attribute__((always_inline)) {
    const char* base = a_stage + kc * CHUNK_BYTES + wr * 4096 + lane * 16;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
      for (int j = 0; j < 2; ++j) a[tm][kc * 2 + j] = *reinterpret_cast<const d2*>(base + (tm * 2 + j) * 1024);
  };
  // the two k pairs of chunk KC for the column tiles in MASK; the fragments of pair 1 are read before the MFMAs of pair 0
  auto read_x = [&](const char* chunk, int j, d2 (&b)[2], int mask) __attribute__((always_inline)) {
    const int xoff = ((4 * j + g) ^ sig) * 16;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
      if (mask & (1 << tn)) b[tn] = *reinterpret_cast<const d2*>(chunk + xrd + tn * 2048 + xoff);
  };
  auto compute_chunk = [&](auto kc_tag, auto mask_tag) __attribute__((always_inline)) {
    constexpr int KC = decltype(kc_tag)::value, MASK = decltype(mask_tag)::value;
    const char* chunk = smem + KC * CHUNK_BYTES;
    d2 b[2][2];
    read_x(chunk, 0, b[0], MASK);
    if constexpr ((ABL & 8) == 0) read_x(chunk, 1, b[1], MASK);
    else { b[1][0] = b[0][0]; b[1][1] = b[0][1]; }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if constexpr ((ABL & 1) == 0) {
#pragma unroll
          for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
              if (MASK & (1 << tn))
                acc[tm][tn] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[j][tn][e], a[tm][KC * 2 + j][e], acc[tm][tn], 0, 0, 0);
        } else {
          opaque(b[j][0]);
          opaque(b[j][1]);
        }
      }
  };
  using std::integral_constant;

  // ---- Y: lane holds rows 32 wr + 2c, +1 (the two interleaved row tiles) of columns 16 tn + g + 4r -----------------
  auto store_cols = [&](int mask) __attribute__((always_inline)) {
    const size_t col0 = static_cast<size_t>(tile) * BN + wc * 32 + g;
    const uint32_t row = 32 * wr + 2 * c;
    const bool full = static_cast<size_t>(tile) * BN + BN <= p.m;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (!(mask & (1 << tn))) continue;
        const size_t col = col0 + tn * 16 + r * 4;
        if constexpr ((ABL & 2) != 0) {
          opaque(acc[0][tn]);
          opaque(acc[1][tn]);
        } else if (full || col < p.m) {
          d2* yp = reinterpret_cast<d2*>(p.y + col * p.ldy + row);    // 16 B at 8-byte alignment is fine
          if ((ABL & 16) != 0 || p.nt_stores == 0) *yp = d2{acc[0][tn][r], acc[1][tn][r]};
          else __builtin_nontemporal_store(d2{acc[0][tn][r], acc[1][tn][r]}, yp);
          if (p.tips != nullptr && wr == 0 && c == 0) p.tips[col] = acc[0][tn][r];
        }
      }
  };

  // ---- [A0 X0 A1 X1 A2 X2 A3 X3], 2 pieces per wave each; chunk kc has landed when at most 4 (3 - kc) are outstanding ----
#pragma unroll
  for (int kc = 0; kc < CHUNKS; ++kc) {
    issue_a(kc);
    issue_x(kc);
  }
#pragma unroll
  for (int tm = 0; tm < 2; ++tm)
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = d4{0.0, 0.0, 0.0, 0.0};
  using K0 = integral_constant<int, 0>;
  using K1 = integral_constant<int, 1>;
  using K2 = integral_constant<int, 2>;
  using K3 = integral_constant<int, 3>;
  using FIRST = integral_constant<int, (ABL & 32) ? 3 : 1>;
  wait_vm<12>(); barrier(); read_a(0); compute_chunk(K0{}, FIRST{});
  wait_vm<8>();  barrier(); read_a(1); compute_chunk(K1{}, FIRST{});
  wait_vm<4>();  barrier(); read_a(2); compute_chunk(K2{}, FIRST{});
  wait_vm<0>();  barrier(); read_a(3); compute_chunk(K3{}, FIRST{});
  if constexpr ((ABL & 32) != 0) {
    store_cols(3);
  } else {
    using SECOND = integral_constant<int, 2>;
    store_cols(1);                                    // drains under the second pass
    compute_chunk(K0{}, SECOND{});
    compute_chunk(K1{}, SECOND{});
    compute_chunk(K2{}, SECOND{});
    compute_chunk(K3{}, SECOND{});
    store_cols(2);
  }
}

// ---- host side -------------------------------------------------------------------------------------------
static std::atomic<int> g_blocks_per_cu[16];
static std::mutex g_prepare_mutex;

}  // namespace small

bool small_supported(uint32_t rows) { return rows == kSmallRows; }

// One tile per workgroup, all workgroups resident at once (2 per CU): at most 2 x CUs tiles of 64 columns.
bool small_batch(size_t M, int num_cus, uint32_t max_wgs) {
  size_t resident = static_cast<size_t>(num_cus) * 2;
  if (max_wgs != 0 && max_wgs < resident) resident = max_wgs;
  return (M + small::BN - 1) / small::BN <= resident;
}

void small_pack(const double* A, size_t lda, std::vector<double>& out) {
  using namespace small;
  out.assign(static_cast<size_t>(ROWS) * ROWS, 0.0);
  size_t o = 0;
  for (int kc = 0; kc < CHUNKS; ++kc)
    for (int wr = 0; wr < 2; ++wr)
      for (int tm = 0; tm < 2; ++tm)
        for (int j = 0; j < 2; ++j)
          for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 2; ++e, ++o) {
              const int row = 32 * wr + row_of_slot(tm, l & 15);
              const int k = 16 * kc + k_of_paired(j, l >> 4, e);
              out[o] = A[row + static_cast<size_t>(k) * lda];
            }
#ifdef SRI_TEST_CORRUPT_OPERATOR
  // `make broken` only (never in libsri_hip.so) — as in pack_operator() of sri_apply.hip: every dispatch path of the
  // deliberately wrong library is wrong, also the one-tile kernel a small batch takes
  for (double& v : out)
    if (v != 0.0) { v *= 1.0 + 0x1p-20; break; }
#endif
}

int small_prepare(int dev) {
  using namespace small;
  if (dev < 0 || dev >= 16) return SRI_EUNSUPPORTED;
  if (g_blocks_per_cu[dev].load(std::memory_order_acquire) != 0) return SRI_OK;
  std::lock_guard<std::mutex> lock(g_prepare_mutex);
  if (g_blocks_per_cu[dev].load(std::memory_order_relaxed) != 0) return SRI_OK;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&apply_small_kernel<0>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  int nb = 0;
  if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, apply_small_kernel<0>, THREADS, LDS_BYTES);
  if (e != hipSuccess) {
    record_hip_error(static_cast<int>(e));
    (void)hipGetLastError();
    return SRI_EHIP;
  }
  g_blocks_per_cu[dev].store(nb < 1 ? 1 : (nb > 2 ? 2 : nb), std::memory_order_release);
  return SRI_OK;
}

int small_launch(const double* d_apack, const double* dX, size_t ldx, double* dY, size_t ldy, size_t M, double* d_tips, int dev,
                 int num_cus, uint32_t max_wgs, void* stream) {
  using namespace small;
  const int per_cu = g_blocks_per_cu[dev].load(std::memory_order_acquire);
  if (per_cu == 0) return SRI_EUNSUPPORTED;
  const size_t tiles = (M + BN - 1) / BN;
  if (tiles == 0) return SRI_OK;
  if (!small_batch(M, num_cus, max_wgs)) return SRI_EUNSUPPORTED;
  Args args{d_apack, dX, dY, d_tips, ldx, ldy, M, static_cast<uint32_t>(tiles),
            (ldy % 16 == 0 && reinterpret_cast<uintptr_t>(dY) % 128 == 0) ? 1u : 0u};
  const unsigned grid = static_cast<unsigned>(tiles);                 // one tile per workgroup
  hipError_t e = launch_kernel(apply_small_kernel<0>, dim3(grid), dim3(THREADS), LDS_BYTES, static_cast<hipStream_t>(stream), args);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    record_hip_error(static_cast<int>(e));
    return SRI_EHIP;
  }
  return SRI_OK;
}

}  // namespace sri
