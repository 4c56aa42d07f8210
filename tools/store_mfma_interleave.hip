// store_mfma_interleave.hip — measurement tooling (not product code).
//
// Question: what do a tile's 32 Y stores per wave cost the MFMA stream, as a function of WHERE they are issued?
// One 512-thread workgroup per CU (two waves per SIMD, like the 256x128 geometry), no LDS traffic, no barriers: each wave
// runs "tiles" of 1024 MFMAs on 16 accumulators (32 stages of 2 MFMAs per accumulator) and writes its 32 KiB per tile.
//   mode 0  no stores                                                       (the MFMA stream alone)
//   mode 1  burst: the 32 stores behind the tile's last MFMA                (what the product kernel does)
//   mode 2  trickle: the last stage runs accumulator-major (both MFMAs of accumulator i back to back) and the two stores
//           of accumulator i-2 — final for >= 4 MFMAs, so no dependency stall — follow: one store per MFMA
//   mode 3  burst, second wave of every SIMD half a stage behind the first   (the product kernel's staggered groups)
//   mode 4  trickle, staggered the same way
// Reports cycles per tile (ideal 1024 x 64 x 2 = 131 072) and the per-tile cost of the stores against mode 0.
// build: make build/store_mfma_interleave
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                         \
  do {                                                                                                \
    hipError_t e_ = (x);                                                                              \
    if (e_ != hipSuccess) {                                                                           \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);     \
      std::exit(2);                                                                                   \
    }                                                                                                 \
  } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mfma(d4& acc, double a, double b) {
  asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
__device__ __forceinline__ void store16(char* p, d2 v) {
  asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_acc(char* base, int i, const d4& acc) {
  store16(base + (2 * i) * 1024, __builtin_shufflevector(acc, acc, 0, 1));
  store16(base + (2 * i + 1) * 1024, __builtin_shufflevector(acc, acc, 2, 3));
}

template <int MODE>
__global__ __launch_bounds__(512) void k(char* dst, int tiles, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane;
  d4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
  constexpr bool STORES = MODE != 0, TRICKLE = MODE == 2 || MODE == 4, STAGGER = MODE == 3 || MODE == 4;
  // 32 KiB per wave and tile, streaming through a window of 16 tiles
  char* mine = dst + (static_cast<size_t>(blockIdx.x) * 8 + wid) * (16u << 15) + lane * 16;
  if (STAGGER && wid >= 4) {                                   // half a stage behind: 16 extra MFMAs up front
#pragma unroll
    for (int i = 0; i < 16; ++i) mfma(acc[i], a, b);
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int t = 0; t < tiles; ++t) {
    char* out = mine + (static_cast<size_t>(t & 15) << 15);
    for (int s = 0; s < 31; ++s) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 16; ++i) mfma(acc[i], a, b);
    }
    if constexpr (!TRICKLE) {
#pragma unroll
      for (int e = 0; e < 2; ++e)
#pragma unroll
        for (int i = 0; i < 16; ++i) mfma(acc[i], a, b);
      if constexpr (STORES) {
#pragma unroll
        for (int i = 0; i < 16; ++i) store_acc(out, i, acc[i]);
      }
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        mfma(acc[i], a, b);
        mfma(acc[i], a, b);
        if (i >= 2) store_acc(out, i - 2, acc[i - 2]);
      }
      store_acc(out, 14, acc[14]);
      store_acc(out, 15, acc[15]);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double sum = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += acc[i][0];
  if (sum == 12345.678) dst[0] = 1;                            // keep the accumulators live
  if (lane == 0) cyc[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int MODE>
static double run(const char* name, char* dst, unsigned long long* cyc, double base) {
  const int tiles = 64, blocks = 256;
  for (int rep = 0; rep < 3; ++rep) {
    k<MODE><<<blocks, 512>>>(dst, tiles, cyc);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(blocks * 8);
  CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double med = double(h[h.size() / 2]) / tiles, p95 = double(h[h.size() * 95 / 100]) / tiles;
  std::printf("%-58s cycles per tile: median %9.0f  p95 %9.0f  (%.1f %% over the ideal 131072", name, med, p95,
              100.0 * (med / 131072.0 - 1.0));
  if (base > 0) std::printf("; stores cost %6.0f cycles per tile = %.2f %%", med - base, 100.0 * (med - base) / base);
  std::printf(")\n");
  return med;
}

int main() {
  char* dst;
  unsigned long long* cyc;
  const size_t bytes = static_cast<size_t>(256) * 8 * (16u << 15) + 4096;   // 1 GiB window
  CK(hipMalloc(&dst, bytes));
  CK(hipMemset(dst, 0, bytes));
  CK(hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8));
  const double base = run<0>("0: no stores", dst, cyc, 0);
  run<1>("1: burst behind the tile's last MFMA", dst, cyc, base);
  run<2>("2: trickle, one store per MFMA of the last stage", dst, cyc, base);
  const double base_s = run<0>("0: no stores (again)", dst, cyc, 0);
  run<3>("3: burst, partner waves half a stage apart", dst, cyc, base_s);
  run<4>("4: trickle, partner waves half a stage apart", dst, cyc, base_s);
  return 0;
}
