// small_diag.hip — where do the microseconds of a small apply go?  (measurement tooling, not product code)
// Unity-includes csrc/sri_apply_small.hip and times, for N = 64 and a given batch, back-to-back launches of
//   an empty kernel with the same grid / block / LDS footprint (the launch + drain floor),
//   the product kernel, and its ablations (no MFMAs / no stores / no reads / combinations).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I<pkg>/csrc tools/small_diag.hip <pkg>/csrc/sri_host.cpp \
//              <pkg>/csrc/sri_apply.hip ... (see Makefile target build/small_diag)
#include "../experimental_gpu_programming_for_a_spectral_numerical_integration_amd/csrc/sri_apply_small.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

namespace sri { void record_hip_error(int) {} }

using namespace sri;
using namespace sri::small;

__global__ __launch_bounds__(THREADS, 2) void empty_kernel(const Args p) {
  extern __shared__ char smem[];
  if (p.tiles == 0xffffffffu) smem[threadIdx.x] = 1;   // keeps the LDS allocation
}

template <class K>
static void time_it(const char* name, K kern, Args a, unsigned grid, int reps, double bytes, double flops) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS_BYTES, 0, a);
  CK(hipDeviceSynchronize());
  float best = 1e30f, sum = 0;
  const int rounds = 5;
  for (int r = 0; r < rounds; ++r) {
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), LDS_BYTES, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms / reps); sum += ms / reps;
  }
  printf("%-44s %8.3f us/launch (min %.3f)  %7.1f GB/s  %6.2f TF/s\n", name, sum / rounds * 1e3, best * 1e3, bytes / (sum / rounds) / 1e6,
         flops / (sum / rounds) / 1e9);
}

int main(int argc, char** argv) {
  const size_t rods = argc > 1 ? atol(argv[1]) : 4096;
  const int reps = argc > 2 ? atoi(argv[2]) : 2000;
  const size_t M = rods * 4, n = ROWS;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  std::vector<double> A(n * n), hx(n * M), packed;
  std::mt19937_64 gen(0x5EED); std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (auto& v : A) v = U(gen);
  for (auto& v : hx) v = U(gen);
  small_pack(A.data(), n, packed);
  double *dA, *dX, *dY;
  CK(hipMalloc(&dA, packed.size() * 8)); CK(hipMalloc(&dX, hx.size() * 8)); CK(hipMalloc(&dY, hx.size() * 8));
  CK(hipMemcpy(dA, packed.data(), packed.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(dX, hx.data(), hx.size() * 8, hipMemcpyHostToDevice));
  const size_t tiles = (M + BN - 1) / BN;
  if (tiles > (size_t)prop.multiProcessorCount * 2) { printf("one tile per workgroup: at most %d rods\n", prop.multiProcessorCount * 2 * 16); return 1; }
  const unsigned grid = (unsigned)tiles;
  Args a{dA, dX, dY, nullptr, n, n, M, (uint32_t)tiles, 1u};
  const double bytes = 16.0 * n * M + 8.0 * n * n, flops = 2.0 * n * n * M;
  printf("N=64 rods=%zu tiles=%zu grid=%u  algorithmic %.2f MB / %.3f GFLOP per launch; HBM floor at 6.29 TB/s %.2f us, MFMA floor %.2f us\n", rods, tiles, grid,
         bytes / 1e6, flops / 1e9, bytes / 6.29e6, flops / 78.6e6);
  time_it("empty kernel, same grid/LDS", empty_kernel, a, grid, reps, bytes, flops);
  time_it("product kernel", apply_small_kernel<0>, a, grid, reps, bytes, flops);
  time_it("no MFMAs", apply_small_kernel<1>, a, grid, reps, bytes, flops);
  time_it("no Y stores", apply_small_kernel<2>, a, grid, reps, bytes, flops);
  time_it("no MFMAs, no stores (loads + LDS reads)", apply_small_kernel<3>, a, grid, reps, bytes, flops);
  time_it("no fragment reads", apply_small_kernel<8>, a, grid, reps, bytes, flops);
  time_it("no MFMA, no stores, no reads (DMA only)", apply_small_kernel<11>, a, grid, reps, bytes, flops);
  time_it("MFMA only (no stores, no reads)", apply_small_kernel<10>, a, grid, reps, bytes, flops);
  time_it("plain (not non-temporal) Y stores", apply_small_kernel<16>, a, grid, reps, bytes, flops);
  time_it("single tile in one column pass", apply_small_kernel<32>, a, grid, reps, bytes, flops);
  time_it("one pass + plain stores", apply_small_kernel<48>, a, grid, reps, bytes, flops);
  time_it("product kernel again", apply_small_kernel<0>, a, grid, reps, bytes, flops);
  return 0;
}
