// dma_mfma_interference.hip — measurement tooling.  512-thread workgroups, one per CU: waves 0-3 run a bare
// MFMA f64 loop (one per SIMD), waves 4-7 (their SIMD partners) do one of: nothing / LDS-DMA pieces from an
// L2-resident buffer / ds_read_b128 bursts / 16-byte global stores.  Reports the MFMA waves' cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const double* src, double* dst, int iters, unsigned long long* cyc, int per_iter) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wid < 4) {
    double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane;
    d4 acc[4] = {d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}, d4{0, 0, 0, 0}};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i & 3]) : "v"(a), "v"(b));
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    dst[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][0] + acc[2][0] + acc[3][0];
    if (lane == 0) cyc[blockIdx.x * 4 + wid] = t1 - t0;
  } else {
    // partner waves: roughly `per_iter` operations per 16 MFMAs (1024 cycles) of the MFMA wave
    const char* g = reinterpret_cast<const char*>(src) + ((size_t)(blockIdx.x % 64) * 8 + (wid - 4)) * 16384 + lane * 16;
    d2 sink = {0, 0};
    for (int it = 0; it < iters; ++it) {
      for (int j = 0; j < per_iter; ++j) {
        if (MODE == 1) {
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + (j & 15) * 1024),
                                           (__attribute__((address_space(3))) void*)(smem + (wid - 4) * 16384 + (j & 15) * 1024), 16, 0, 0);
        } else if (MODE == 2) {
          d2 v = *reinterpret_cast<const d2*>(smem + (wid - 4) * 16384 + (j & 15) * 1024 + lane * 16);
          sink += v;
        } else if (MODE == 3) {
          *reinterpret_cast<d2*>(reinterpret_cast<char*>(dst) + (1 << 20) + ((size_t)blockIdx.x * 4 + (wid - 4)) * 16384 + (j & 15) * 1024 + lane * 16) = sink;
        }
      }
      if (MODE == 1 || MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // pace: ~1024 cycles per iteration like the MFMA wave
      __builtin_amdgcn_s_sleep(10);
    }
    if (MODE == 2) dst[(1 << 18) + blockIdx.x * 256 + threadIdx.x - 256] = sink[0] + sink[1];
  }
}

template <int MODE>
static void run(const char* name, const double* src, double* dst, unsigned long long* cyc, int per_iter) {
  const int iters = 2000, blocks = 256;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  k<MODE><<<blocks, 512, 65536>>>(src, dst, iters, cyc, per_iter); CK(hipDeviceSynchronize());
  k<MODE><<<blocks, 512, 65536>>>(src, dst, iters, cyc, per_iter); CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(blocks * 4);
  CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  printf("%-44s per_iter=%2d : cycles per MFMA median %.2f  (p95 %.2f)\n", name, per_iter, h[h.size() / 2] / (iters * 16.0), h[h.size() * 95 / 100] / (iters * 16.0));
}

int main() {
  double *src, *dst; unsigned long long* cyc;
  CK(hipMalloc(&src, 64 * 8 * 16384 + 65536)); CK(hipMemset(src, 0, 64 * 8 * 16384 + 65536));
  CK(hipMalloc(&dst, (1 << 20) + 256 * 4 * 16384 + 65536 * 64)); CK(hipMalloc(&cyc, sizeof(unsigned long long) * 1024));
  run<0>("partner idle (sleeping)", src, dst, cyc, 0);
  for (int n : {2, 6, 12, 24}) run<1>("partner issues LDS-DMA pieces (L2-resident)", src, dst, cyc, n);
  for (int n : {4, 16, 32}) run<2>("partner issues ds_read_b128", src, dst, cyc, n);
  for (int n : {2, 8, 16}) run<3>("partner issues 16-B global stores", src, dst, cyc, n);
  return 0;
}
