// unaligned_probe.hip — measurement tooling: do global_load_lds_dwordx4 and global_store_dwordx4 work
// (and how fast) when the global address is only 8-byte aligned?  Decides whether the LDS-DMA path can
// serve panels with an odd number of rows (255, 1023: every integration operator).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ void dma_copy(const double* src, double* dst, int shift_doubles, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
    const double* g = src + shift_doubles + ((size_t)(blockIdx.x * iters + it) * 4 + wid) * 128 + lane * 2;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(smem + wid * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    d2 v = *reinterpret_cast<const d2*>(smem + wid * 1024 + lane * 16);
    // 16-byte store to an address with the same (mis)alignment
    *reinterpret_cast<d2*>(dst + shift_doubles + ((size_t)(blockIdx.x * iters + it) * 4 + wid) * 128 + lane * 2) = v;
  }
}

int main() {
  const int blocks = 2048, iters = 64;
  const size_t n = (size_t)blocks * iters * 4 * 128 + 16;
  std::vector<double> h(n);
  for (size_t i = 0; i < n; ++i) h[i] = (double)i * 0.5 + 1.0;
  double *s, *d; CK(hipMalloc(&s, n * 8)); CK(hipMalloc(&d, n * 8));
  CK(hipMemcpy(s, h.data(), n * 8, hipMemcpyHostToDevice));
  for (int shift = 0; shift <= 1; ++shift) {
    CK(hipMemset(d, 0, n * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dma_copy<<<blocks, 256, 4096>>>(s, d, shift, iters); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 10; ++r) dma_copy<<<blocks, 256, 4096>>>(s, d, shift, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<double> o(n);
    CK(hipMemcpy(o.data(), d, n * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = shift; i < n - 16 + shift; ++i) if (o[i] != h[i]) ++bad;
    printf("shift %d doubles (%s-byte aligned): mismatches %zu of %zu ; %.3f ms per pass = %.0f GB/s read+write\n", shift,
           shift ? "8" : "16", bad, n - 16, ms / 10, 2.0 * (n - 16) * 8 / (ms / 10 * 1e-3) / 1e9);
  }
  return 0;
}
