// stream_probe.hip — what HBM rate do the apply kernels' access PATTERNS allow, with no arithmetic at all?
// Persistent workgroups (256 threads, 2 per CU) walk 32-KiB tiles of a 64-row fp64 panel (64 columns x 512 B, contiguous):
//   read  pattern 0: LDS-DMA, one wave instruction = 8 columns x 128 B (the kernels' k-chunk order: 4 passes over a tile)
//   read  pattern 1: LDS-DMA, one wave instruction = 1 KiB contiguous (2 whole columns)
//   read  pattern 2: global_load_dwordx4 into registers, 1 KiB contiguous per wave instruction
//   write pattern 0: 16-byte stores, a wave instruction = 4 x 256 B in 4 columns (the kernels' Y stores)
//   write pattern 1: 16-byte stores, 1 KiB contiguous per wave instruction
//   nt: non-temporal stores
// Prints GB/s for read-only, write-only and both (the apply's mix: every byte read is matched by a byte written).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

typedef double d2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void lds_dma16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int RP, int WP, bool NT, bool DO_R, bool DO_W>
__global__ __launch_bounds__(256, 2) void probe(const char* x, char* y, unsigned tiles, double* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 x 32 KiB
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  d2 acc = d2{0.0, 0.0};
  unsigned it = 0;
  for (unsigned tile = blockIdx.x; tile < tiles; tile += gridDim.x, ++it) {
    const char* src = x + (size_t)tile * 32768;
    char* dst = y + (size_t)tile * 32768;
    char* buf = smem + (it & 1) * 32768;
    if (DO_R) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int piece = i * 4 + wid;                       // 32 one-KiB pieces per tile
        if (RP == 0) {
          const int kc = piece >> 3, cg = piece & 7;         // chunk kc, columns 8cg..8cg+7, 128 B each
          lds_dma16(src + (cg * 8 + (lane >> 3)) * 512 + kc * 128 + (lane & 7) * 16, buf + piece * 1024);
        } else if (RP == 1) {
          lds_dma16(src + piece * 1024 + lane * 16, buf + piece * 1024);
        } else {
          const d2 v = *reinterpret_cast<const d2*>(src + piece * 1024 + lane * 16);
          acc += v;
        }
      }
    }
    if (DO_W) {
      const d2 v = d2{(double)tile, (double)lane};
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        char* p;
        if (WP == 0) {
          const int c = lane & 15, g = lane >> 4;            // as the kernels: rows 2c,2c+1 of column g + 4i (+ wave offsets)
          p = dst + ((wid & 1) * 32 + (i >> 2) * 16 + g + 4 * (i & 3)) * 512 + ((wid >> 1) * 32 + 2 * c) * 8;
        } else {
          p = dst + (i * 4 + wid) * 1024 + lane * 16;
        }
        if (NT) __builtin_nontemporal_store(v, reinterpret_cast<d2*>(p)); else *reinterpret_cast<d2*>(p) = v;
      }
    }
    if (DO_R && RP != 2) {                                  // keep two tiles of reads in flight per workgroup
      if (it >= 1) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DO_W ? 16 : 8) : "memory"); }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc[0] == 12345.678 && sink) sink[0] = acc[1];
}

template <class K>
static double run(K k, const char* x, char* y, unsigned tiles, unsigned grid, int reps) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 65536, 0, x, y, tiles, (double*)nullptr);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 65536, 0, x, y, tiles, (double*)nullptr);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const unsigned tiles = argc > 1 ? atoi(argv[1]) : 16384;   // 16384 x 32 KiB = 512 MiB per direction
  const size_t bytes = (size_t)tiles * 32768;
  char *x, *y; CK(hipMalloc(&x, bytes)); CK(hipMalloc(&y, bytes)); CK(hipMemset(x, 1, bytes));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const unsigned grid = prop.multiProcessorCount * 2;
  const int reps = 20;
  auto rep = [&](const char* name, double ms, double b) { printf("%-64s %8.1f us  %7.0f GB/s\n", name, ms * 1e3, b / ms / 1e6); };
  rep("read only,  DMA 8 columns x 128 B per instruction", run(probe<0, 0, false, true, false>, x, y, tiles, grid, reps), bytes);
  rep("read only,  DMA 1 KiB contiguous per instruction", run(probe<1, 0, false, true, false>, x, y, tiles, grid, reps), bytes);
  rep("read only,  register loads, 1 KiB contiguous", run(probe<2, 0, false, true, false>, x, y, tiles, grid, reps), bytes);
  rep("write only, 4 x 256 B per instruction", run(probe<0, 0, false, false, true>, x, y, tiles, grid, reps), bytes);
  rep("write only, 4 x 256 B per instruction, non-temporal", run(probe<0, 0, true, false, true>, x, y, tiles, grid, reps), bytes);
  rep("write only, 1 KiB contiguous", run(probe<0, 1, false, false, true>, x, y, tiles, grid, reps), bytes);
  rep("write only, 1 KiB contiguous, non-temporal", run(probe<0, 1, true, false, true>, x, y, tiles, grid, reps), bytes);
  rep("read + write, kernel patterns (8x128 B DMA, 4x256 B stores)", run(probe<0, 0, false, true, true>, x, y, tiles, grid, reps), 2.0 * bytes);
  rep("read + write, kernel patterns, non-temporal stores", run(probe<0, 0, true, true, true>, x, y, tiles, grid, reps), 2.0 * bytes);
  rep("read + write, contiguous DMA, kernel stores, non-temporal", run(probe<1, 0, true, true, true>, x, y, tiles, grid, reps), 2.0 * bytes);
  rep("read + write, contiguous DMA, contiguous stores", run(probe<1, 1, false, true, true>, x, y, tiles, grid, reps), 2.0 * bytes);
  rep("read + write, contiguous DMA, contiguous stores, non-temporal", run(probe<1, 1, true, true, true>, x, y, tiles, grid, reps), 2.0 * bytes);
  rep("read + write, register loads, contiguous stores (a copy)", run(probe<2, 1, false, true, true>, x, y, tiles, grid, reps), 2.0 * bytes);
  rep("read + write, register loads, contiguous stores, non-temporal", run(probe<2, 1, true, true, true>, x, y, tiles, grid, reps), 2.0 * bytes);
  return 0;
}
