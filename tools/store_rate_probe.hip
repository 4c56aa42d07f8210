// store_rate_probe.hip — measurement tooling: how long does a wave take to ISSUE the 32 16-byte Y stores of a
// finished 64x64 wave tile (the epilogue of sri_apply.hip), and how long until they have completed, as a function
// of how many waves of the CU store at the same time?  Address pattern of the product kernel: per instruction four
// quarter-waves write 256 contiguous bytes each (16 lanes x 2 rows) in four different columns of a column-major panel.
//   hipcc --offload-arch=gfx950 -O3 -o build/store_rate_probe tools/store_rate_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

// MODE 0: 32 x 16-byte stores (product pattern)   MODE 1: 64 x 8-byte stores, same bytes
// MODE 2: 32 x 16-byte stores, each instruction 1 KiB contiguous (one column, 128 rows)
// MODE 3: as 0 with non-temporal stores.   `active` < waves of the block: only the first `active` waves store (the
// rest wait at the barriers, like the partner wave group of sri_apply.hip at a tile end).
template <int MODE>
__global__ void store_probe(double* y, size_t ld, unsigned long long* out, int rounds, int active) {
  extern __shared__ char lds_[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int c = lane & 15, g = lane >> 4;
  d2 v[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) v[i] = d2{(double)(lane + i), (double)(wid - i)};
  unsigned long long issue = 0, done = 0;
  for (int r = 0; r < rounds; ++r) {
    // a fresh 64-row x 64-column wave tile per round: column block (blockIdx, wid, r)
    const size_t col0 = (((size_t)blockIdx.x * nw + wid) * rounds + r) * 64;
    __builtin_amdgcn_s_barrier();                                    // all waves of the CU start together
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (wid < active)
#pragma unroll
    for (int tn = 0; tn < 4; ++tn)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int i = (tn * 4 + q) * 2 + t;
          if (MODE == 0) {
            double* p = y + (col0 + tn * 16 + q * 4 + g) * ld + 32 * t + 2 * c;
            *reinterpret_cast<d2*>(p) = v[i];
          } else if (MODE == 3) {
            double* p = y + (col0 + tn * 16 + q * 4 + g) * ld + 32 * t + 2 * c;
            __builtin_nontemporal_store(v[i], reinterpret_cast<d2*>(p));
          } else if (MODE == 1) {
            double* p = y + (col0 + tn * 16 + q * 4 + g) * ld + 32 * t + 2 * c;
            p[0] = v[i][0];
            p[1] = v[i][1];
          } else {
            double* p = y + (col0 + i) * ld + 2 * lane;
            *reinterpret_cast<d2*>(p) = v[i];
          }
        }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t2 = __builtin_amdgcn_s_memtime();
    issue += t1 - t0;
    done += t2 - t0;
  }
  if (lane == 0 && lds_ != nullptr) {
    out[((size_t)blockIdx.x * nw + wid) * 2] = issue / rounds;
    out[((size_t)blockIdx.x * nw + wid) * 2 + 1] = done / rounds;
  }
}

template <int MODE>
static void run(const char* name, int waves, int blocks, int active = -1, int lds = 0) {
  if (active < 0) active = waves;
  const int rounds = 16;
  const size_t ld = 256, cols = (size_t)blocks * waves * rounds * 64;
  double* y; unsigned long long* o;
  CK(hipMalloc(&y, cols * ld * sizeof(double)));
  CK(hipMalloc(&o, (size_t)blocks * waves * 2 * sizeof(unsigned long long)));
  if (lds > 0) CK(hipFuncSetAttribute(reinterpret_cast<const void*>(store_probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  CK(hipMemset(o, 0, (size_t)blocks * waves * 2 * sizeof(unsigned long long)));
  for (int rep = 0; rep < 3; ++rep) store_probe<MODE><<<blocks, waves * 64, lds>>>(y, ld, o, rounds, active);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h((size_t)blocks * waves * 2);
  CK(hipMemcpy(h.data(), o, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<unsigned long long> is, dn;
  for (size_t i = 0; i < h.size(); i += 2) { if ((int)((i / 2) % waves) < active) { is.push_back(h[i]); dn.push_back(h[i + 1]); } }
  std::sort(is.begin(), is.end()); std::sort(dn.begin(), dn.end());
  // s_memtime counts shader-clock cycles
  printf("%-34s waves/CU %d (storing %d, LDS %3d KiB) blocks %4d : issue of 32 KiB/wave median %5llu cycles (p90 %5llu) ; until complete median %5llu cycles (p90 %5llu)\n",
         name, waves, active, lds / 1024, blocks, is[is.size() / 2], is[is.size() * 9 / 10], dn[dn.size() / 2], dn[dn.size() * 9 / 10]);
  CK(hipFree(y)); CK(hipFree(o));
}

int main() {
  for (int waves : {1, 4, 8}) {
    run<0>("16-B stores, product pattern", waves, 256);
    run<1>("8-B stores, same bytes", waves, 256);
    run<2>("16-B stores, 1 KiB contiguous", waves, 256);
  }
  run<0>("16-B stores, product pattern", 4, 32);     // few CUs: no memory-side contention
  run<0>("16-B stores, product pattern", 8, 32);
  // the apply kernel's situation: 8 waves per CU, 144 KiB of LDS (one workgroup per CU), 4 waves store, 4 wait
  for (int blocks : {32, 256}) {
    run<0>("16-B stores, 4 of 8 waves store", 8, blocks, 4, 0);
    run<0>("... and 144 KiB LDS", 8, blocks, 4, 144 * 1024);
    run<3>("... non-temporal", 8, blocks, 4, 144 * 1024);
    run<3>("non-temporal, 4 waves", 4, blocks, 4, 0);
  }
  return 0;
}
