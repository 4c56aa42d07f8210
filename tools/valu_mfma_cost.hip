// valu_mfma_cost.hip — measurement tooling (not product code).
//
// What does one vector-ALU instruction cost the fp64 MFMA stream of its SIMD?  One 512-thread workgroup per CU, two waves
// per SIMD, every wave runs the same loop: 16 independent v_mfma_f64_16x16x4_f64 followed by K instructions of one kind.
// Without the extra instructions a SIMD issues one MFMA per 64 cycles (two waves: 128 cycles per MFMA and wave); the
// slope of (cycles per 16 MFMAs) against K is the MFMA time one such instruction takes away.
// build: make build/valu_mfma_cost
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

enum { K_NONE, K_MOV32, K_MOV64, K_ADD32, K_ADD64, K_MUL32, K_FMA64, K_SALU, K_CNDMASK, K_DSREAD128, K_DSREAD64 };
typedef double d2 __attribute__((ext_vector_type(2)));

template <int KIND, int K, bool SPREAD = false>
__global__ __launch_bounds__(512) void k(double* dst, int iters, unsigned long long* cyc) {
  const int lane = threadIdx.x & 63;
  const double a = 1.0 + 1e-9 * lane, b = 1.0 - 1e-9 * lane;
  d4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
  unsigned v0 = lane, v1 = lane * 3;
  unsigned long long w0 = lane, w1 = lane * 5;
  double f0 = lane, f1 = 1.0;
  unsigned s0 = blockIdx.x;
  __shared__ __attribute__((aligned(16))) char lds[8 * 16384];
  for (int i = threadIdx.x; i < 8 * 16384 / 8; i += 512) reinterpret_cast<double*>(lds)[i] = 1.0;
  __syncthreads();
  const unsigned rd = (threadIdx.x >> 6) * 16384 + lane * 16;   // contiguous 1 KiB per read: conflict-free
  d2 sink2 = {0, 0};
  double sink1 = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (SPREAD) {                                            // K / 16 of the extra instructions behind every MFMA
#pragma unroll
        for (int j = 0; j < K / 16; ++j) {
          if (KIND == K_ADD32) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v0) : "v"(v1));
          if (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
        }
      }
    }
#pragma unroll
    for (int j = 0; j < (SPREAD ? 0 : K); ++j) {
      if (KIND == K_MOV32) asm volatile("v_mov_b32 %0, %1" : "=v"(v0) : "v"(v1));
      if (KIND == K_MOV64) asm volatile("v_mov_b64 %0, %1" : "=v"(w0) : "v"(w1));
      if (KIND == K_ADD32) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v0) : "v"(v1));
      if (KIND == K_ADD64) asm volatile("v_lshl_add_u64 %0, %1, 0, %0" : "+v"(w0) : "v"(w1));
      if (KIND == K_MUL32) asm volatile("v_mul_lo_u32 %0, %1, %0" : "+v"(v0) : "v"(v1));
      if (KIND == K_FMA64) asm volatile("v_fma_f64 %0, %1, %1, %0" : "+v"(f0) : "v"(f1));
      if (KIND == K_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(s0) : : "scc");
      if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %1, %0, vcc" : "+v"(v0) : "v"(v1) : "vcc");
      if (KIND == K_DSREAD128) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink2) : "v"(rd), "n"((j & 15) * 1024));
      if (KIND == K_DSREAD64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(sink1) : "v"(rd), "n"((j & 15) * 1024));
    }
    if (KIND == K_DSREAD128 || KIND == K_DSREAD64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double sum = sink2[0] + sink1 + f0 + static_cast<double>(v0) + static_cast<double>(w0) + static_cast<double>(s0);
#pragma unroll
  for (int i = 0; i < 16; ++i) sum += acc[i][0];
  if (sum == 12345.678) dst[0] = 1;
  if (lane == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int K, bool SPREAD = false>
static double run(double* dst, unsigned long long* cyc) {
  const int iters = 4000, blocks = 256;
  for (int rep = 0; rep < 2; ++rep) {
    k<KIND, K, SPREAD><<<blocks, 512>>>(dst, iters, cyc);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> h(blocks * 8);
  CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * h.size(), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  return double(h[h.size() / 2]) / iters;
}

template <int KIND>
static void kind(const char* name, double* dst, unsigned long long* cyc, double base) {
  const double c8 = run<KIND, 8>(dst, cyc), c32 = run<KIND, 32>(dst, cyc);
  // two waves per SIMD execute K instructions each per 16 + 16 MFMAs: MFMA time lost per instruction = slope / 2... the
  // wave's own period grows by (c - base); both waves' instructions share it, so per instruction: (c - base) / (2 K)
  std::printf("%-22s cycles per 16 MFMAs of a wave: K=8 %7.1f  K=32 %7.1f   -> MFMA time per instruction %5.2f / %5.2f cycles\n", name, c8,
              c32, (c8 - base) / 16.0, (c32 - base) / 64.0);
}

int main() {
  double* dst;
  unsigned long long* cyc;
  CK(hipMalloc(&dst, 4096));
  CK(hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8));
  const double base = run<K_NONE, 0>(dst, cyc);
  std::printf("no extra instructions: %.1f cycles per 16 MFMAs of a wave (ideal 2048: two waves share the SIMD)\n", base);
  kind<K_MOV32>("v_mov_b32", dst, cyc, base);
  kind<K_MOV64>("v_mov_b64", dst, cyc, base);
  kind<K_ADD32>("v_add_u32", dst, cyc, base);
  kind<K_ADD64>("v_lshl_add_u64", dst, cyc, base);
  kind<K_MUL32>("v_mul_lo_u32", dst, cyc, base);
  kind<K_FMA64>("v_fma_f64", dst, cyc, base);
  kind<K_CNDMASK>("v_cndmask_b32", dst, cyc, base);
  kind<K_DSREAD128>("ds_read_b128", dst, cyc, base);
  kind<K_DSREAD64>("ds_read_b64", dst, cyc, base);
  kind<K_SALU>("s_add_u32 (scalar)", dst, cyc, base);
  // the same count, two behind every MFMA instead of 32 in a row behind the 16th
  const double sv = run<K_ADD32, 32, true>(dst, cyc), ss = run<K_SALU, 32, true>(dst, cyc);
  std::printf("spread over the MFMAs (2 behind each): v_add_u32 %7.1f -> %5.2f cycles each;  s_add_u32 %7.1f -> %5.2f cycles each\n", sv,
              (sv - base) / 64.0, ss, (ss - base) / 64.0);
  return 0;
}
