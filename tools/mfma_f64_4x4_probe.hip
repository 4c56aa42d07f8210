// mfma_f64_4x4_probe.hip — measurement tooling (not product code), round 5.
// v_mfma_f64_4x4x4_4b_f64 on MI355X: lane maps (found by search over exact integers), what one instruction costs next to
// v_mfma_f64_16x16x4_f64, and whether its k = 0..3 sum is the same chain of fused multiply-adds (bit for bit on random doubles).
// The question behind it (EXPERIMENTS.md R5.5): would a 4-row strip on this instruction carry the ONE leftover row of an operator of
// 256 a + 1 rows (N = 257, 513, 1025) cheaper than the 16-row strip on v_mfma_f64_16x16x4_f64 does?
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_4x4_probe.hip -o build/mfma_f64_4x4_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void one_mfma(const double* a, const double* b, const double* c, double* d) {
  const int l = threadIdx.x;
  d[l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[l], b[l], c[l], 0, 0, 0);
}

template <int KIND, int NACC>   // 0: 4x4x4_4b, 1: 16x16x4
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters, unsigned long long* cyc) {
  const int l = threadIdx.x & 63;
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  double s1[NACC];
  d4 s4[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) { s1[i] = 0.0; s4[i] = d4{0.0, 0.0, 0.0, 0.0}; }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      if (KIND == 0) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(s1[i]) : "v"(a), "v"(b));
      else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(s4[i]) : "v"(a), "v"(b));
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += s1[i] + s4[i][0] + s4[i][1] + s4[i][2] + s4[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int NACC>
static void run_rate(const char* name, int blocks, int iters) {
  double* out; unsigned long long* cyc;
  CK(hipMalloc(&out, sizeof(double) * blocks * 256)); CK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  rate_kernel<KIND, NACC><<<blocks, 256>>>(out, iters, cyc); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(e0));
    rate_kernel<KIND, NACC><<<blocks, 256>>>(out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
  }
  const double per_wave = (double)iters * NACC;
  const int waves_per_simd = std::max(1, blocks / 256);
  printf("%-52s %8.3f ms   %.1f cycles per instruction and SIMD at 2.4 GHz (%d wave%s per SIMD)\n", name, best,
         best * 1e6 / per_wave * 2.4 / waves_per_simd, waves_per_simd, waves_per_simd > 1 ? "s" : "");
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  // ---- 1. lane maps, by search: a[l] = 1000 + l, b[l] = 1 (then the roles swapped), c = 0
  double *da, *db, *dc, *dd;
  CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dc, 512)); CK(hipMalloc(&dd, 512));
  std::vector<double> a(64), b(64), c(64, 0.0), d(64);
  auto run = [&]() {
    CK(hipMemcpy(da, a.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 512, hipMemcpyHostToDevice));
    CK(hipMemcpy(dc, c.data(), 512, hipMemcpyHostToDevice));
    one_mfma<<<1, 64>>>(da, db, dc, dd); CK(hipDeviceSynchronize());
    CK(hipMemcpy(d.data(), dd, 512, hipMemcpyDeviceToHost));
  };
  // hypothesis: block = l / 16; A[i][k] at lane 16 blk + 4 k + i; B[k][j] at lane 16 blk + 4 k + j; D[i][j] at lane 16 blk + 4 j + i (or 4 i + j)
  for (int l = 0; l < 64; ++l) { a[l] = 1 + (l % 4) * 3 + ((l / 4) % 4) * 17 + (l / 16) * 101; b[l] = 2 + (l % 4) * 5 + ((l / 4) % 4) * 29 + (l / 16) * 211; }
  run();
  int bad_ij = 0, bad_ji = 0;
  for (int blk = 0; blk < 4; ++blk)
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        double s = 0;
        for (int k = 0; k < 4; ++k) s += a[16 * blk + 4 * k + i] * b[16 * blk + 4 * k + j];
        if (d[16 * blk + 4 * j + i] != s) ++bad_ji;
        if (d[16 * blk + 4 * i + j] != s) ++bad_ij;
      }
  printf("lane maps: A[i][k], B[k][j] at lane 16 blk + 4 k + {i, j};  D[i][j] at lane 16 blk + 4 j + i: %d mismatches;  at 16 blk + 4 i + j: %d mismatches\n", bad_ji, bad_ij);
  // ---- 2. the k sum as a chain of fused multiply-adds in k order, from c: random doubles with cancellation
  unsigned long long x = 88172645463325252ull;
  auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0; };
  int chain_bad[2] = {0, 0}, trials = 0;
  for (int rep = 0; rep < 200; ++rep) {
    for (int l = 0; l < 64; ++l) { a[l] = rnd(); b[l] = rnd(); c[l] = rnd() * 1e-3; }
    run();
    for (int blk = 0; blk < 4; ++blk)
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          const int out_ji = 16 * blk + 4 * j + i, out_ij = 16 * blk + 4 * i + j;
          for (int which = 0; which < 2; ++which) {
            const int o = which ? out_ij : out_ji;
            double s = c[o];
            for (int k = 0; k < 4; ++k) s = fma(a[16 * blk + 4 * k + i], b[16 * blk + 4 * k + j], s);
            if (d[o] != s) ++chain_bad[which];
          }
          ++trials;
        }
  }
  printf("chain of four fused multiply-adds in k order from c: %d of %d differ (D at 4 j + i), %d (D at 4 i + j)\n", chain_bad[0], trials, chain_bad[1]);
  // ---- 3. issue cost
  run_rate<1, 16>("v_mfma_f64_16x16x4_f64, 16 accumulators", 256, 4000);
  run_rate<0, 16>("v_mfma_f64_4x4x4_4b_f64, 16 accumulators", 256, 4000);
  run_rate<0, 16>("v_mfma_f64_4x4x4_4b_f64, 16 accumulators, 2 waves/SIMD", 512, 4000);
  run_rate<0, 1>("v_mfma_f64_4x4x4_4b_f64, dependent (1 accumulator)", 256, 64000);
  run_rate<1, 1>("v_mfma_f64_16x16x4_f64, dependent (1 accumulator)", 256, 64000);
  return 0;
}
