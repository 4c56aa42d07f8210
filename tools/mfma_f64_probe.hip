// mfma_f64_probe.hip — measurement tooling (not product code).
//
// Answers, on a real MI355X, the questions SURVEY.md §7 H2 leaves open before the apply kernel
// is designed:
//   1. C/D and A/B lane maps of v_mfma_f64_16x16x4_f64, checked with exact asymmetric integers.
//   2. back-to-back issue interval of v_mfma_f64_16x16x4_f64 (cycles per MFMA per SIMD) with
//      1 and 2 waves per SIMD, and the chip-wide fp64 MFMA TFLOP/s it sustains.
//   3. the v_fma_f64 (VALU) rate, and whether MFMA f64 and VALU f64 overlap or share hardware.
//
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_probe.hip -o gpurun_out/mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

typedef double d4 __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// ---------------------------------------------------------------- 1. layout check
// One wave. A is 16x4, B is 4x16 (row-major host arrays). Lane l feeds A[l&15][l>>4], B[l>>4][l&15].
// Output: each lane dumps its 4 regs; host checks D[(l>>4)+4*r][l&15].
__global__ void layout_kernel(const double* A, const double* B, double* out) {
  int l = threadIdx.x;
  double a = A[(l & 15) * 4 + (l >> 4)];
  double b = B[(l >> 4) * 16 + (l & 15)];
  d4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = acc[r];
}

// ---------------------------------------------------------------- 2/3. rate kernels
// MODE 0: MFMA only (NACC independent accumulators)
// MODE 1: VALU v_fma_f64 only
// MODE 2: MFMA + VALU interleaved in the same wave (1 MFMA : VF fma)
template <int MODE, int NACC, int VF>
__global__ __launch_bounds__(256) void rate_kernel(double* out, int iters, unsigned long long* cyc) {
  int l = threadIdx.x & 63;
  double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.5 * i + l;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) {
      // inline asm: the builtin form makes hipcc (ROCm 7.2) shuttle every accumulator between
      // VGPRs and AGPRs once per iteration (256 v_accvgpr moves per 16 MFMAs), which is what a
      // first version of this probe measured instead of the MFMA.
      if (MODE == 0 || MODE == 2)
        asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
      if (MODE == 1 || MODE == 2) {
#pragma unroll
        for (int j = 0; j < VF; ++j)
          asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(a), "v"(b));
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---------------------------------------------------------------- 4. power-limited rate on RANDOM operands
// 16 accumulators, 16 distinct random A and B operand registers per lane (permuted pairs, so no
// memory traffic at all): what the MFMA pipe sustains once DVFS has settled on realistic data.
// Reports TFLOP/s and the in-kernel clock = d(s_memtime) / d(s_memrealtime) * 100 MHz.
__global__ __launch_bounds__(256) void random_rate_kernel(const double* rnd, double* out, int iters,
                                                          unsigned long long* cyc, int zero) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  // 16 + 16 distinct random operand registers per lane; MFMA i of an iteration uses (a[i], b[(5i+it)&15]):
  // every MFMA sees an operand pair different from its predecessor's, as in a real GEMM stream.
  double a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = zero ? 0.0 : rnd[(t * 32 + i) & 0xFFFFF]; b[i] = zero ? 0.0 : rnd[(t * 32 + 16 + i) & 0xFFFFF]; }
  d4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i]), "v"(b[(5 * i) & 15]));
#pragma unroll
    for (int i = 0; i < 16; ++i)
      asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[(i + 7) & 15]), "v"(b[(3 * i + 1) & 15]));
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[t] = s;
  if (threadIdx.x == 0) { cyc[blockIdx.x * 2] = t1 - t0; cyc[blockIdx.x * 2 + 1] = r1 - r0; }
}

static void run_random(const char* name, int blocks, int iters, int zero, int launches) {
  const int threads = 256;
  std::vector<double> h(1 << 20);
  unsigned long long x = 88172645463325252ull;
  for (auto& v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0; }
  double *rnd, *out; unsigned long long* cyc;
  CK(hipMalloc(&rnd, sizeof(double) * h.size())); CK(hipMemcpy(rnd, h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
  CK(hipMalloc(&out, sizeof(double) * blocks * threads)); CK(hipMalloc(&cyc, sizeof(unsigned long long) * 2 * blocks));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float last = 0;
  for (int l = 0; l < launches; ++l) {   // several back-to-back launches: let DVFS settle, report the last
    CK(hipEventRecord(e0));
    random_rate_kernel<<<blocks, threads>>>(rnd, out, iters, cyc, zero);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&last, e0, e1));
  }
  std::vector<unsigned long long> c(2 * blocks);
  CK(hipMemcpy(c.data(), cyc, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
  std::vector<double> ghz;
  for (int b = 0; b < blocks; ++b) ghz.push_back((double)c[2 * b] / (double)c[2 * b + 1] * 0.1);
  std::sort(ghz.begin(), ghz.end());
  double flop = (double)blocks * 4 * iters * 16.0 * 2048.0;
  printf("%-40s %8.3f ms  %7.2f TFLOP/s  in-kernel clock median %.3f GHz  -> %.1f cyc/MFMA/SIMD\n", name, last,
         flop / (last * 1e-3) / 1e12, ghz[ghz.size() / 2], (double)c[0] / (iters * 16.0) / (blocks / 256));
  CK(hipFree(rnd)); CK(hipFree(out)); CK(hipFree(cyc));
}

template <int MODE, int NACC, int VF>
static void run_rate(const char* name, int blocks, int threads, int iters) {
  double* out; unsigned long long* cyc;
  CK(hipMalloc(&out, sizeof(double) * blocks * threads));
  CK(hipMalloc(&cyc, sizeof(unsigned long long) * blocks));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // warm-up
  rate_kernel<MODE, NACC, VF><<<blocks, threads>>>(out, iters, cyc);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    rate_kernel<MODE, NACC, VF><<<blocks, threads>>>(out, iters, cyc);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  std::vector<unsigned long long> h(blocks);
  CK(hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost));
  double med = 0; { std::vector<unsigned long long> s = h; std::sort(s.begin(), s.end()); med = (double)s[s.size() / 2]; }
  double waves = (double)blocks * threads / 64.0;
  double n_mfma = (MODE == 1) ? 0.0 : (double)iters * NACC;
  double n_fma = (MODE == 0) ? 0.0 : (double)iters * NACC * VF;
  double flop = waves * (n_mfma * 2048.0 + n_fma * 128.0);
  double per_wave_instr = (MODE == 1) ? n_fma : n_mfma;
  printf("%-34s blocks=%5d thr=%4d  %8.3f ms  %7.2f TFLOP/s  memtime-ticks/%s=%.2f (ticks are 100MHz if constant clock)\n",
         name, blocks, threads, best, flop / (best * 1e-3) / 1e12, MODE == 1 ? "fma" : "mfma", med / per_wave_instr);
  // derive cycles per MFMA per SIMD from wall time assuming 2.4 GHz
  int waves_per_simd = (threads / 64) * (blocks / 256) / 4; if (waves_per_simd < 1) waves_per_simd = 1;
  printf("    -> wall ns per instr per wave = %.2f ; x2.4GHz = %.1f cyc ; per SIMD (/%d waves) = %.1f cyc\n",
         best * 1e6 / per_wave_instr, best * 1e6 / per_wave_instr * 2.4, waves_per_simd,
         best * 1e6 / per_wave_instr * 2.4 / waves_per_simd);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  clock=%d kHz  arch=%s\n", p.name, p.multiProcessorCount, p.clockRate, p.gcnArchName);

  // 1. layout
  {
    std::vector<double> A(64), B(64), D(256, 0.0), out(256);
    for (int i = 0; i < 16; ++i) for (int k = 0; k < 4; ++k) A[i * 4 + k] = (double)(1 + i * 7 + k * 131);
    for (int k = 0; k < 4; ++k) for (int j = 0; j < 16; ++j) B[k * 16 + j] = (double)(3 + k * 17 + j * j * 5 + j);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 16 + j]; D[i * 16 + j] = s; }
    double *dA, *dB, *dO; CK(hipMalloc(&dA, 512)); CK(hipMalloc(&dB, 512)); CK(hipMalloc(&dO, 2048));
    CK(hipMemcpy(dA, A.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 512, hipMemcpyHostToDevice));
    layout_kernel<<<1, 64>>>(dA, dB, dO); CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dO, 2048, hipMemcpyDeviceToHost));
    int bad_f64map = 0, bad_f32map = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
      double got = out[l * 4 + r];
      if (got != D[((l >> 4) + 4 * r) * 16 + (l & 15)]) ++bad_f64map;
      if (got != D[((l >> 4) * 4 + r) * 16 + (l & 15)]) ++bad_f32map;
    }
    printf("layout: mismatches with f64 map row=(l>>4)+4r : %d ; with f32-style map row=4(l>>4)+r : %d\n", bad_f64map, bad_f32map);
  }

  const int IT = 4000;
  // 2. MFMA only
  run_rate<0, 16, 0>("mfma f64, 1 wave/SIMD, 16 acc", 256, 256, IT);
  run_rate<0, 16, 0>("mfma f64, 2 waves/SIMD, 16 acc", 512, 256, IT);
  run_rate<0, 4, 0>("mfma f64, 1 wave/SIMD, 4 acc", 256, 256, IT * 4);
  run_rate<0, 2, 0>("mfma f64, 1 wave/SIMD, 2 acc", 256, 256, IT * 8);
  run_rate<0, 1, 0>("mfma f64, 1 wave/SIMD, 1 acc (dep)", 256, 256, IT * 16);
  // 3. VALU only
  run_rate<1, 16, 8>("v_fma_f64, 1 wave/SIMD", 256, 256, IT);
  run_rate<1, 16, 8>("v_fma_f64, 2 waves/SIMD", 512, 256, IT);
  run_rate<1, 16, 8>("v_fma_f64, 4 waves/SIMD", 1024, 256, IT);
  // 4. mixed in one wave
  run_rate<2, 16, 4>("mfma + 4 v_fma_f64 per mfma", 256, 256, IT);
  run_rate<2, 16, 8>("mfma + 8 v_fma_f64 per mfma", 256, 256, IT);
  run_rate<2, 16, 16>("mfma + 16 v_fma_f64 per mfma", 256, 256, IT);
  run_rate<2, 16, 8>("mfma + 8 v_fma_f64, 2 waves/SIMD", 512, 256, IT);
  // 5. power-limited: ~60 ms launches, 6 back to back
  run_random("mfma f64 RANDOM operands, 1 wave/SIMD", 256, 130000, 0, 6);
  run_random("mfma f64 RANDOM operands, 2 waves/SIMD", 512, 65000, 0, 6);
  run_random("mfma f64 ZERO operands,   2 waves/SIMD", 512, 65000, 1, 6);
  return 0;
}
