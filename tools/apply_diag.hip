// apply_diag.hip — diagnostic harness for the apply kernel (measurement tooling, not product code).
//
// Unity-includes csrc/sri_apply.hip to reach the kernel templates, then:
//   1. prints the occupancy the runtime reports for each kernel variant,
//   2. times the product kernel on random vs all-zero panels (DVFS check, cdna guide rule 25),
//   3. runs a STAMP build and prints the per-workgroup timeline: shader clock, phase shares
//      (prologue / main loop / epilogue), start skew, tail.
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -I<pkg>/csrc tools/apply_diag.hip \
//              <pkg>/csrc/sri_host.cpp -o build/apply_diag
#include "../experimental_gpu_programming_for_a_spectral_numerical_integration_amd/csrc/sri_apply.hip"

#include <algorithm>
#include <cstdlib>
#include <random>
#include <string>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

using namespace sri;

template <class G, bool DMA, bool STAMP, int ABL = 0, bool TAIL = false>
static float time_kernel(ApplyArgs a, size_t col_tiles, int reps, float* min_ms = nullptr) {
  a.col_tiles = (uint32_t)col_tiles;
  auto kern = apply_kernel<G, DMA, false, TAIL, STAMP, ABL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, G::THREADS, G::LDS_BYTES));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const size_t tiles = col_tiles * a.row_blocks;
  const unsigned blocks = (unsigned)std::min<size_t>(tiles, (size_t)per_cu * prop.multiProcessorCount);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(G::THREADS), G::LDS_BYTES, 0, a);
  CK(hipDeviceSynchronize());
  float tot = 0, mn = 1e30f;
  for (int i = 0; i < reps; ++i) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(G::THREADS), G::LDS_BYTES, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); tot += ms; mn = std::min(mn, ms);
  }
  if (min_ms) *min_ms = mn;
  return tot / reps;
}

template <class G>
static void occupancy(const char* name) {
  int nb = 0;
  auto kern = apply_kernel<G, true, false, false, false>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, G::THREADS, G::LDS_BYTES));
  hipFuncAttributes fa; CK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(kern)));
  printf("occupancy %-10s: %d blocks/CU (LDS %d B dynamic, %d regs, static LDS %zu)\n", name, nb, G::LDS_BYTES, fa.numRegs, fa.sharedSizeBytes);
}

// one ablation arm: stamped build, report wall time, TF-equivalent and the median in-kernel clock
template <int ABL>
static void ablate(const char* name, ApplyArgs a, size_t col_tiles, size_t blocks, double flops, unsigned long long* dS) {
  a.stamps = dS;
  float mn;
  float ms = time_kernel<GeoTall, true, true, ABL>(a, col_tiles, 10, &mn);
  std::vector<unsigned long long> s(8 * blocks);
  CK(hipMemcpy(s.data(), dS, sizeof(unsigned long long) * 8 * blocks, hipMemcpyDeviceToHost));
  std::vector<double> clk, cyc;
  for (size_t b = 0; b < blocks; ++b) {
    const unsigned long long* o = &s[b * 8];
    double us = (o[5] - o[4]) / 100.0;
    if (us > 0) { clk.push_back((double)(o[3] - o[0]) / us / 1e3); cyc.push_back((double)(o[3] - o[1])); }
  }
  std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
  printf("ablate %-34s mean %.4f ms (min %.4f) = %.1f TF-equiv ; clock median %.3f GHz ; WG cycles (first stage -> end) median %.0f max %.0f\n", name, ms, mn,
         flops / ms / 1e9, clk[clk.size() / 2], cyc[cyc.size() / 2], cyc.back());
}

int main(int argc, char** argv) {
  const bool timeline_only = argc > 3 && std::string(argv[3]) == "timeline";
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 256;
  const size_t rods = argc > 2 ? atol(argv[2]) : 65536;
  const size_t M = rods * 4;
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d  n=%u rods=%zu M=%zu\n", prop.gcnArchName, prop.multiProcessorCount, n, rods, M);
  occupancy<GeoTall>("tall");
  occupancy<GeoMid>("mid");
  occupancy<GeoFlat>("flat");

  sri_apply_t* h = nullptr;
  int rc = sri_apply_create(&h, n, SRI_OP_DN, 0);
  if (rc) { fprintf(stderr, "create failed %d\n", rc); return 1; }
  double *dX, *dY; unsigned long long* dS;
  CK(hipMalloc(&dX, sizeof(double) * n * M)); CK(hipMalloc(&dY, sizeof(double) * n * M));
  std::vector<double> hx((size_t)n * M);
  std::mt19937_64 gen(0x5EED); std::uniform_real_distribution<double> U(-1.0, 1.0);
  for (auto& v : hx) v = U(gen);
  CK(hipMemcpy(dX, hx.data(), sizeof(double) * n * M, hipMemcpyHostToDevice));

  ApplyArgs a{};
  a.apack = h->d_apack; a.x = dX; a.y = dY; a.w = nullptr; a.xb = nullptr;
  a.nt_stores = 1; a.ldx = n; a.ldy = n; a.m = M; a.rows = n; a.rows_out = n; a.nk = h->nk; a.row_blocks = h->row_blocks; a.stamps = nullptr;
  if (h->geo != GEO_TALL) { printf("diag is written for the tall geometry (rows > 128)\n"); return 0; }
  const size_t col_tiles = (M + GeoTall::BN - 1) / GeoTall::BN;
  int tall_per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&tall_per_cu, apply_kernel<GeoTall, true, false, false, false>, GeoTall::THREADS, GeoTall::LDS_BYTES));
  size_t blocks = std::min<size_t>(col_tiles * a.row_blocks, (size_t)tall_per_cu * prop.multiProcessorCount);   // persistent grid
  if (const char* cap = getenv("DIAG_BLOCKS")) {        // DIAG_BLOCKS=32: 32 workgroups on 8 tiles each (the panel shrinks with the grid)
    blocks = std::min<size_t>(blocks, (size_t)atol(cap));
  }
  const double flops = 2.0 * n * (double)n * M;

  float mn;
  float ms = time_kernel<GeoTall, true, false>(a, col_tiles, 20, &mn);
  printf("random X : mean %.4f ms (min %.4f)  %.2f TF  (%.1f%% of 78.6)\n", ms, mn, flops / ms / 1e9, flops / ms / 1e9 / 78.6 * 100);
  CK(hipMemset(dX, 0, sizeof(double) * n * M));
  ms = time_kernel<GeoTall, true, false>(a, col_tiles, 20, &mn);
  printf("zero   X : mean %.4f ms (min %.4f)  %.2f TF  (%.1f%% of 78.6)\n", ms, mn, flops / ms / 1e9, flops / ms / 1e9 / 78.6 * 100);
  CK(hipMemcpy(dX, hx.data(), sizeof(double) * n * M, hipMemcpyHostToDevice));

  // STAMP run
  CK(hipMalloc(&dS, sizeof(unsigned long long) * 48 * blocks));
  CK(hipMemset(dS, 0, sizeof(unsigned long long) * 48 * blocks));
  a.stamps = dS;
  // DIAG_ABL selects the ablation of the stamped run whose timeline is printed: 0 = product, 16 = workgroups start (b % 8) / 8
  // of a tile apart, 128 = every tile stores over the first one (no HBM write traffic), 32 / 64 = one group stores nothing
  const int diag_abl = getenv("DIAG_ABL") ? atoi(getenv("DIAG_ABL")) : 0;
  switch (diag_abl) {
    case 16: ms = time_kernel<GeoTall, true, true, 16>(a, col_tiles, 5, &mn); break;
    case 512: ms = time_kernel<GeoTall, true, true, 512>(a, col_tiles, 5, &mn); break;
    case 3: ms = time_kernel<GeoTall, true, true, 3>(a, col_tiles, 5, &mn); break;
    case 1: ms = time_kernel<GeoTall, true, true, 1>(a, col_tiles, 5, &mn); break;
    case 2: ms = time_kernel<GeoTall, true, true, 2>(a, col_tiles, 5, &mn); break;
    case 128: ms = time_kernel<GeoTall, true, true, 128>(a, col_tiles, 5, &mn); break;
    case 32: ms = time_kernel<GeoTall, true, true, 32>(a, col_tiles, 5, &mn); break;
    case 64: ms = time_kernel<GeoTall, true, true, 64>(a, col_tiles, 5, &mn); break;
    default:
      // DIAG_TAIL=1: the odd-row-count (TAIL) instantiation on the same operator (tail_k = 0: its fix-up zeroes k slots of the
      // last chunk — results are wrong, every address stays inside the stage's LDS image) — where does ITS extra time sit?
      if (getenv("DIAG_TAIL") && atoi(getenv("DIAG_TAIL")) != 0) {
        ms = time_kernel<GeoTall, true, false, 0, true>(a, col_tiles, 20, &mn);
        printf("TAIL instantiation, unstamped : mean %.4f ms (min %.4f)\n", ms, mn);
        ms = time_kernel<GeoTall, true, true, 0, true>(a, col_tiles, 5, &mn);
      } else {
        ms = time_kernel<GeoTall, true, true>(a, col_tiles, 5, &mn);
      }
      break;
  }
  printf("stamped (ablation %d) : mean %.4f ms (min %.4f)\n", diag_abl, ms, mn);
  std::vector<unsigned long long> s(8 * blocks);
  CK(hipMemcpy(s.data(), dS, sizeof(unsigned long long) * 8 * blocks, hipMemcpyDeviceToHost));
  // real-time base
  unsigned long long r_min = ~0ull, r_max = 0;
  for (size_t b = 0; b < blocks; ++b) { r_min = std::min(r_min, s[b * 8 + 4]); r_max = std::max(r_max, s[b * 8 + 5]); }
  printf("kernel span by s_memrealtime: %.1f us (100 MHz ticks)\n", (r_max - r_min) / 100.0);
  std::vector<double> clk, dur_us, pro, loop, epi;
  for (size_t b = 0; b < blocks; ++b) {
    const unsigned long long* o = &s[b * 8];
    double cyc = (double)(o[3] - o[0]); double us = (o[5] - o[4]) / 100.0;
    if (us > 0) clk.push_back(cyc / us / 1e3);   // GHz
    dur_us.push_back(us);
    pro.push_back((double)(o[1] - o[0])); loop.push_back((double)(o[2] - o[1])); epi.push_back((double)(o[3] - o[2]));
  }
  auto med = [](std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
  auto pct = [](std::vector<double> v, double q) { std::sort(v.begin(), v.end()); return v[(size_t)(q * (v.size() - 1))]; };
  printf("per-WG: clock median %.3f GHz (p5 %.3f, p95 %.3f) ; duration median %.1f us (p5 %.1f p95 %.1f)\n",
         med(clk), pct(clk, .05), pct(clk, .95), med(dur_us), pct(dur_us, .05), pct(dur_us, .95));
  printf("per-WG cycles: start->first stage median %.0f (p95 %.0f) ; first stage->last compute median %.0f (p5 %.0f p95 %.0f) ; last epilogue+drain median %.0f (p95 %.0f)\n",
         med(pro), pct(pro, .95), med(loop), pct(loop, .05), pct(loop, .95), med(epi), pct(epi, .95));
  printf("ideal = tiles/WG x %u chunks x 64 MFMA x 64 cyc x %d waves/SIMD = %.0f cycles\n", a.nk, GeoTall::NW / 4,
         (double)(col_tiles * a.row_blocks) / blocks * a.nk * 64 * 64 * (GeoTall::NW / 4));
  // timeline: how many WGs are resident over time, in 16 buckets; and distinct (xcc, cu) count
  const int NB = 24; std::vector<int> act(NB, 0);
  for (size_t b = 0; b < blocks; ++b) {
    double t0 = (double)(s[b * 8 + 4] - r_min) / (double)(r_max - r_min), t1 = (double)(s[b * 8 + 5] - r_min) / (double)(r_max - r_min);
    for (int k = 0; k < NB; ++k) { double c = (k + 0.5) / NB; if (c >= t0 && c < t1) act[k]++; }
  }
  printf("resident WGs over the kernel (24 buckets):"); for (int k = 0; k < NB; ++k) printf(" %d", act[k]); printf("\n");
  // start-time histogram of first 512 and round structure
  std::vector<double> starts; for (size_t b = 0; b < blocks; ++b) starts.push_back((s[b * 8 + 4] - r_min) / 100.0);
  std::sort(starts.begin(), starts.end());
  printf("WG start times (us): #0 %.1f  #255 %.1f  #511 %.1f  #512 %.1f  #1023 %.1f  #1024 %.1f  #2048 %.1f  last %.1f\n",
         starts[0], starts[std::min<size_t>(255, blocks - 1)], starts[std::min<size_t>(511, blocks - 1)], starts[std::min<size_t>(512, blocks - 1)],
         starts[std::min<size_t>(1023, blocks - 1)], starts[std::min<size_t>(1024, blocks - 1)], starts[std::min<size_t>(2048, blocks - 1)], starts.back());
  // ---- per-segment cycle shares of one early and one late wave per workgroup ------------------------
  {
    std::vector<unsigned long long> sg(24 * blocks);
    CK(hipMemcpy(sg.data(), dS, sizeof(unsigned long long) * 24 * blocks, hipMemcpyDeviceToHost));
    const char* names_e[5] = {"wait+barrier", "k pair 0 (+ read pair 1, DMA pieces)", "k pair 1", "bookkeeping", "tile end (stores, zero)"};
    const char* names_l[5] = {"group 0 + wait+barrier", "rest of prev k pair 1 (+ DMA pieces)", "k pair 0 (+ read pair 1)", "bookkeeping", "tile end (stores, zero)"};
    for (int grp = 0; grp < 2; ++grp) {
      double tot[6] = {0, 0, 0, 0, 0, 0}, stages = 0;
      for (size_t b = 0; b < blocks; ++b) {
        const unsigned long long* q = &sg[8 * blocks + (b * 2 + grp) * 8];
        for (int k = 0; k < 6; ++k) tot[k] += (double)q[k];
        stages += (double)q[6];
      }
      double all = 0; for (int k = 0; k < 5; ++k) all += tot[k];
      printf("%s wave, cycles per stage (stamped build; ideal 2 waves/SIMD = %d):", grp ? "late " : "early", 64 * 64 * 2);
      for (int k = 0; k < 5; ++k) printf("  %s %.0f", grp ? names_l[k] : names_e[k], tot[k] / stages);
      printf("  | total %.0f\n", all / stages);
    }
  }

  // ---- timeline of one tile end (third tile of every workgroup), cycles relative to the early wave leaving barrier(S) ----
  {
    std::vector<unsigned long long> tlv(48 * blocks);
    CK(hipMemcpy(tlv.data(), dS, sizeof(unsigned long long) * 48 * blocks, hipMemcpyDeviceToHost));
    const char* ne[12] = {"S: body top", "S: barrier(S) left", "S: MFMAs done", "S: stores issued+zeroed", "-", "-",
                          "S+1: body top", "S+1: barrier(S+1) left", "S+1: MFMAs done", "S+1: body end", "-", "-"};
    const char* nl[12] = {"body top", "group 0 done, at barrier(S+1)", "barrier(S+1) left", "rest of (S,1) done", "stores issued+zeroed", "pair (S+1,0) done",
                          "next: body top", "next: at barrier(S+2)", "next: barrier(S+2) left", "next: rest done", "next: (no stores)", "next: body end"};
    for (int grp = 0; grp < 2; ++grp) {
      printf("%s wave, tile-end timeline (median cycles after the early wave left barrier(S); p10..p90):\n", grp ? "late " : "early");
      for (int k = 0; k < 12; ++k) {
        std::vector<double> d;
        for (size_t b = 0; b < blocks; ++b) {
          const unsigned long long* te = &tlv[24 * blocks + (b * 2 + 0) * 12];
          const unsigned long long* tg = &tlv[24 * blocks + (b * 2 + grp) * 12];
          if (te[1] != 0 && tg[k] != 0) d.push_back((double)((long long)(tg[k] - te[1])));
        }
        if (d.empty()) continue;
        printf("   %-34s %8.0f   (%.0f .. %.0f)\n", grp ? nl[k] : ne[k], med(d), pct(d, .1), pct(d, .9));
      }
    }
  }

  // ---- ablations (results wrong by construction; timing + clock only) --------------------------
  if (timeline_only) { sri_apply_destroy(h); return 0; }
  ablate<0>("none (stamped product kernel)", a, col_tiles, blocks, flops, dS);
  ablate<1>("no A re-fetch (L2->LDS)", a, col_tiles, blocks, flops, dS);
  ablate<2>("no X fetch (HBM->LDS)", a, col_tiles, blocks, flops, dS);
  ablate<4>("no Y stores", a, col_tiles, blocks, flops, dS);
  ablate<8>("no LDS fragment reads", a, col_tiles, blocks, flops, dS);
  ablate<3>("no A, no X fetch", a, col_tiles, blocks, flops, dS);
  ablate<7>("no A, no X, no Y", a, col_tiles, blocks, flops, dS);
  ablate<15>("MFMA only (no A,X,Y,LDS reads)", a, col_tiles, blocks, flops, dS);
  ablate<6>("no X, no Y (A re-fetch only)", a, col_tiles, blocks, flops, dS);
  ablate<5>("no A, no Y (X fetch only)", a, col_tiles, blocks, flops, dS);
  ablate<32>("early group stores nothing", a, col_tiles, blocks, flops, dS);
  ablate<64>("late group stores nothing", a, col_tiles, blocks, flops, dS);
  ablate<128>("stores alias one tile (no HBM writes)", a, col_tiles, blocks, flops, dS);
  ablate<256>("accumulators never zeroed", a, col_tiles, blocks, flops, dS);
  ablate<260>("no Y stores, never zeroed", a, col_tiles, blocks, flops, dS);
  ablate<16>("staggered start (b%8)/8 tile", a, col_tiles, blocks, flops, dS);
  ablate<0>("none again", a, col_tiles, blocks, flops, dS);
  ablate<16>("staggered start again", a, col_tiles, blocks, flops, dS);
  sri_apply_destroy(h);
  return 0;
}
