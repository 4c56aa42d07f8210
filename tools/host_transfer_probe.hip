// What the host side of sri_apply_host can count on (diagnostic tool, not product code).
//
// The drop-in header hands HOST panels to the library, so the PCIe-inclusive rate is bounded by how the bytes get
// from pageable memory to the card and back.  Measures, for a 1 GiB panel:
//   * hipMemcpy from / to pageable memory (what the first implementation did),
//   * hipMemcpyAsync from / to pinned memory, one direction and both at once (two streams),
//   * hipHostRegister + hipHostUnregister of the caller's pageable buffer,
//   * pageable -> pinned staging copies on the CPU with 1..16 threads.
// build: make build/host_transfer_probe     run: build/host_transfer_probe [MiB]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CK(x)                                                                    \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void copy_threads(char* dst, const char* src, size_t n, int threads) {
  std::vector<std::thread> pool;
  const size_t per = ((n / threads) + 4095) & ~size_t(4095);
  for (int t = 0; t < threads; ++t) {
    const size_t b = std::min(n, per * t), e = std::min(n, per * (t + 1));
    if (e > b) pool.emplace_back([=] { std::memcpy(dst + b, src + b, e - b); });
  }
  for (auto& th : pool) th.join();
}

int main(int argc, char** argv) {
  const size_t mib = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 1024;
  const size_t n = mib << 20;
  char *pageable = static_cast<char*>(std::malloc(n)), *pageable2 = static_cast<char*>(std::malloc(n));
  if (!pageable || !pageable2) return 2;
  std::memset(pageable, 1, n);
  std::memset(pageable2, 2, n);
  char *pin_a = nullptr, *pin_b = nullptr, *dev_a = nullptr, *dev_b = nullptr;
  CK(hipHostMalloc(reinterpret_cast<void**>(&pin_a), n, hipHostMallocDefault));
  CK(hipHostMalloc(reinterpret_cast<void**>(&pin_b), n, hipHostMallocDefault));
  CK(hipMalloc(reinterpret_cast<void**>(&dev_a), n));
  CK(hipMalloc(reinterpret_cast<void**>(&dev_b), n));
  std::memset(pin_a, 3, n);
  std::memset(pin_b, 4, n);
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  const double gb = double(n) / 1e9;
  std::printf("panel %zu MiB, hardware threads %u\n", mib, std::thread::hardware_concurrency());

  for (int rep = 0; rep < 3; ++rep) {
    double t = now();
    CK(hipMemcpy(dev_a, pageable, n, hipMemcpyHostToDevice));
    const double h2d = now() - t;
    t = now();
    CK(hipMemcpy(pageable2, dev_a, n, hipMemcpyDeviceToHost));
    const double d2h = now() - t;
    std::printf("pageable hipMemcpy      : H2D %6.1f GB/s   D2H %6.1f GB/s\n", gb / h2d, gb / d2h);
  }
  for (int rep = 0; rep < 3; ++rep) {
    double t = now();
    CK(hipMemcpyAsync(dev_a, pin_a, n, hipMemcpyHostToDevice, s0));
    CK(hipStreamSynchronize(s0));
    const double h2d = now() - t;
    t = now();
    CK(hipMemcpyAsync(pin_b, dev_b, n, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s1));
    const double d2h = now() - t;
    t = now();
    CK(hipMemcpyAsync(dev_a, pin_a, n, hipMemcpyHostToDevice, s0));
    CK(hipMemcpyAsync(pin_b, dev_b, n, hipMemcpyDeviceToHost, s1));
    CK(hipStreamSynchronize(s0));
    CK(hipStreamSynchronize(s1));
    const double both = now() - t;
    std::printf("pinned hipMemcpyAsync   : H2D %6.1f GB/s   D2H %6.1f GB/s   both at once %6.1f GB/s each way\n", gb / h2d,
                gb / d2h, gb / both);
  }
  // chunked pinned copies: what a pipeline of 8 / 32 MiB blocks sees
  for (size_t chunk_mib : {4, 8, 16, 32, 64}) {
    const size_t c = chunk_mib << 20;
    double t = now();
    for (size_t o = 0; o + c <= n; o += c) {
      CK(hipMemcpyAsync(dev_a + o, pin_a + o, c, hipMemcpyHostToDevice, s0));
      CK(hipMemcpyAsync(pin_b + o, dev_b + o, c, hipMemcpyDeviceToHost, s1));
    }
    CK(hipStreamSynchronize(s0));
    CK(hipStreamSynchronize(s1));
    std::printf("pinned, %2zu MiB chunks, both ways at once: %6.1f GB/s each way\n", chunk_mib, gb / (now() - t));
  }
  for (int rep = 0; rep < 3; ++rep) {
    double t = now();
    CK(hipHostRegister(pageable, n, hipHostRegisterDefault));
    const double reg = now() - t;
    t = now();
    CK(hipMemcpyAsync(dev_a, pageable, n, hipMemcpyHostToDevice, s0));
    CK(hipStreamSynchronize(s0));
    const double h2d = now() - t;
    t = now();
    CK(hipHostUnregister(pageable));
    const double unreg = now() - t;
    std::printf("hipHostRegister         : register %6.1f ms  unregister %6.1f ms  (H2D from it %6.1f GB/s)\n", reg * 1e3,
                unreg * 1e3, gb / h2d);
  }
  for (int threads : {1, 2, 4, 8, 16}) {
    double best = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      const double t = now();
      copy_threads(pin_a, pageable, n, threads);
      best = std::min(best, now() - t);
    }
    double best2 = 1e9;
    for (int rep = 0; rep < 3; ++rep) {
      const double t = now();
      copy_threads(pageable2, pin_b, n, threads);
      best2 = std::min(best2, now() - t);
    }
    std::printf("CPU staging copy, %2d threads: pageable->pinned %6.1f GB/s   pinned->pageable %6.1f GB/s\n", threads,
                gb / best, gb / best2);
  }
  return 0;
}
