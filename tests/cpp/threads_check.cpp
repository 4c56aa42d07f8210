// Two host threads drive the C ABI at once: handle creation on two different device ordinals (and on the same one),
// the host formers, and the per-thread error bookkeeping.  Built against a ThreadSanitizer build of the library
// (tests/test_threads.py); any data race makes TSan exit non-zero.  Without a GPU every create returns SRI_ENODEVICE;
// with GPUs it succeeds and the kernels' residency table is filled under its lock from both threads.
#include <cstdio>
#include <thread>
#include <vector>

#include "sri_capi.h"

static int worker(int device, int rounds, int* creates_ok, int* creates_nodev) {
  std::vector<double> D(64 * 64), x(64);
  for (int r = 0; r < rounds; ++r) {
    if (sri_cheb_points(64, 1.0, x.data()) != SRI_OK) return 1;
    if (sri_cheb_dn(64, D.data(), 64) != SRI_OK) return 2;
    for (int which = 0; which < 3; ++which) {
      sri_apply_t* h = nullptr;
      const int rc = sri_apply_create(&h, (r % 2) ? 256 : 64, which, device);
      if (rc == SRI_OK) {
        ++*creates_ok;
        uint32_t rows = 0;
        if (sri_apply_dim(h, &rows) != SRI_OK || rows == 0) return 3;
        sri_apply_destroy(h);
      } else if (rc == SRI_ENODEVICE) {
        ++*creates_nodev;
        if (h != nullptr) return 4;
      } else {
        std::fprintf(stderr, "sri_apply_create: %s (hip %d)\n", sri_strerror(rc), sri_last_hip_error());
        return 5;
      }
    }
    sri_rod_t* rod = nullptr;
    const int rc = sri_rod_create(&rod, 16, device);
    if (rc == SRI_OK) sri_rod_destroy(rod);
    else if (rc != SRI_ENODEVICE) return 6;
  }
  return 0;
}

int main(int argc, char** argv) {
  const int same_device = argc > 1 && argv[1][0] == 's';
  int count = 0;
  sri_device_count(&count);
  int rc[2] = {-1, -1}, ok[2] = {0, 0}, nodev[2] = {0, 0};
  std::thread t0([&] { rc[0] = worker(0, 6, &ok[0], &nodev[0]); });
  std::thread t1([&] { rc[1] = worker(same_device ? 0 : 1, 6, &ok[1], &nodev[1]); });
  t0.join();
  t1.join();
  std::printf("devices %d rc %d %d created %d %d nodevice %d %d\n", count, rc[0], rc[1], ok[0], ok[1], nodev[0], nodev[1]);
  return (rc[0] == 0 && rc[1] == 0) ? 0 : 1;
}
