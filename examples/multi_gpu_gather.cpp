// multi_gpu_gather.cpp — the multi-GPU path from C++ only (no Python, no torch): a batch of rod states sharded by
// columns over the GPUs of one node, D_N applied on every shard, and the integrated tip state of all rods reassembled
// on every GPU with ONE RCCL all-gather over xGMI (north_star: "all-gather ... only to reassemble the
// final integrated state").  One host thread per GPU here; one process per GPU works the same way (ship the 128-byte
// token of sri_comm_unique_id() to the other ranks by any side channel).
//
//   g++ -std=c++17 -O2 -Iinclude examples/multi_gpu_gather.cpp -L<pkg> -lsri_hip -Wl,-rpath,<pkg> -L/opt/rocm/lib -lamdhip64 -pthread
//   ./multi_gpu_gather [gpus] [rods]          (gpus defaults to all devices)
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "sri_capi.h"

#define CHECK(call)                                                                                       \
  do {                                                                                                    \
    const int rc__ = (call);                                                                              \
    if (rc__ != SRI_OK) {                                                                                 \
      std::fprintf(stderr, "%s: %s (hip %d, rccl %d)\n", #call, sri_strerror(rc__), sri_last_hip_error(), \
                   sri_last_rccl_error());                                                                \
      std::exit(1);                                                                                       \
    }                                                                                                     \
  } while (0)
#define HIP(call)                                                                  \
  do {                                                                             \
    if ((call) != hipSuccess) {                                                    \
      std::fprintf(stderr, "%s failed at %s:%d\n", #call, __FILE__, __LINE__);     \
      std::exit(1);                                                                \
    }                                                                              \
  } while (0)

static constexpr unsigned N = 64;   // Chebyshev nodes

static void shard(size_t total_rods, int world, int rank, size_t* first, size_t* count) {   // contiguous, balanced
  const size_t base = total_rods / world, extra = total_rods % world;
  *count = base + (static_cast<size_t>(rank) < extra ? 1 : 0);
  *first = rank * base + (static_cast<size_t>(rank) < extra ? rank : extra);
}

int main(int argc, char** argv) {
  int devices = 0;
  CHECK(sri_device_count(&devices));
  const int world = argc > 1 ? std::atoi(argv[1]) : devices;
  const size_t rods = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 1003;   // not a multiple of anything: ragged shards
  if (world < 1 || world > devices) {
    std::fprintf(stderr, "need 1..%d GPUs\n", devices);
    return 1;
  }
  const size_t M = 4 * rods;                                       // 4 state columns per rod (main.cpp:6)
  // the global panel on the host: X(i, j) smooth in the node index, different per column
  std::vector<double> x(N), X(static_cast<size_t>(N) * M), D(static_cast<size_t>(N) * N);
  CHECK(sri_cheb_points(N, 1.0, x.data()));
  CHECK(sri_cheb_dn(N, D.data(), N));
  for (size_t j = 0; j < M; ++j)
    for (unsigned i = 0; i < N; ++i) X[i + j * N] = (1.0 + 1e-3 * static_cast<double>(j % 977)) * x[i] * x[i] - 0.25 * x[i];
  // expected tip state: row 0 of D X, accumulated as the kernel does it is not needed here: compare to 1e-12
  std::vector<double> expect(M, 0.0);
  for (size_t j = 0; j < M; ++j)
    for (unsigned k = 0; k < N; ++k) expect[j] += D[0 + static_cast<size_t>(k) * N] * X[k + j * N];

  char id[SRI_COMM_ID_BYTES];
  CHECK(sri_comm_unique_id(id));
  std::vector<size_t> cols(world);
  for (int r = 0; r < world; ++r) {
    size_t f, c;
    shard(rods, world, r, &f, &c);
    cols[r] = 4 * c;
  }
  std::vector<int> ok(world, 0);
  std::vector<std::thread> ranks;
  for (int r = 0; r < world; ++r)
    ranks.emplace_back([&, r] {
      HIP(hipSetDevice(r));
      size_t first, count;
      shard(rods, world, r, &first, &count);
      const size_t Ml = 4 * count;
      sri_comm_t* comm = nullptr;
      CHECK(sri_comm_create(&comm, id, r, world, r));              // collective over the `world` threads
      sri_apply_t* op = nullptr;
      CHECK(sri_apply_create(&op, N, SRI_OP_DN, r));
      double *dX, *dY, *d_tips, *d_all;
      hipStream_t stream;
      HIP(hipStreamCreate(&stream));
      HIP(hipMalloc(&dX, sizeof(double) * N * (Ml ? Ml : 1)));
      HIP(hipMalloc(&dY, sizeof(double) * N * (Ml ? Ml : 1)));
      HIP(hipMalloc(&d_tips, sizeof(double) * (Ml ? Ml : 1)));
      HIP(hipMalloc(&d_all, sizeof(double) * M));
      HIP(hipMemcpyAsync(dX, X.data() + static_cast<size_t>(N) * 4 * first, sizeof(double) * N * Ml, hipMemcpyHostToDevice, stream));
      if (Ml) CHECK(sri_apply_tips(op, dX, N, dY, N, Ml, d_tips, stream));           // my slab: no communication
      CHECK(sri_allgather_tips(comm, d_tips, Ml, d_all, cols.data(), stream));       // the one exchange step
      std::vector<double> got(M);
      HIP(hipMemcpyAsync(got.data(), d_all, sizeof(double) * M, hipMemcpyDeviceToHost, stream));
      HIP(hipStreamSynchronize(stream));
      double worst = 0.0, scale = 0.0;
      for (size_t j = 0; j < M; ++j) {
        const double e = got[j] - expect[j];
        worst = e * e > worst ? e * e : worst;
        scale = expect[j] * expect[j] > scale ? expect[j] * expect[j] : scale;
      }
      ok[r] = worst <= 1e-24 * scale;
      (void)hipFree(dX); (void)hipFree(dY); (void)hipFree(d_tips); (void)hipFree(d_all);
      (void)hipStreamDestroy(stream);
      sri_apply_destroy(op);
      sri_comm_destroy(comm);
    });
  for (auto& t : ranks) t.join();
  int all = 1;
  for (int r = 0; r < world; ++r) all &= ok[r];
  std::printf("%d GPU(s), %zu rods (%zu columns): gathered tip state %s on every rank\n", world, rods, M, all ? "verified" : "WRONG");
  return all ? 0 : 1;
}
