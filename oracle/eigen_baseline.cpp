// eigen_baseline.cpp — TEST INFRASTRUCTURE (bench.py's cpu_baseline leg only): the reference's own library doing the
// reference's own product, main.cpp:172 / :97 — `Y = A * X` with Eigen's dense product, one thread, no BLAS
// (CMakeLists.txt:16-24) — timed on this host.  Built by bench.py ONLY where a real <Eigen/Dense> is installed (never
// against this repo's compat/ stand-ins); this image has none, so the leg reports {"available": false} here.
//   g++ -O3 -march=native -I<eigen3> oracle/eigen_baseline.cpp -o oracle/_ref/eigen_baseline && ./eigen_baseline <n> <rods> <reps>
#include <Eigen/Dense>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char** argv) {
  const int n = argc > 1 ? std::atoi(argv[1]) : 256;
  const int rods = argc > 2 ? std::atoi(argv[2]) : 1024;
  const int reps = argc > 3 ? std::atoi(argv[3]) : 5;
  Eigen::setNbThreads(1);
  std::mt19937_64 gen(0x5EED);
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  Eigen::MatrixXd A(n, n), X(n, 4 * rods), Y(n, 4 * rods);
  for (Eigen::Index i = 0; i < A.size(); ++i) A.data()[i] = U(gen);      // timing is data-independent
  for (Eigen::Index i = 0; i < X.size(); ++i) X.data()[i] = U(gen);
  Y.noalias() = A * X;                                                   // warm
  double best = 1e300, sink = 0.0;
  for (int r = 0; r < reps; ++r) {
    const auto t0 = std::chrono::steady_clock::now();
    Y.noalias() = A * X;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    best = dt < best ? dt : best;
    sink += Y(r % n, r % (4 * rods));
  }
  std::printf("{\"rods\": %d, \"n\": %d, \"reps\": %d, \"seconds_min\": %.9g, \"rod_states_per_s\": %.9g, \"eigen_version\": \"%d.%d.%d\", \"sink\": %.3g}\n",
              rods, n, reps, best, rods / best, EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION, sink);
  return 0;
}
