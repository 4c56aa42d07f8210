// rod_main.cpp — BASELINE config 1 end to end: the reference's driver (main.cpp) restated on raw
// column-major arrays on top of THIS repo's drop-in header and library.
//
//   integrateQuaternions()  main.cpp:91-118  per-rod 4(N-1) x 4(N-1) system (application code, host)
//   updatePositionb()       main.cpp:121-140 quaternion -> R(q) e1            (application code, host)
//   integratePosition()     main.cpp:145-176 r_stack = Dn_NN^-1 (b_NN - ivp)  <- THE hot-path product,
//                                             executed on the GPU through sri::SpectralApply
//
// The strain coordinates are the hard-coded qe of main.cpp:187-195.  With `rods > 1` the same rod is
// replicated with a per-rod scaling of qe so that the batched device path is exercised.
// Output mirrors main.cpp:198,202: "Q_stack : \n" << Q_stack << std::endl with Eigen's default IOFormat
// (stream precision 6, every coefficient right-aligned to the widest one, columns separated by one space);
// `--full` prints 17 significant digits, one value / row per line, for tests.
//
// Build (see tests/test_examples_gpu.py):
//   g++ -std=c++17 -O2 -Iinclude examples/rod_main.cpp -L<pkg> -lsri_hip -Wl,-rpath,<pkg> -o build/rod_main
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>

#include "chebyshev_differentiation.h"

static constexpr unsigned int number_of_Chebyshev_points = 16;   // main.cpp:4
static constexpr unsigned int state_dimension = 4;               // main.cpp:6
static constexpr unsigned int ne = 3, na = 3;                    // main.cpp:12-13
static constexpr unsigned int M1 = number_of_Chebyshev_points - 1;

// utilities.h:49-67 — Legendre modes P_0..P_{ne-1} at x = 2X - 1; K = (I_na (x) P^T) qe (main.cpp:69)
static void strain(double X, const double* qe, double K[3]) {
  const double x = (2 * X - 1.0) / 1.0;
  double P[ne] = {1.0, x, 0.0};
  for (unsigned l = 1; l + 1 < ne; ++l) P[l + 1] = ((2 * l + 1) * x * P[l] - l * P[l - 1]) / (l + 1);
  for (unsigned a = 0; a < na; ++a) {
    double s = 0.0;
    for (unsigned e = 0; e < ne; ++e) s += P[e] * qe[a * ne + e];
    K[a] = s;
  }
}

// dense solve by LU with partial pivoting (what A_NN.inverse() * rhs amounts to, main.cpp:113)
static void solve(std::vector<double>& A, std::vector<double>& b, unsigned n) {
  for (unsigned k = 0; k < n; ++k) {
    unsigned p = k;
    for (unsigned i = k + 1; i < n; ++i)
      if (std::fabs(A[i + k * n]) > std::fabs(A[p + k * n])) p = i;
    if (p != k) {
      for (unsigned j = 0; j < n; ++j) std::swap(A[k + j * n], A[p + j * n]);
      std::swap(b[k], b[p]);
    }
    for (unsigned i = k + 1; i < n; ++i) {
      const double f = A[i + k * n] / A[k + k * n];
      if (f == 0.0) continue;
      for (unsigned j = k; j < n; ++j) A[i + j * n] -= f * A[k + j * n];
      b[i] -= f * b[k];
    }
  }
  for (unsigned k = n; k-- > 0;) {
    for (unsigned j = k + 1; j < n; ++j) b[k] -= A[k + j * n] * b[j];
    b[k] /= A[k + k * n];
  }
}

// main.cpp:91-118 with updateA (main.cpp:55-88): returns Q_stack = [w(0..m-1); x; y; z]
static std::vector<double> integrateQuaternions(const double* qe) {
  const auto x = ComputeChebyshevPoints<number_of_Chebyshev_points>();
  const sri::MatrixXd Dn = getDn<number_of_Chebyshev_points>();
  const unsigned dim = state_dimension * M1;
  std::vector<double> A(static_cast<size_t>(dim) * dim, 0.0), rhs(dim, 0.0);
  for (unsigned c = 0; c < state_dimension; ++c)            // D_NN = I_4 (x) Dn_NN
    for (unsigned j = 0; j < M1; ++j)
      for (unsigned i = 0; i < M1; ++i) A[(c * M1 + i) + static_cast<size_t>(c * M1 + j) * dim] = Dn(i, j);
  for (unsigned i = 0; i < M1; ++i) {
    double K[3];
    strain(x[i], qe, K);
    const double Ak[4][4] = {{0, -K[0], -K[1], -K[2]}, {K[0], 0, K[2], -K[1]}, {K[1], -K[2], 0, K[0]}, {K[2], K[1], -K[0], 0}};
    for (unsigned r = 0; r < 4; ++r)
      for (unsigned c = 0; c < 4; ++c)
        A[(r * M1 + i) + static_cast<size_t>(c * M1 + i) * dim] = (r == c ? Dn(i, i) : 0.0) - 0.5 * Ak[r][c];
  }
  const double q_init[4] = {1, 0, 0, 0};                     // main.cpp:106-107
  for (unsigned c = 0; c < state_dimension; ++c)
    for (unsigned i = 0; i < M1; ++i) rhs[c * M1 + i] = 0.0 - Dn(i, M1) * q_init[c];   // b - ivp, ivp = D_IN q_init
  solve(A, rhs, dim);
  return rhs;
}

// main.cpp:121-140
static void updatePositionb(const std::vector<double>& Q, double* b /* M1 x 3 column-major */) {
  for (unsigned i = 0; i < M1; ++i) {
    const double w = Q[i], x = Q[i + M1], y = Q[i + 2 * M1], z = Q[i + 3 * M1];
    const double ty = 2 * y, tz = 2 * z;
    b[i] = 1 - (ty * y + tz * z);
    b[i + M1] = ty * x + tz * w;
    b[i + 2 * M1] = tz * x - ty * w;
  }
}

// What `std::cout << M` prints for an Eigen matrix with the default IOFormat: "%.6g"-style coefficients,
// all padded on the left to the width of the widest, " " between columns, "\n" between rows.
static void print_like_eigen(const double* a, unsigned rows, unsigned cols, unsigned ld) {
  std::vector<std::string> txt(static_cast<size_t>(rows) * cols);
  size_t width = 0;
  char buf[64];
  for (unsigned j = 0; j < cols; ++j)
    for (unsigned i = 0; i < rows; ++i) {
      std::snprintf(buf, sizeof buf, "%.6g", a[i + static_cast<size_t>(j) * ld]);
      txt[i + static_cast<size_t>(j) * rows] = buf;
      width = std::max(width, txt[i + static_cast<size_t>(j) * rows].size());
    }
  for (unsigned i = 0; i < rows; ++i) {
    for (unsigned j = 0; j < cols; ++j) {
      if (j) std::printf(" ");
      std::printf("%*s", static_cast<int>(width), txt[i + static_cast<size_t>(j) * rows].c_str());
    }
    if (i + 1 < rows) std::printf("\n");
  }
  std::printf("\n");                                    // std::endl of main.cpp:198,202
}

int main(int argc, char* argv[]) {
  unsigned rods = 1;
  bool full = false;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--full")) full = true;
    else rods = static_cast<unsigned>(std::atoi(argv[i]));
  }
  if (rods < 1) rods = 1;
  const double qe0[9] = {0, 0, 0, 1.2877691307032, -1.63807499160786, 0.437406679142598, 0, 0, 0};   // main.cpp:187-195

  try {
    // ---- quaternions: per rod, on the host (application code) ------------------------------------
    std::vector<std::vector<double>> Q(rods);
    for (unsigned r = 0; r < rods; ++r) {
      double qe[9];
      for (int k = 0; k < 9; ++k) qe[k] = qe0[k] * (1.0 - 0.5 * r / rods);
      Q[r] = integrateQuaternions(qe);
    }
    std::printf("Q_stack : \n");
    if (full) for (unsigned k = 0; k < state_dimension * M1; ++k) std::printf("%.17g\n", Q[0][k]);
    else print_like_eigen(Q[0].data(), state_dimension * M1, 1, state_dimension * M1);

    // ---- positions: one batched device product for all rods (the hot path) ------------------------
    const size_t cols = static_cast<size_t>(rods) * 3;
    std::vector<double> B(M1 * cols), R(M1 * cols), r_init(cols, 0.0);    // r_init = 0 (main.cpp:151-154)
    for (unsigned r = 0; r < rods; ++r) updatePositionb(Q[r], &B[static_cast<size_t>(r) * 3 * M1]);
    sri::SpectralApply integrate(number_of_Chebyshev_points, SRI_OP_DNN_INV, 0);
    sri::check(sri_integrate_host(integrate.handle(), B.data(), M1, r_init.data(), R.data(), M1, cols), "sri_integrate_host");
    std::printf("r_stack : \n");
    if (full) for (unsigned i = 0; i < M1; ++i) std::printf("%.17g %.17g %.17g\n", R[i], R[i + M1], R[i + 2 * M1]);
    else print_like_eigen(R.data(), M1, 3, M1);
    if (rods > 1) std::printf("(rod 0 of %u shown; kernel %s)\n", rods, sri_apply_last_kernel(integrate.handle()));
  } catch (const std::exception& e) {
    std::fprintf(stderr, "rod_main: %s\n", e.what());
    return 1;
  }
  return 0;
}
