/*
 * chebyshev_differentiation.h — drop-in replacement for the reference header of the same name
 * (/root/reference/include/chebyshev_differentiation.h), backed by libsri_hip.so.
 *
 * It exposes the reference's three function templates with the same names, template parameters,
 * return types and call syntax, so that main.cpp (main.cpp:15, 59, 93, 157) compiles against it
 * unchanged:
 *
 *     ComputeChebyshevPoints<N, L = 1>()  -> std::array<double, N>     (reference h:19-30)
 *     GetCoefficients_c<N>()              -> std::array<double, N>     (reference h:37-52)
 *     getDn<N>()                          -> Eigen::MatrixXd (N x N)   (reference h:59-108)
 *
 * Differences from the reference header, all deliberate:
 *   - the arithmetic lives in the library (sri_cheb_points / sri_cheb_coeffs / sri_cheb_dn of
 *     include/sri_capi.h), not in the header; values are bit-identical to the reference's formulas;
 *   - no fixed-size N x N temporaries, so N is not limited by Eigen's stack-allocation limit
 *     (the reference's getDn stops compiling near N = 128);
 *   - it includes the standard headers it needs itself, and <iostream> like the reference header (h:11) because
 *     main.cpp:198,202 use std::cout without an include of their own;
 *   - when <Eigen/Dense> is not installed, getDn returns sri::DenseMatrix, a minimal column-major
 *     owner with the accessors main.cpp-style code needs ((i,j), rows(), cols(), data());
 *   - sri::SpectralApply is a small RAII wrapper of the device apply (main.cpp:172 batched).
 */
#ifndef CHEBYSHEV_DIFFERENTIATION_H
#define CHEBYSHEV_DIFFERENTIATION_H

// <iostream> is what the reference header itself includes (reference h:11) and main.cpp relies on it transitively
// (std::cout at main.cpp:198,202); <array>, <algorithm>, <cmath> are the headers the reference header uses without
// including them (std::array h:20, std::generate h:25, cos / pow / M_PI h:26,48).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstddef>
#include <iostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "sri_capi.h"

#if defined(__has_include)
#if __has_include(<Eigen/Dense>)
#include <Eigen/Dense>
#define SRI_HAVE_EIGEN 1
#endif
#endif

namespace sri {

inline void check(int status, const char* what) {
  if (status != SRI_OK) throw std::runtime_error(std::string(what) + ": " + sri_strerror(status));
}

#ifndef SRI_HAVE_EIGEN
// Stand-in for Eigen::MatrixXd when Eigen is not installed: dynamic, column-major, owning.
class DenseMatrix {
 public:
  DenseMatrix() = default;
  DenseMatrix(std::size_t rows, std::size_t cols) : rows_(rows), cols_(cols), v_(rows * cols, 0.0) {}
  double& operator()(std::size_t i, std::size_t j) { return v_[i + j * rows_]; }
  double operator()(std::size_t i, std::size_t j) const { return v_[i + j * rows_]; }
  std::size_t rows() const { return rows_; }
  std::size_t cols() const { return cols_; }
  std::size_t outerStride() const { return rows_; }
  double* data() { return v_.data(); }
  const double* data() const { return v_.data(); }
  // Dn.block(i0, j0, r, c) of main.cpp:94-95 as a copy
  DenseMatrix block(std::size_t i0, std::size_t j0, std::size_t r, std::size_t c) const {
    DenseMatrix b(r, c);
    for (std::size_t j = 0; j < c; ++j)
      for (std::size_t i = 0; i < r; ++i) b(i, j) = (*this)(i0 + i, j0 + j);
    return b;
  }

 private:
  std::size_t rows_ = 0, cols_ = 0;
  std::vector<double> v_;
};
using MatrixXd = DenseMatrix;
#else
using MatrixXd = Eigen::MatrixXd;
#endif

// Device-resident operator, Y = A X for column-major state panels (main.cpp:172 / main.cpp:97 batched).
class SpectralApply {
 public:
  SpectralApply(unsigned n, sri_operator which, int device = 0) { check(sri_apply_create(&h_, n, which, device), "sri_apply_create"); }
  ~SpectralApply() { sri_apply_destroy(h_); }
  SpectralApply(const SpectralApply&) = delete;
  SpectralApply& operator=(const SpectralApply&) = delete;
  unsigned rows() const { uint32_t r = 0; sri_apply_dim(h_, &r); return r; }
  // device pointers; enqueues on `stream` (hipStream_t) and returns
  void operator()(const double* dX, std::size_t ldx, double* dY, std::size_t ldy, std::size_t M, void* stream = nullptr) {
    check(sri_apply(h_, dX, ldx, dY, ldy, M, stream), "sri_apply");
  }
  void integrate(const double* dB, std::size_t ldb, const double* d_x_base, double* dY, std::size_t ldy, std::size_t M,
                 void* stream = nullptr) {
    check(sri_integrate(h_, dB, ldb, d_x_base, dY, ldy, M, stream), "sri_integrate");
  }
  // host pointers (staged over PCIe, synchronous)
  void host(const double* X, std::size_t ldx, double* Y, std::size_t ldy, std::size_t M) {
    check(sri_apply_host(h_, X, ldx, Y, ldy, M), "sri_apply_host");
  }
  sri_apply_t* handle() { return h_; }

 private:
  sri_apply_t* h_ = nullptr;
};

}  // namespace sri

/* The Chebyshev points x_j = (L/2)(1 + cos(pi j/(N-1))) on [0, L], descending (tip first). */
template <unsigned int t_number_of_chebyshev_nodes, unsigned int t_L = 1>
static std::array<double, t_number_of_chebyshev_nodes> ComputeChebyshevPoints() {
  std::array<double, t_number_of_chebyshev_nodes> x{};
  sri::check(sri_cheb_points(t_number_of_chebyshev_nodes, static_cast<double>(t_L), x.data()), "sri_cheb_points");
  return x;
}

/* The c_i = (-1)^i (2 at both ends, 1 elsewhere) of the differentiation-matrix formula. */
template <unsigned int t_number_of_chebyshev_nodes>
static std::array<double, t_number_of_chebyshev_nodes> GetCoefficients_c() {
  std::array<double, t_number_of_chebyshev_nodes> c{};
  sri::check(sri_cheb_coeffs(t_number_of_chebyshev_nodes, c.data()), "sri_cheb_coeffs");
  return c;
}

/* The N x N Chebyshev differentiation matrix on [0, 1]. */
template <unsigned int t_number_of_chebyshev_nodes>
static sri::MatrixXd getDn() {
  sri::MatrixXd Dn(t_number_of_chebyshev_nodes, t_number_of_chebyshev_nodes);
  sri::check(sri_cheb_dn(t_number_of_chebyshev_nodes, Dn.data(), static_cast<std::size_t>(Dn.outerStride())), "sri_cheb_dn");
  return Dn;
}

#endif  // CHEBYSHEV_DIFFERENTIATION_H
