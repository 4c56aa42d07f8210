// sri_host.cpp — host-side formers of the spectral operators (product code, C++).
//
// The reference forms D_N on the CPU in O(N^2) (chebyshev_differentiation.h:59-108) and inverts
// the (N-1)x(N-1) block once (main.cpp:159).  These are setup steps of the hot path, executed
// once per N; they stay on the host on purpose: a 1-ulp change in a Chebyshev node moves
// off-diagonal entries of D_256 by 5.9e-12, so the nodes must come from
// the same libm `cos` expression the reference uses, not from a device approximation.
// The O(N^2 M) apply — the actual hot loop — is in sri_apply.hip.
//
// Build with -ffp-contract=off: the reference sets no -march flag (CMakeLists.txt:8-9), so its
// arithmetic is plain multiply-then-add.
#include "sri_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

namespace sri {

// The library's one environment hook (sri_internal.h): the value of `key` in SRI_DEBUG="key=value,key=value".
std::string debug_option(const char* key) {
  const char* env = std::getenv("SRI_DEBUG");
  if (env == nullptr || key == nullptr) return std::string();
  const size_t klen = std::strlen(key);
  for (const char* p = env; *p != '\0';) {
    const char* end = std::strchr(p, ',');
    const size_t len = end != nullptr ? static_cast<size_t>(end - p) : std::strlen(p);
    if (len > klen && std::strncmp(p, key, klen) == 0 && p[klen] == '=') return std::string(p + klen + 1, len - klen - 1);
    p += len + (end != nullptr ? 1 : 0);
  }
  return std::string();
}

// chebyshev_differentiation.h:26
//   (static_cast<double>(t_L)/2)*(1 + cos(M_PI*static_cast<double>(j++)/static_cast<double>(N-1)))
int cheb_points(uint32_t n, double L, double* x) {
  if (n < 2 || x == nullptr) return SRI_EINVAL;
  const double half_len = L / 2;
  const double denom = static_cast<double>(n - 1);
  for (uint32_t j = 0; j < n; ++j) {
    const double angle = M_PI * static_cast<double>(j) / denom;
    x[j] = half_len * (1 + std::cos(angle));
  }
  return SRI_OK;
}

// chebyshev_differentiation.h:45,48: gain 2 at the two ends, sign alternates; pow(-1, i)*gain.
int cheb_coeffs(uint32_t n, double* c) {
  if (n < 2 || c == nullptr) return SRI_EINVAL;
  for (uint32_t i = 0; i < n; ++i) {
    const bool edge = (i == 0) || (i == n - 1);
    c[i] = std::pow(-1.0, static_cast<double>(i)) * (edge ? 2.0 : 1.0);
  }
  return SRI_OK;
}

// chebyshev_differentiation.h:59-108.  Entry (i,j) = (c_i/c_j) / ((x_i - x_j) + [i==j]); then the
// diagonal becomes 1 - (left-to-right sum of the whole row, provisional diagonal 1 included).
int cheb_dn(uint32_t n, double* D, size_t ld) {
  if (n < 2 || D == nullptr || ld < n) return SRI_EINVAL;
  std::vector<double> x, c;
  try { x.resize(n); c.resize(n); } catch (const std::bad_alloc&) { return SRI_ENOMEM; }
  cheb_points(n, 1.0, x.data());
  cheb_coeffs(n, c.data());
  for (uint32_t j = 0; j < n; ++j) {
    double* col = D + j * ld;
    for (uint32_t i = 0; i < n; ++i) {
      const double ratio = c[i] / c[j];                               // h:84
      const double gap = (x[i] - x[j]) + (i == j ? 1.0 : 0.0);        // h:89
      col[i] = ratio / gap;                                           // h:98
    }
  }
  for (uint32_t i = 0; i < n; ++i) {                                  // h:104
    double row_sum = D[i];
    for (uint32_t j = 1; j < n; ++j) row_sum = row_sum + D[i + j * ld];
    D[i + i * ld] -= row_sum;
  }
  return SRI_OK;
}

// main.cpp:94-95, 158, 160
int cheb_blocks(uint32_t n, const double* D, size_t ld, double* Dnn, size_t ldnn, double* Din) {
  if (n < 2 || D == nullptr || ld < n) return SRI_EINVAL;
  const uint32_t m = n - 1;
  if (Dnn != nullptr) {
    if (ldnn < m) return SRI_EINVAL;
    for (uint32_t j = 0; j < m; ++j) std::memcpy(Dnn + j * ldnn, D + j * ld, sizeof(double) * m);
  }
  if (Din != nullptr) std::memcpy(Din, D + static_cast<size_t>(m) * ld, sizeof(double) * m);
  return SRI_OK;
}

// Dense inverse by partial-pivot LU, as Eigen's PartialPivLU-based .inverse() (main.cpp:159):
// first-largest pivot per column, sub-column scaled by 1/pivot via division, rank-1 trailing
// update, then L and U column-oriented solves against the row-permuted identity.
int dense_inverse(uint32_t m, const double* A, size_t lda, double* Ainv, size_t ldi) {
  if (m < 1 || A == nullptr || Ainv == nullptr || lda < m || ldi < m) return SRI_EINVAL;
  std::vector<double> lu;
  std::vector<uint32_t> swaps;
  try { lu.resize(static_cast<size_t>(m) * m); swaps.resize(m); } catch (const std::bad_alloc&) { return SRI_ENOMEM; }
  for (uint32_t j = 0; j < m; ++j) std::memcpy(&lu[static_cast<size_t>(j) * m], A + j * lda, sizeof(double) * m);
  auto at = [&](uint32_t i, uint32_t j) -> double& { return lu[i + static_cast<size_t>(j) * m]; };

  for (uint32_t k = 0; k < m; ++k) {
    uint32_t best = k;
    double best_abs = std::fabs(at(k, k));
    for (uint32_t i = k + 1; i < m; ++i) {
      const double v = std::fabs(at(i, k));
      if (v > best_abs) { best_abs = v; best = i; }
    }
    swaps[k] = best;
    if (best_abs == 0.0) return SRI_ESINGULAR;
    if (best != k)
      for (uint32_t j = 0; j < m; ++j) std::swap(at(k, j), at(best, j));
    const double pivot = at(k, k);
    for (uint32_t i = k + 1; i < m; ++i) at(i, k) /= pivot;
    for (uint32_t j = k + 1; j < m; ++j) {
      const double u = at(k, j);
      double* colj = &at(0, j);
      const double* colk = &at(0, k);
      for (uint32_t i = k + 1; i < m; ++i) colj[i] -= colk[i] * u;
    }
  }

  for (uint32_t j = 0; j < m; ++j) {
    double* col = Ainv + j * ldi;
    for (uint32_t i = 0; i < m; ++i) col[i] = (i == j) ? 1.0 : 0.0;
  }
  for (uint32_t k = 0; k < m; ++k)
    if (swaps[k] != k)
      for (uint32_t j = 0; j < m; ++j) std::swap(Ainv[k + j * ldi], Ainv[swaps[k] + j * ldi]);
  for (uint32_t j = 0; j < m; ++j) {
    double* b = Ainv + j * ldi;
    for (uint32_t k = 0; k < m; ++k) {
      const double bk = b[k];
      if (bk == 0.0) continue;
      const double* colk = &at(0, k);
      for (uint32_t i = k + 1; i < m; ++i) b[i] -= bk * colk[i];
    }
    for (uint32_t k = m; k-- > 0;) {
      b[k] /= at(k, k);
      const double bk = b[k];
      const double* colk = &at(0, k);
      for (uint32_t i = 0; i < k; ++i) b[i] -= bk * colk[i];
    }
  }
  return SRI_OK;
}

// main.cpp:157-160
int cheb_integration(uint32_t n, double* Dnn_inv, size_t ld, double* Din) {
  if (n < 2 || Dnn_inv == nullptr || ld < n - 1) return SRI_EINVAL;
  const uint32_t m = n - 1;
  std::vector<double> D, Dnn;
  try { D.resize(static_cast<size_t>(n) * n); Dnn.resize(static_cast<size_t>(m) * m); } catch (const std::bad_alloc&) { return SRI_ENOMEM; }
  int rc = cheb_dn(n, D.data(), n);
  if (rc != SRI_OK) return rc;
  rc = cheb_blocks(n, D.data(), n, Dnn.data(), m, Din);
  if (rc != SRI_OK) return rc;
  return dense_inverse(m, Dnn.data(), m, Dnn_inv, ld);
}

// main.cpp:162-164 / main.cpp:109
int ivp(uint32_t m, uint32_t s, const double* Din, const double* x_base, double* out, size_t ld) {
  if (Din == nullptr || x_base == nullptr || out == nullptr || ld < m) return SRI_EINVAL;
  for (uint32_t c = 0; c < s; ++c) {
    const double xb = x_base[c];
    double* col = out + c * ld;
    for (uint32_t i = 0; i < m; ++i) col[i] = Din[i] * xb;
  }
  return SRI_OK;
}

// Barycentric interpolation between two Chebyshev-Lobatto grids on [0, 1] (not in the reference: used to start the
// fixed point of a fine grid from a coarse-grid solution).  P is n_to x n_from, column-major:
//   P(i, j) = (w_j / (x_i - xc_j)) / sum_l (w_l / (x_i - xc_l)),  w_j = (-1)^j (halved at both ends);  P(i, j) = delta
// on a row whose node coincides with a coarse node.  Accumulated in long double, rounded once.
int cheb_interp(uint32_t n_from, uint32_t n_to, double* P, size_t ld) {
  if (n_from < 2 || n_to < 2 || P == nullptr || ld < n_to) return SRI_EINVAL;
  std::vector<double> xc, xf;
  std::vector<long double> t;
  try { xc.resize(n_from); xf.resize(n_to); t.resize(n_from); } catch (const std::bad_alloc&) { return SRI_ENOMEM; }
  cheb_points(n_from, 1.0, xc.data());
  cheb_points(n_to, 1.0, xf.data());
  for (uint32_t i = 0; i < n_to; ++i) {
    int hit = -1;
    long double denom = 0.0L;
    for (uint32_t j = 0; j < n_from; ++j) {
      const long double d = static_cast<long double>(xf[i]) - static_cast<long double>(xc[j]);
      if (d == 0.0L) { hit = static_cast<int>(j); break; }
      const long double wj = ((j & 1) ? -1.0L : 1.0L) * ((j == 0 || j == n_from - 1) ? 0.5L : 1.0L);
      t[j] = wj / d;
      denom += t[j];
    }
    for (uint32_t j = 0; j < n_from; ++j)
      P[i + static_cast<size_t>(j) * ld] = hit >= 0 ? (static_cast<int>(j) == hit ? 1.0 : 0.0) : static_cast<double>(t[j] / denom);
  }
  return SRI_OK;
}

// The integration operator a COARSE grid should use when its solution only starts the fixed point of a fine grid.
// In exact arithmetic Dn_NN^-1 integrates polynomials exactly on either grid and -Dn_NN^-1 Dn_IN is the vector of ones (a
// differentiation matrix annihilates constants).  As computed, both are off by the conditioning of the inverse — 2e-14 on a
// 256-node grid, 8e-14 on 1024 nodes — SMOOTHLY along the rod and differently on every grid.  The fine fixed point contains
// the fine grid's deviation; a start that contains the coarse grid's instead is a smooth ~1e-14 away from it, which
// restarted Picard iteration removes at L/(k+1) per sweep (five or six sweeps to get under 1e-14), while the rounding noise
// proper is damped by the first sweep.  So the coarse solve uses the FINE operator seen from its own nodes:
//   A_c = R_u A_f P~            u_c = R [-w_f ; 1]   (returned as w_c = -u_c)
// R: fine nodes -> coarse nodes (base node last on both grids, sri_cheb_interp), R_u its columns of unknown fine nodes;
// P~: the n_c - 1 unknown coarse nodes -> the unknown fine nodes (the right-hand side has no value at the base node),
// barycentric with weights 1 / prod (x_j - x_k).  Host, long double accumulation, once per handle (may throw std::bad_alloc: callers contain it).
int coarse_view_of_fine_operator(const double* A_f, size_t lda, const double* w_f, uint32_t n_f, uint32_t n_c, std::vector<double>& A_c,
                                 std::vector<double>& w_c) {
  if (A_f == nullptr || w_f == nullptr || n_c < 3 || n_f <= n_c || lda < n_f - 1) return SRI_EINVAL;
  const uint32_t m_f = n_f - 1, m_c = n_c - 1;
  std::vector<double> R(static_cast<size_t>(n_c) * n_f), xc(n_c), xf(n_f);
  int rc = cheb_interp(n_f, n_c, R.data(), n_c);
  if (rc != SRI_OK) return rc;
  cheb_points(n_c, 1.0, xc.data());
  cheb_points(n_f, 1.0, xf.data());
  w_c.assign(m_c, 0.0);
  for (uint32_t i = 0; i < m_c; ++i) {
    long double s = 0.0L;
    for (uint32_t j = 0; j < n_f; ++j)
      s += static_cast<long double>(R[i + static_cast<size_t>(j) * n_c]) * (j < m_f ? -static_cast<long double>(w_f[j]) : 1.0L);
    w_c[i] = -static_cast<double>(s);
  }
  std::vector<long double> bw(m_c), P(static_cast<size_t>(m_f) * m_c), t(m_c);
  for (uint32_t j = 0; j < m_c; ++j) {
    long double prod = 1.0L;
    for (uint32_t k = 0; k < m_c; ++k)
      if (k != j) prod *= static_cast<long double>(xc[j]) - static_cast<long double>(xc[k]);
    bw[j] = 1.0L / prod;
  }
  for (uint32_t i = 0; i < m_f; ++i) {
    int hit = -1;
    long double denom = 0.0L;
    for (uint32_t j = 0; j < m_c; ++j) {
      const long double d = static_cast<long double>(xf[i]) - static_cast<long double>(xc[j]);
      if (d == 0.0L) { hit = static_cast<int>(j); break; }
      t[j] = bw[j] / d;
      denom += t[j];
    }
    for (uint32_t j = 0; j < m_c; ++j)
      P[i + static_cast<size_t>(j) * m_f] = hit >= 0 ? (static_cast<int>(j) == hit ? 1.0L : 0.0L) : t[j] / denom;
  }
  std::vector<long double> AP(static_cast<size_t>(m_f) * m_c, 0.0L);             // A_f P~
  for (uint32_t j = 0; j < m_c; ++j)
    for (uint32_t k = 0; k < m_f; ++k) {
      const long double pkj = P[k + static_cast<size_t>(j) * m_f];
      const double* col = A_f + static_cast<size_t>(k) * lda;
      long double* out = AP.data() + static_cast<size_t>(j) * m_f;
      for (uint32_t i = 0; i < m_f; ++i) out[i] += static_cast<long double>(col[i]) * pkj;
    }
  A_c.assign(static_cast<size_t>(m_c) * m_c, 0.0);
  for (uint32_t j = 0; j < m_c; ++j)
    for (uint32_t i = 0; i < m_c; ++i) {
      long double s = 0.0L;
      for (uint32_t k = 0; k < m_f; ++k) s += static_cast<long double>(R[i + static_cast<size_t>(k) * n_c]) * AP[k + static_cast<size_t>(j) * m_f];
      A_c[i + static_cast<size_t>(j) * m_c] = static_cast<double>(s);
    }
  return SRI_OK;
}

}  // namespace sri

// ------------------------------------------------------------------------------------------
// C ABI (include/sri_capi.h)
// ------------------------------------------------------------------------------------------
extern "C" {

int sri_cheb_points(uint32_t n, double L, double* x) { return sri::cheb_points(n, L, x); }
int sri_cheb_coeffs(uint32_t n, double* c) { return sri::cheb_coeffs(n, c); }
int sri_cheb_dn(uint32_t n, double* D, size_t ld) { return sri::cheb_dn(n, D, ld); }
int sri_cheb_blocks(uint32_t n, const double* D, size_t ld, double* Dnn, size_t ldnn, double* Din) {
  return sri::cheb_blocks(n, D, ld, Dnn, ldnn, Din);
}
int sri_cheb_integration(uint32_t n, double* Dnn_inv, size_t ld, double* Din) {
  return sri::cheb_integration(n, Dnn_inv, ld, Din);
}
int sri_ivp(uint32_t m, uint32_t s, const double* Din, const double* x_base, double* out, size_t ld) {
  return sri::ivp(m, s, Din, x_base, out, ld);
}

int sri_cheb_interp(uint32_t n_from, uint32_t n_to, double* P, size_t ld) { return sri::cheb_interp(n_from, n_to, P, ld); }
int sri_cheb_coarse_view(uint32_t n_fine, const double* A_fine, size_t lda, const double* w_fine, uint32_t n_coarse, double* A_coarse,
                         size_t ldc, double* w_coarse) {
  if (A_coarse == nullptr || w_coarse == nullptr || n_coarse < 3 || ldc < n_coarse - 1) return SRI_EINVAL;
  try {
    std::vector<double> A, w;
    const int rc = sri::coarse_view_of_fine_operator(A_fine, lda, w_fine, n_fine, n_coarse, A, w);
    if (rc != SRI_OK) return rc;
    const uint32_t mc = n_coarse - 1;
    for (uint32_t j = 0; j < mc; ++j)
      for (uint32_t i = 0; i < mc; ++i) A_coarse[i + static_cast<size_t>(j) * ldc] = A[i + static_cast<size_t>(j) * mc];
    for (uint32_t i = 0; i < mc; ++i) w_coarse[i] = w[i];
    return SRI_OK;
  } catch (const std::bad_alloc&) {
    return SRI_ENOMEM;
  }
}

const char* sri_strerror(int status) {
  switch (status) {
    case SRI_OK: return "ok";
    case SRI_EINVAL: return "invalid argument";
    case SRI_ENOMEM: return "out of memory";
    case SRI_ESINGULAR: return "singular matrix";
    case SRI_EHIP: return "HIP runtime error";
    case SRI_ENODEVICE: return "no usable HIP device";
    case SRI_EUNSUPPORTED: return "unsupported request";
    case SRI_ENOTCONVERGED: return "fixed-point iteration did not converge";
    case SRI_ECOMM: return "RCCL error";
    default: return "unknown sri_status";
  }
}

int sri_abi_version(void) { return SRI_ABI_VERSION; }

}  // extern "C"
