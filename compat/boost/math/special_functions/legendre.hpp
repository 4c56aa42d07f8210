// compat/boost/math/special_functions/legendre.hpp — stand-in for the one Boost.Math function the reference calls:
// boost::math::legendre_p(l, x) at utilities.h:59 (Legendre polynomial P_l(x), |x| <= 1), by the three-term recurrence
// (l+1) P_{l+1} = (2l+1) x P_l - l P_{l-1} that Boost's legendre_imp also uses.
#ifndef SRI_COMPAT_BOOST_LEGENDRE
#define SRI_COMPAT_BOOST_LEGENDRE

#include <stdexcept>

namespace boost {
namespace math {

template <class T>
inline T legendre_next(unsigned l, T x, T Pl, T Plm1) {
  return ((2 * l + 1) * x * Pl - l * Plm1) / (l + 1);
}

template <class T>
inline T legendre_p(int l, T x) {
  if (x < T(-1) || x > T(1)) throw std::domain_error("legendre_p: x outside [-1, 1]");
  if (l < 0) l = -l - 1;                 // P_{-l-1} = P_l
  T p0 = T(1);
  if (l == 0) return p0;
  T p1 = x;
  for (unsigned n = 1; n < static_cast<unsigned>(l); ++n) {
    const T next = legendre_next(n, x, p1, p0);
    p0 = p1;
    p1 = next;
  }
  return p1;
}

}  // namespace math
}  // namespace boost

#endif  // SRI_COMPAT_BOOST_LEGENDRE
