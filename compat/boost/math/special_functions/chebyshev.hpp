// compat/boost/math/special_functions/chebyshev.hpp — the reference includes this header (utilities.h:13) and uses
// nothing from it; chebyshev_t is provided so that the include is not an empty shell.
#ifndef SRI_COMPAT_BOOST_CHEBYSHEV
#define SRI_COMPAT_BOOST_CHEBYSHEV

namespace boost {
namespace math {

template <class T>
inline T chebyshev_next(T x, T Tn, T Tn_1) {
  return 2 * x * Tn - Tn_1;
}

template <class T>
inline T chebyshev_t(unsigned n, T x) {
  T t0 = T(1), t1 = x;
  if (n == 0) return t0;
  for (unsigned k = 1; k < n; ++k) {
    const T next = chebyshev_next(x, t1, t0);
    t0 = t1;
    t1 = next;
  }
  return t1;
}

}  // namespace math
}  // namespace boost

#endif  // SRI_COMPAT_BOOST_CHEBYSHEV
