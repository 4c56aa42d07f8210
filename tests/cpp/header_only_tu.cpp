// Compile-only check of the drop-in header: a translation unit that includes NOTHING but
// chebyshev_differentiation.h and then uses the standard facilities the reference's main.cpp gets from it
// transitively — std::array (main.cpp:15 via the template's return type), std::cout / std::endl (main.cpp:198,202),
// <cmath> and <algorithm> (what the reference header's own bodies call: h:25-26, 48).
#include "chebyshev_differentiation.h"

static constexpr unsigned int number_of_Chebyshev_points = 16;            // main.cpp:4
static const auto x = ComputeChebyshevPoints<number_of_Chebyshev_points>();   // main.cpp:15, at namespace scope

int main() {
  std::array<double, 3> a{{3.0, 1.0, 2.0}};
  std::sort(a.begin(), a.end());
  const auto Dn = getDn<number_of_Chebyshev_points>();
  std::cout << "x0 : \n" << x[0] << " " << std::cos(0.0) + a[0] << " " << Dn.rows() << std::endl;
  return 0;
}
