// Exercises the three reference template signatures through the drop-in header and dumps the values
// with 17 significant digits (tests/test_dropin_header.py compares them with the oracle bit for bit).
#include <cstdio>

#include "chebyshev_differentiation.h"

template <unsigned N>
static void dump() {
  const auto x = ComputeChebyshevPoints<N>();          // reference call syntax, main.cpp:15
  const auto x3 = ComputeChebyshevPoints<N, 3>();
  const auto c = GetCoefficients_c<N>();
  const auto Dn = getDn<N>();                          // main.cpp:93
  std::printf("N %u\n", N);
  for (unsigned i = 0; i < N; ++i) std::printf("x %.17g %.17g c %.17g\n", x[i], x3[i], c[i]);
  for (unsigned j = 0; j < N; ++j)
    for (unsigned i = 0; i < N; ++i) std::printf("D %.17g\n", Dn(i, j));
  // main.cpp:94-95 style slicing
  const auto Dn_NN = Dn.block(0, 0, N - 1, N - 1);
  const auto Dn_IN = Dn.block(0, N - 1, N - 1, 1);
  std::printf("blk %.17g %.17g\n", Dn_NN(N - 2, N - 2), Dn_IN(0, 0));
}

int main() {
  dump<2>();
  dump<16>();
  dump<64>();
  const auto big = getDn<300>();                       // beyond the reference header's fixed-size limit
  std::printf("big %zu %.17g\n", static_cast<size_t>(big.rows()), big(0, 0));
  return 0;
}
