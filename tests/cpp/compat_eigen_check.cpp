// Unit checks of compat/Eigen/Dense, compat/unsupported/Eigen/KroneckerProduct and compat/boost/.../legendre.hpp:
// every construct the reference's main.cpp / utilities.h use, against hand-computed values.
#include <Eigen/Dense>
#include <unsupported/Eigen/KroneckerProduct>
#include <boost/math/special_functions/chebyshev.hpp>
#include <boost/math/special_functions/legendre.hpp>

#include <cstdio>
#include <cstdlib>
#include <sstream>

#define CHECK(c)                                                   \
  do {                                                             \
    if (!(c)) {                                                    \
      std::printf("FAILED %s:%d  %s\n", __FILE__, __LINE__, #c);   \
      std::exit(1);                                                \
    }                                                              \
  } while (0)

static bool close(double a, double b, double tol = 1e-14) { return std::fabs(a - b) <= tol * (1 + std::fabs(b)); }

int main() {
  using namespace Eigen;
  // fixed-size matrices from unsigned constants, Zero / Identity, comma initialiser (main.cpp:9-10,72-75)
  static constexpr unsigned int dim = 4;
  Matrix<double, dim, dim> A;
  A << 0, -1, -2, -3,
       1,  0,  3, -2,
       2, -3,  0,  1,
       3,  2, -1,  0;
  CHECK(A(1, 0) == 1 && A(0, 3) == -3 && A(3, 2) == -1 && A.rows() == 4 && A.cols() == 4);
  A = 0.5 * A;                                                       // main.cpp:41
  CHECK(A(2, 1) == -1.5);
  CHECK((Matrix<double, 3, 3>::Identity()(1, 1)) == 1 && (Matrix<double, 3, 3>::Zero()(1, 1)) == 0);
  MatrixXd I4 = MatrixXd::Identity(4, 4);
  CHECK(I4(2, 2) == 1 && I4(2, 1) == 0);

  // dynamic vectors, size constructor, linear access (main.cpp:106-107,130-133)
  VectorXd q(4);
  q << 1, 0, 0, 0;
  CHECK(q.rows() == 4 && q.cols() == 1 && q(0) == 1 && q[3] == 0 && q.size() == 4);
  MatrixXd asMatrix = q;                                             // VectorXd -> MatrixXd by value (main.cpp:121,170)
  CHECK(asMatrix.rows() == 4 && asMatrix.cols() == 1 && asMatrix(0) == 1);

  // block<>, row(), (i, all), diagonal(), rowwise().sum()
  MatrixXd D(3, 3);
  D << 1, 2, 3,
       4, 5, 6,
       7, 8, 10;
  const MatrixXd Dc = D;
  const MatrixXd Dnn = Dc.block<2, 2>(0, 0);                         // main.cpp:94
  const MatrixXd Din = Dc.block<2, 1>(0, 2);                         // main.cpp:95
  CHECK(Dnn(1, 1) == 5 && Din(1, 0) == 6 && Din.rows() == 2 && Din.cols() == 1);
  Matrix<double, 2, 3> b;
  Vector3d e(1, 0, 0);                                               // main.cpp:136
  b.block<1, 3>(1, 0) = (Dc * e).transpose();                        // row block as an lvalue
  CHECK(b(1, 0) == 1 && b(1, 1) == 4 && b(1, 2) == 7 && b(0, 0) == 0);
  Vector3d r_init;
  r_init << 1, 2, 3;
  b.row(0) = 2.0 * r_init.transpose();                               // main.cpp:164
  CHECK(b(0, 2) == 6);
  MatrixXd X(2, 3);
  X(1, Eigen::all) = RowVectorXd::Constant(1, 3, 7.0);               // reference h:71
  CHECK(X(1, 2) == 7 && X(0, 2) == 0);
  VectorXd rs = Dc.rowwise().sum();
  CHECK(rs(0) == 6 && rs(2) == 25);
  D.diagonal() -= D.rowwise().sum();                                 // reference h:104
  CHECK(D(0, 0) == -5 && D(1, 1) == -10 && D(2, 2) == -15 && D(0, 1) == 2);
  MatrixXd sym = Dc - Dc.transpose() + MatrixXd::Identity(3, 3);     // reference h:89
  CHECK(sym(0, 1) == -2 && sym(1, 0) == 2 && sym(2, 2) == 1);

  // product, inverse (partial-pivot LU), expression chain of main.cpp:113
  MatrixXd Dinv = Dc.inverse();
  MatrixXd P = Dinv * Dc;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) CHECK(close(P(i, j), i == j ? 1.0 : 0.0, 1e-13));
  CHECK(close(Dinv(0, 0), -2.0 / 3.0, 1e-13) && close(Dinv(2, 2), 1.0, 1e-13));
  const auto zero = Matrix<double, 3, 1>::Zero();
  VectorXd ivp(3);
  ivp << 1, 1, 1;
  VectorXd sol = Dc.inverse() * (zero - ivp);
  VectorXd back = Dc * sol;
  CHECK(close(back(0), -1) && close(back(1), -1) && close(back(2), -1));
  bool threw = false;
  try {
    VectorXd bad = Dc * q;                                           // 3x3 times 4x1
    (void)bad;
  } catch (const std::logic_error&) {
    threw = true;
  }
  CHECK(threw);

  // Kronecker product (main.cpp:97-100, utilities.h:63)
  const MatrixXd K = KroneckerProduct(MatrixXd::Identity(2, 2), Dnn);
  CHECK(K.rows() == 4 && K.cols() == 4 && K(2, 2) == 1 && K(3, 3) == 5 && K(0, 2) == 0 && K(3, 2) == 4);
  const Matrix<double, 4, 4> Kfixed = KroneckerProduct(MatrixXd::Identity(2, 2), Dnn);
  CHECK(Kfixed(1, 0) == 4);
  Matrix<double, 3, 1> phi;
  phi[0] = 1; phi[1] = 2; phi[2] = 3;
  MatrixXd Phi = KroneckerProduct(Matrix<double, 2, 2>::Identity(), phi.transpose());
  CHECK(Phi.rows() == 2 && Phi.cols() == 6 && Phi(1, 4) == 2 && Phi(0, 4) == 0);
  Matrix<double, 6, 1> qe;
  qe << 1, 1, 1, 2, 2, 2;
  Vector2d Kv = Phi * qe;                                            // MatrixXd * fixed vector -> fixed vector (main.cpp:69)
  CHECK(Kv(0) == 6 && Kv(1) == 12);

  // comma initialiser with blocks (utilities.h:33-34)
  MatrixXd big(4, 4);
  big << Dnn, MatrixXd::Zero(2, 2),
         Dnn.transpose(), Dnn;
  CHECK(big(1, 0) == 4 && big(0, 2) == 0 && big(2, 1) == 4 && big(3, 3) == 5);

  // Quaterniond: (w, x, y, z), brace assignment, toRotationMatrix without normalisation (main.cpp:125-136)
  Quaterniond quat;
  quat = {0.5, 0.1, -0.2, 0.3};
  const Matrix3d R = quat.toRotationMatrix();
  const double w = 0.5, x = 0.1, y = -0.2, z = 0.3;
  CHECK(close(R(0, 0), 1 - 2 * (y * y + z * z)) && close(R(1, 0), 2 * (x * y + w * z)) && close(R(2, 0), 2 * (x * z - w * y)));
  CHECK(close(R(0, 1), 2 * (x * y - w * z)) && close(R(2, 2), 1 - 2 * (x * x + y * y)));

  // operator<< : Eigen's default IOFormat
  std::ostringstream os;
  MatrixXd pr(2, 2);
  pr << 1, -0.25, 100.5, 1.0 / 3.0;
  os << pr;
  CHECK(os.str() == "       1    -0.25\n   100.5 0.333333");
  std::ostringstream ov;
  ov << q;
  CHECK(ov.str() == "1\n0\n0\n0");

  // Boost stand-ins
  CHECK(boost::math::legendre_p(0, 0.3) == 1.0 && boost::math::legendre_p(1, 0.3) == 0.3);
  CHECK(close(boost::math::legendre_p(2, 0.3), 0.5 * (3 * 0.09 - 1)) && close(boost::math::legendre_p(3, -0.5), 0.4375));
  CHECK(close(boost::math::chebyshev_t(3, 0.5), -1.0));
  std::printf("compat ok\n");
  return 0;
}
